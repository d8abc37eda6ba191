"""TEST AND MEASUREMENT TOOLING, not part of the product (the product's multi-GPU sum is C++ inside libpandepth_amd.so:
pd_comm_*, pd_sliced_sum_start / _finish, csrc/pd_capi.hip).  The same protocols written with torch.distributed: the world_size-2
gloo tests drive them on the CPU, bench.py keeps them as comparison modes.

Multi-GPU plumbing for the `#.list` path: one sample (BAM) per GPU, then ONE sum of the
accumulating buffers (difference arrays + tile sums are contiguous int32, include/pandepth_amd.h
pd_device_buffer) over RCCL/xGMI, scan on the root.  Difference arrays are linear, so summing them
before the prefix sum equals the reference's sequential accumulation of every file into one array
(PD:2704-3014).  torch.distributed is used as the launcher/collective layer only."""
import torch
import torch.distributed as dist


class _DevBuf:
    """Zero-copy view of a raw device pointer through __cuda_array_interface__."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4",
                                         "data": (int(ptr), False), "version": 3}


def buffer_view(engine, device):
    """int32 torch tensor aliasing the engine's accumulating buffer (no copy)."""
    ptr, n_words, _ = engine.device_buffer()
    return torch.as_tensor(_DevBuf(ptr, n_words), device=device)


def assign_samples(n_samples, rank, world):
    """Round-robin assignment of list entries to ranks (sample k -> rank k % world)."""
    return [k for k in range(n_samples) if k % world == rank]


def sum_to_root(buf, root=0, group=None):
    """In-place int32 sum of every rank's buffer into `root`'s.  Returns True on the root."""
    if dist.is_initialized():
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM, group=group)
        return dist.get_rank(group) == root
    return True


class PackedSum:
    """Sum of every rank's difference arrays into `root` with int8 transport (4x fewer bytes on the
    xGMI links than the int32 buffers):

        export   each rank packs its cells to biased bytes (d + thr, thr = 127 // world: no byte of
                 the sum can reach 256) + a short (cell, value) exception list        pd_export_i8
        reduce   the image as int32 words (1 B/cell) and the int32 tile sums          RCCL reduce
        gather   exception lists (padded to the longest, normally a few entries)     RCCL gather
        import   root widens the summed image back to int32 and adds the exceptions   pd_import_i8
    """

    EXC_CAP = 1 << 22

    def __init__(self, engine, device, group=None, sums=None):
        self.e, self.dev, self.group = engine, torch.device(device), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        n_cells, n_sums = engine.device_layout()
        self.n_cells, self.n_sums = n_cells, n_sums
        self.i8 = torch.empty(n_cells, dtype=torch.uint8, device=device)     # n_cells is a multiple of 8192
        self.i8_words = self.i8.view(torch.int32)
        self.exc = torch.zeros((self.EXC_CAP, 2), dtype=torch.int64, device=device)     # pd_exc = {u64 cell; i32 value; i32 pad}
        self.count = torch.zeros(1, dtype=torch.int32, device=device)
        # int32 tile sums: the tail of the engine's buffer (a test double may pass its own tensor)
        self.sums = sums if sums is not None else buffer_view(engine, device)[n_cells:]
        self.threshold = max(1, 127 // self.world)

    def _sync(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize()

    def run(self, root=0):
        """Returns True on the root (whose engine then holds the summed arrays)."""
        e = self.e
        e.export_i8(self.threshold, self.i8.data_ptr(), self.exc.data_ptr(), self.EXC_CAP, self.count.data_ptr())
        e.synchronize()                                   # engine stream -> torch stream: order by host sync
        n_exc = int(self.count.item())
        if n_exc > self.EXC_CAP:
            raise RuntimeError("more than %d cells exceed the int8 range: use sum_to_root" % self.EXC_CAP)
        is_root = self.rank == root
        if self.world > 1 or dist.is_initialized():
            dist.reduce(self.i8_words, dst=root, op=dist.ReduceOp.SUM, group=self.group)
            dist.reduce(self.sums, dst=root, op=dist.ReduceOp.SUM, group=self.group)
            counts = torch.zeros(self.world, dtype=torch.int64, device=self.dev)
            counts[self.rank] = n_exc
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.group)
            m = int(counts.max().item())
            if m > 0:
                mine = self.exc[:m].contiguous()
                mine[n_exc:] = 0                           # padding: cell 0, value 0 (adds nothing)
                parts = [torch.empty_like(mine) for _ in range(self.world)] if is_root else None
                dist.gather(mine, parts, dst=root, group=self.group)
                if is_root:
                    self._all_exc = torch.cat(parts, 0).contiguous()        # kept alive until the import ran
                    self._sync()
                    e.import_i8(self.i8.data_ptr(), self.world * self.threshold, self._all_exc.data_ptr(), self._all_exc.shape[0])
                    e.synchronize()
                    return True
                return False
        self._sync()
        if is_root:
            e.import_i8(self.i8.data_ptr(), self.world * self.threshold, self.exc.data_ptr(), n_exc if self.world == 1 else 0)
            e.synchronize()
        return is_root


TILE = 8192                    # include/pandepth_amd.h PD_TILE_CELLS
PART_BYTES = 24                # PD_TILE_PARTIAL_BYTES


class SlicedSum:
    """Multi-sample sum shaped for point-to-point xGMI (whole-chromosome / wide-window mode): no GPU
    ever receives everybody's arrays.  Rank r keeps the tiles [r * slice_tiles, (r + 1) * slice_tiles):

        start    export   4-bit image of this rank's difference arrays (+ exception block)   pd_export_i4
                 exchange all-to-all of the image: every pair of GPUs moves 1/world of it over its
                          own link, all links at once (grouped send/recv in messages of at most
                          MSG_BYTES: RCCL 2.26.6 silently drops the second half of a send/recv
                          larger than 1 GiB — tools/rccl_large_message_check.py); all-reduce of the int32 tile sums
                          (+ the exception counts), all-gather of the exception blocks
        finish   sweep    one fused kernel: sum of the `world` images of the slice, prefix sum with
                          carries from the summed tile sums, 18-bit wrap, window partials pd_slice_sweep_i4
                 gather   24 B per tile to the root, which adds them up per window       pd_gather_windows

    start() only enqueues (the collectives are asynchronous), so a caller with several samples per
    rank can scatter sample k+1 while sample k's image is on the links: `depth` slots of buffers.
    With stream_mode "engine" torch runs on the context's own HIP stream (ExternalStream), so kernels
    and collectives are ordered on the device and the host never waits in between; "sync" orders
    them with host synchronisation instead."""

    EXC_BLOCK = 1 << 18
    MSG_BYTES = 1 << 28

    def __init__(self, engine, device, group=None, sums=None, depth=2, stream_mode=None, self_via_collective=False):
        self.e, self.dev, self.group = engine, torch.device(device), group
        self.dist = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank = dist.get_rank(group) if self.dist else 0
        n_cells, n_sums = engine.device_layout()
        self.n_cells, self.n_sums = n_cells, n_sums
        self.n_tiles = n_cells // TILE
        self.slice_tiles = max(1, -(-self.n_tiles // self.world))
        self.slice_bytes = self.slice_tiles * (TILE // 2)
        self.tile_first = min(self.rank * self.slice_tiles, self.n_tiles)
        self.tile_count = min(self.slice_tiles, self.n_tiles - self.tile_first)
        self.sums = sums if sums is not None else buffer_view(engine, device)[n_cells:]
        if stream_mode is None:
            stream_mode = "engine" if self.dev.type == "cuda" else "sync"
        self.stream_mode = stream_mode
        self.self_via_collective = self_via_collective      # tests: route this rank's own part through send/recv too
        self._ext = torch.cuda.ExternalStream(engine.stream(), device=self.dev) if stream_mode == "engine" else None
        W, B = self.world, self.EXC_BLOCK
        kw = dict(device=self.dev)
        self.slots = []
        with self._stream():
            for _ in range(depth):
                self.slots.append(dict(
                    send=torch.zeros(W * self.slice_bytes, dtype=torch.uint8, **kw),      # tail beyond n_cells / 2 stays 0
                    recv=torch.empty(W * self.slice_bytes, dtype=torch.uint8, **kw),
                    meta=torch.zeros(n_sums + W, dtype=torch.int32, **kw),               # tile sums | exception counts
                    exc=torch.zeros((B, 2), dtype=torch.int64, **kw),                    # pd_exc = {u64 cell; i32 value; i32 pad}
                    exc_all=torch.zeros((W * B, 2), dtype=torch.int64, **kw),
                    count=torch.zeros(1, dtype=torch.int32, **kw), works=None))
            self.part_mine = torch.zeros(self.slice_tiles * PART_BYTES, dtype=torch.uint8, **kw)
            self.part_all = torch.zeros(W * self.slice_tiles * PART_BYTES, dtype=torch.uint8, **kw)
        self._sync_all()

    # -- ordering between the context's stream and torch's ------------------------------------
    def _stream(self):
        import contextlib
        return torch.cuda.stream(self._ext) if self._ext is not None else contextlib.nullcontext()

    def _sync_all(self):
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)

    def _engine_then_torch(self):
        if self._ext is None:
            self.e.synchronize()

    def _torch_then_engine(self):
        if self._ext is None:
            self._sync_all()

    # -- protocol -----------------------------------------------------------------------------
    def start(self, slot=0):
        """Packs this rank's arrays and puts them on the links; returns immediately.  The
        context may be reset and refilled right after."""
        s, W = self.slots[slot], self.world
        with self._stream():
            self.e.export_i4(s["send"].data_ptr(), s["exc"].data_ptr(), self.EXC_BLOCK, s["count"].data_ptr())
            self._engine_then_torch()
            meta = s["meta"]
            meta[:self.n_sums].copy_(self.sums)
            meta[self.n_sums:].zero_()
            meta[self.n_sums + self.rank:self.n_sums + self.rank + 1].copy_(s["count"])
            s["works"] = self._exchange(s["send"], s["recv"])
            if self.dist:
                s["works"] += [
                    dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=self.group, async_op=True),
                    dist.all_gather_into_tensor(s["exc_all"], s["exc"], group=self.group, async_op=True)]
            else:
                s["exc_all"].copy_(s["exc"])
            if self._ext is None:
                self._sync_all()          # "sync" mode: the context's next kernels must not overtake the copies above

    def _exchange(self, send, recv):
        """recv[p] <- rank p's send[self.rank], for every p: the all-to-all, as one group of
        send/recv pairs per MSG_BYTES of slice.  Returns the pending work handles."""
        W, sb = self.world, self.slice_bytes
        rs, rr = send.view(W, sb), recv.view(W, sb)
        peers = [p for p in range(W) if p != self.rank or (self.dist and self.self_via_collective)]
        if self.rank not in peers:
            rr[self.rank].copy_(rs[self.rank])
        works = []
        if peers:
            to_global = (lambda p: dist.get_global_rank(self.group, p)) if self.group is not None else (lambda p: p)
            for c0 in range(0, sb, self.MSG_BYTES):
                c1 = min(sb, c0 + self.MSG_BYTES)
                ops = []
                for p in peers:
                    ops.append(dist.P2POp(dist.isend, rs[p][c0:c1], to_global(p), group=self.group))
                    ops.append(dist.P2POp(dist.irecv, rr[p][c0:c1], to_global(p), group=self.group))
                works += dist.batch_isend_irecv(ops)
        return works

    def finish(self, slot=0, w=10000000, min_dep=1, wrap_bits=18, root=0):
        """Completes the sum started in `slot`.  Returns (win_off, cover, depth_sum) on the root —
        the results pd_scan_reduce_windows gives on a context holding every sample — else None."""
        s, W = self.slots[slot], self.world
        with self._stream():
            for wk in s["works"]:
                wk.wait()
            s["works"] = None
            self._torch_then_engine()
            meta = s["meta"]
            self.e.slice_sweep_i4(s["recv"].data_ptr(), W, self.slice_bytes, self.tile_first, self.tile_count,
                                  meta.data_ptr(), s["exc_all"].data_ptr(), self.EXC_BLOCK,
                                  meta.data_ptr() + 4 * self.n_sums, w, min_dep, wrap_bits, self.part_mine.data_ptr())
            self._engine_then_torch()
            is_root = self.rank == root
            if self.dist:
                chunks = list(self.part_all.view(W, -1).unbind(0)) if is_root else None
                dist.gather(self.part_mine, chunks, dst=root, group=self.group)
            else:
                self.part_all.copy_(self.part_mine)
            res = None
            if is_root:
                self._torch_then_engine()
                res = self.e.gather_windows(self.part_all.data_ptr(), w)
            counts = meta[self.n_sums:].tolist()
            if max(counts) > self.EXC_BLOCK:
                raise RuntimeError("a sample has %d cells outside the 4-bit range (block of %d): use PackedSum / sum_to_root"
                                   % (max(counts), self.EXC_BLOCK))
            return res

    def run(self, w=10000000, min_dep=1, wrap_bits=18, root=0):
        self.start(0)
        return self.finish(0, w, min_dep, wrap_bits, root)
