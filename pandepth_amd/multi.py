"""Multi-GPU plumbing for the `#.list` path: one sample (BAM) per GPU, then ONE sum of the
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
