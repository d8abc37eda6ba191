"""world_size-2 CPU (gloo) coverage of the multi-BAM path: each rank builds the accumulating
buffer of its own sample (difference arrays + tile sums, same layout as the engine), the buffers
are summed with tools/multi_torch.py: sum_to_root, the root prefix-sums with the 18-bit wrap of list
mode — and must equal the oracle's per-base increments over BOTH samples (PD:2704-3014)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILE = 8192
LENS = [30000, 9000, 1, 20001]


def layout(lens):
    off = [0]
    for l in lens:
        off.append(off[-1] + (l + 1 + TILE - 1) // TILE * TILE)
    return np.array(off, dtype=np.int64)


def sample_runs(seed, n=40000, pile=0):
    rng = np.random.default_rng(seed)
    tid = rng.integers(0, len(LENS), n)
    L = np.asarray(LENS)[tid]
    beg = (rng.random(n) * L).astype(np.int64)
    end = np.minimum(beg + rng.integers(1, 300, n), L)
    runs = np.stack([tid, beg, end], axis=1)
    if pile:
        runs = np.concatenate([runs, np.tile(np.array([[0, 500, 520]]), (pile, 1))])
    return runs[runs[:, 1] < runs[:, 2]].astype(np.int32)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from tools import multi_torch as multi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert multi.assign_samples(5, rank, world) == [k for k in range(5) if k % world == rank]
        off = layout(LENS)
        n_cells = int(off[-1])
        buf = np.zeros(n_cells + n_cells // TILE, dtype=np.int32)       # [diff | tile sums]
        runs = sample_runs(100 + rank, pile=140000)                     # 2 x 140000 > 2^18: the wrap matters
        gb = off[runs[:, 0]] + runs[:, 1]
        ge = off[runs[:, 0]] + runs[:, 2]
        np.add.at(buf, gb, 1); np.add.at(buf, ge, -1)
        np.add.at(buf, n_cells + gb // TILE, 1); np.add.at(buf, n_cells + ge // TILE, -1)
        t = torch.from_numpy(buf)
        is_root = multi.sum_to_root(t, 0)
        if is_root:
            diff = buf[:n_cells].astype(np.int64)
            sums = buf[n_cells:].astype(np.int64)
            # the engine's sweep: tile carries from the tile sums, then a scan inside each tile
            carry = np.concatenate([[0], np.cumsum(sums)[:-1]])
            depth = (np.cumsum(diff.reshape(-1, TILE), axis=1) + carry[:, None]).reshape(-1)
            assert np.array_equal(depth, np.cumsum(diff))               # tile sums stay consistent under the reduce
            depth = (depth & 0x3FFFF).astype(np.uint32)
            import pd_oracle as O
            both = np.concatenate([sample_runs(100 + r, pile=140000) for r in range(world)])
            d, ooff = O.depth_from_intervals(LENS, both, wrap18=True)
            for c, ln in enumerate(LENS):
                assert np.array_equal(depth[off[c]:off[c] + ln], d[ooff[c]:ooff[c] + ln]), c
            assert depth[off[0] + 510] >= (2 * 140000) % 262144        # the pile wrapped past 2^18
        out.put((rank, "ok"))
    except Exception as e:          # pragma: no cover - surfaced through the queue
        out.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


class FakeEngine:
    """CPU stand-in with the engine's export/import contract (pd_export_i8 / pd_import_i8), so that
    tools/multi_torch.py: PackedSum's collective protocol can run under gloo."""

    def __init__(self, buf, n_cells):
        self.buf, self.n_cells, self.reg = buf, n_cells, {}

    def device_layout(self):
        return self.n_cells, self.buf.size - self.n_cells

    def synchronize(self):
        pass

    def export_i8(self, thr, i8_ptr, exc_ptr, cap, count_ptr):
        i8, exc, cnt = self.reg[i8_ptr], self.reg[exc_ptr], self.reg[count_ptr]
        d = self.buf[:self.n_cells]
        big = np.nonzero(np.abs(d) > thr)[0]
        img = d.copy(); img[big] = 0
        i8.numpy()[:] = (img + thr).astype(np.uint8)                    # biased bytes
        cnt.numpy()[0] = big.size
        e = exc.numpy()
        e[:big.size, 0] = big
        e[:big.size, 1] = d[big].astype(np.int64) & 0xFFFFFFFF        # {i32 value; i32 pad} little-endian

    def import_i8(self, i8_ptr, bias, exc_ptr, n_exc):
        self.buf[:self.n_cells] = self.reg[i8_ptr].numpy().astype(np.int32) - bias
        if n_exc:
            e = self.reg[exc_ptr].numpy()[:n_exc]
            np.add.at(self.buf, e[:, 0], (e[:, 1] & 0xFFFFFFFF).astype(np.uint32).view(np.int32))


def _worker_packed(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from tools import multi_torch as multi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off = layout(LENS)
        n_cells = int(off[-1])
        buf = np.zeros(n_cells + n_cells // TILE, dtype=np.int32)
        runs = sample_runs(200 + rank, pile=500 + 100 * rank)             # the pile exceeds 127 // world
        gb = off[runs[:, 0]] + runs[:, 1]
        ge = off[runs[:, 0]] + runs[:, 2]
        np.add.at(buf, gb, 1); np.add.at(buf, ge, -1)
        np.add.at(buf, n_cells + gb // TILE, 1); np.add.at(buf, n_cells + ge // TILE, -1)
        eng = FakeEngine(buf, n_cells)
        sums = torch.from_numpy(buf)[n_cells:]
        ps = multi.PackedSum(eng, "cpu", sums=sums)
        for t in (ps.i8, ps.exc, ps.count):
            eng.reg[t.data_ptr()] = t
        orig_import = eng.import_i8

        def import_with_gathered(i8_ptr, bias, exc_ptr, n_exc):             # the gathered list is a new tensor
            if exc_ptr not in eng.reg and n_exc:
                eng.reg[exc_ptr] = ps._all_exc
            orig_import(i8_ptr, bias, exc_ptr, n_exc)
        eng.import_i8 = import_with_gathered
        is_root = ps.run(0)
        assert is_root == (rank == 0)
        if is_root:
            diff = buf[:n_cells].astype(np.int64)
            sums64 = buf[n_cells:].astype(np.int64)
            assert np.array_equal(sums64, diff.reshape(-1, TILE).sum(axis=1))   # reduced tile sums match the reduced cells
            depth = (np.cumsum(diff) & 0x3FFFF).astype(np.uint32)
            import pd_oracle as O
            both = np.concatenate([sample_runs(200 + r, pile=500 + 100 * r) for r in range(world)])
            d, ooff = O.depth_from_intervals(LENS, both, wrap18=True)
            for c, ln in enumerate(LENS):
                assert np.array_equal(depth[off[c]:off[c] + ln], d[ooff[c]:ooff[c] + ln]), c
        out.put((rank, "ok"))
    except Exception as e:          # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def test_packed_int8_sum_protocol():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_sum_of_difference_arrays_equals_list_mode():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---------------------------------------------------------------------------------------------
# sliced sum (tools/multi_torch.py: SlicedSum): all-to-all of 4-bit images, every rank sweeps its slice
# ---------------------------------------------------------------------------------------------
PART = np.dtype([("c0", "<u4"), ("c1", "<u4"), ("s0", "<u8"), ("s1", "<u8")])      # PD_TILE_PARTIAL_BYTES = 24


class FakeSliceEngine:
    """numpy stand-in with the contract of pd_export_i4 / pd_slice_sweep_i4 / pd_gather_windows
    (include/pandepth_amd.h), so that SlicedSum's collective protocol can run under gloo."""

    def __init__(self, buf, n_cells, lens):
        self.buf, self.n_cells, self.lens, self.reg = buf, n_cells, list(lens), []
        self.off = layout(lens)

    def device_layout(self):
        return self.n_cells, self.buf.size - self.n_cells

    def stream(self):
        return 0

    def synchronize(self):
        pass

    def at(self, ptr, dtype, count):
        """numpy view of `count` items at address `ptr` inside one of the registered tensors"""
        for t in self.reg:
            base, nbytes = t.data_ptr(), t.numel() * t.element_size()
            if base <= ptr < base + nbytes or (ptr == base and nbytes == 0):
                flat = t.view(-1).view(torch.uint8).numpy()
                o = ptr - base
                return flat[o:o + count * np.dtype(dtype).itemsize].view(dtype)
        raise KeyError(hex(ptr))

    def export_i4(self, i4_ptr, exc_ptr, cap, count_ptr):
        d = self.buf[:self.n_cells]
        big = np.nonzero((d > 7) | (d < -8))[0]
        img = d.copy(); img[big] = 0
        nib = (img + 8).astype(np.uint8)
        self.at(i4_ptr, np.uint8, self.n_cells // 2)[:] = nib[0::2] | (nib[1::2] << 4)
        self.at(count_ptr, np.int32, 1)[0] = big.size
        e = self.at(exc_ptr, np.int64, 2 * cap).reshape(cap, 2)
        k = min(big.size, cap)
        e[:k, 0] = big[:k]
        e[:k, 1] = d[big[:k]].astype(np.int64) & 0xFFFFFFFF

    def slice_sweep_i4(self, parts_ptr, n_parts, stride, tile_first, tile_count, sums_ptr, exc_ptr, exc_stride,
                       counts_ptr, w, min_dep, wrap_bits, part_ptr):
        assert w >= TILE
        n_tiles = self.n_cells // TILE
        n = tile_count * TILE
        acc = np.full(n, -8 * n_parts, dtype=np.int64)
        for j in range(n_parts):
            b = self.at(parts_ptr + j * stride, np.uint8, n // 2)
            acc[0::2] += b & 0xf
            acc[1::2] += b >> 4
        counts = self.at(counts_ptr, np.int32, n_parts)
        for j in range(n_parts):
            e = self.at(exc_ptr + j * exc_stride * 16, np.int64, 2 * exc_stride).reshape(exc_stride, 2)[:min(int(counts[j]), exc_stride)]
            for cell, val in e:
                c = int(cell) - tile_first * TILE
                if 0 <= c < n:
                    acc[c] += np.int64(np.uint32(val & 0xFFFFFFFF).astype(np.int32))
        sums = self.at(sums_ptr, np.int32, n_tiles).astype(np.int64)
        carry = np.concatenate([[0], np.cumsum(sums)[:-1]])
        mask = 0xFFFFFFFF if wrap_bits in (0, 32) else (1 << wrap_bits) - 1
        parts = self.at(part_ptr, PART, tile_count)
        tile_contig = np.searchsorted(self.off, np.arange(n_tiles) * TILE, side="right") - 1
        for i in range(tile_count):
            t = tile_first + i
            depth = (np.cumsum(acc[i * TILE:(i + 1) * TILE]) + carry[t]) & mask
            c = int(tile_contig[t]); local0 = t * TILE - int(self.off[c]); clen = self.lens[c]
            k0 = local0 // w; nb = (k0 + 1) * w - local0
            pos = np.arange(TILE)
            ok = (local0 + pos < clen) & (depth >= min_dep)
            a, b = ok & (pos < nb), ok & (pos >= nb)
            parts[i] = (a.sum(), b.sum(), depth[a].sum(), depth[b].sum())

    def window_layout(self, w):
        return np.concatenate([[0], np.cumsum([(l + w - 1) // w for l in self.lens])]).astype(np.uint64)

    def gather_windows(self, part_ptr, w):
        n_tiles = self.n_cells // TILE
        parts = self.at(part_ptr, PART, n_tiles)
        wo = self.window_layout(w)
        cover = np.zeros(int(wo[-1]), dtype=np.uint32); tot = np.zeros(int(wo[-1]), dtype=np.uint64)
        for c, ln in enumerate(self.lens):
            for t in range(int(self.off[c]) // TILE, int(self.off[c + 1]) // TILE):
                local0 = t * TILE - int(self.off[c])
                if local0 >= ln:
                    continue
                k0 = local0 // w
                cover[int(wo[c]) + k0] += parts[t]["c0"]; tot[int(wo[c]) + k0] += parts[t]["s0"]
                if parts[t]["c1"]:
                    cover[int(wo[c]) + k0 + 1] += parts[t]["c1"]; tot[int(wo[c]) + k0 + 1] += parts[t]["s1"]
        return wo, cover, tot


def _worker_sliced(rank, world, port, out, lens, pipelined):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from tools import multi_torch as multi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pd_oracle as O
        off = layout(lens)
        n_cells = int(off[-1])
        W, MIN_DEP = 10000, 2

        def fill(buf, seed):
            buf[:] = 0
            runs = sample_runs_on(lens, seed, pile=300 + 50 * rank)          # piles beyond the nibble range: exceptions
            gb = off[runs[:, 0]] + runs[:, 1]
            ge = off[runs[:, 0]] + runs[:, 2]
            np.add.at(buf, gb, 1); np.add.at(buf, ge, -1)
            np.add.at(buf, n_cells + gb // TILE, 1); np.add.at(buf, n_cells + ge // TILE, -1)

        def expect(seeds):
            both = np.concatenate([sample_runs_on(lens, s + r, pile=300 + 50 * r) for s in seeds for r in range(world)])
            d, ooff = O.depth_from_intervals(lens, both, wrap18=True)
            cov, tot = [], []
            for c, ln in enumerate(lens):
                x = d[ooff[c]:ooff[c] + ln].astype(np.uint64)
                for s0 in range(0, ln, W):
                    seg = x[s0:s0 + W]; m = seg >= MIN_DEP
                    cov.append(int(m.sum())); tot.append(int(seg[m].sum()))
            return np.array(cov, dtype=np.uint32), np.array(tot, dtype=np.uint64)

        buf = np.zeros(n_cells + n_cells // TILE, dtype=np.int32)
        eng = FakeSliceEngine(buf, n_cells, lens)
        if pipelined:
            multi.SlicedSum.MSG_BYTES = 8192            # several send/recv groups per exchange
        ss = multi.SlicedSum(eng, "cpu", sums=torch.from_numpy(buf)[n_cells:])
        assert ss.stream_mode == "sync" and ss.slice_tiles * world >= n_cells // TILE
        for s in ss.slots:
            eng.reg += [s["send"], s["recv"], s["meta"], s["exc"], s["exc_all"], s["count"]]
        eng.reg += [ss.part_mine, ss.part_all]
        if not pipelined:
            fill(buf, 300 + rank)
            res = ss.run(W, MIN_DEP, 18, 0)
            if rank == 0:
                cov, tot = expect([300])
                assert np.array_equal(res[1], cov) and np.array_equal(res[2], tot)
            else:
                assert res is None
        else:
            # two samples per rank in flight: sample 1 is scattered while sample 0 is on the links
            fill(buf, 300 + rank); ss.start(0)
            fill(buf, 400 + rank); ss.start(1)
            r0 = ss.finish(0, W, MIN_DEP, 18, 0)
            r1 = ss.finish(1, W, MIN_DEP, 18, 0)
            if rank == 0:
                for res, seed in ((r0, 300), (r1, 400)):
                    cov, tot = expect([seed])
                    assert np.array_equal(res[1], cov) and np.array_equal(res[2], tot)
        out.put((rank, "ok"))
    except Exception:          # pragma: no cover
        import traceback
        out.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def sample_runs_on(lens, seed, n=40000, pile=0):
    rng = np.random.default_rng(seed)
    tid = rng.integers(0, len(lens), n)
    L = np.asarray(lens)[tid]
    beg = (rng.random(n) * L).astype(np.int64)
    end = np.minimum(beg + rng.integers(1, 300, n), L)
    runs = np.stack([tid, beg, end], axis=1)
    if pile:
        runs = np.concatenate([runs, np.tile(np.array([[0, 500, 520]]), (pile, 1)),
                               np.tile(np.array([[len(lens) - 1, 8190, 8200]]), (pile, 1))])
    return runs[runs[:, 1] < runs[:, 2]].astype(np.int32)


@pytest.mark.parametrize("world,lens,pipelined", [(2, LENS, False), (3, LENS + [8192, 12000], False), (2, LENS + [9000], True)])
def test_sliced_sum_protocol(world, lens, pipelined):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 33500 + (os.getpid() * 7 + world + len(lens)) % 2000
    procs = [ctx.Process(target=_worker_sliced, args=(r, world, port, out, lens, pipelined)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
