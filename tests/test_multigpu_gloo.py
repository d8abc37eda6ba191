"""world_size-2 CPU (gloo) coverage of the multi-BAM path: each rank builds the accumulating
buffer of its own sample (difference arrays + tile sums, same layout as the engine), the buffers
are summed with pandepth_amd.multi.sum_to_root, the root prefix-sums with the 18-bit wrap of list
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
    from pandepth_amd import multi
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
    pandepth_amd.multi.PackedSum's collective protocol can run under gloo."""

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
    from pandepth_amd import multi
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
