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
