"""The library's multi-GPU code with N >= 2 ranks on ONE GPU: RCCL's entry points are taken from a loopback transport
(tests/harness/loopback_nccl.hip, selected with PANDEPTH_RCCL_LIB) that moves the bytes between contexts sharing the device,
so pd_comm_init / pd_comm_init_all, the grouped send/recv of the 4-bit slices in messages of <= 256 MiB, the all-reduce of
tile sums + exception counts, the all-gather of the exception blocks, pd_slice_sweep_i4 on every rank's slice and the gather
of the per-tile partials to the root all run as they do across GPUs — against the depth oracle (reference: the #.list
accumulate loop, PD:2704-3014).  Real multi-GPU boxes additionally run tests/test_comm_gpu.py's two-process case."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "pandepth_amd", "pandepth")
SHIM = os.path.join(HERE, "harness", "libpd_loopback_nccl.so")
MANIFEST = json.load(open(os.path.join(HERE, "golden", "manifest.json")))


@pytest.fixture(scope="module")
def shim():
    if not os.path.exists(SHIM):
        subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "libpd_loopback_nccl.so"], check=True, stdout=subprocess.DEVNULL)
    return SHIM


CHILD = r'''
import os, sys, threading, traceback, numpy as np
root, world, mode, transport = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import pandepth_amd as pda
from test_gpu_engine import LENS, rand_intervals, oracle_depth, windows_ref
uid = pda.comm_unique_id() if transport == "rccl" else None
engines, comms, bar = [None] * world, [None] * world, threading.Barrier(world)
def make_comm(e, rank):
    """rccl: pd_comm_init with the unique id, every rank for itself (here: the loopback stand-in for librccl);
    local: pd_comm_init_local over every rank's context at once — the executable's in-process transport (no library)"""
    if transport == "rccl":
        return pda.Comm(e, uid, rank, world)
    engines[rank] = e
    bar.wait()
    if rank == 0:
        comms[:] = pda.Comm.local(engines)
    bar.wait()
    return comms[rank]
K = 3                                                     # samples per rank, two in flight
def sample(r, k, n=20000):
    rng = np.random.default_rng(1000 + 10 * r + k)
    iv = rand_intervals(rng, LENS, n)
    # pile-ups beyond the 4-bit range (exceptions), one across a tile edge, one on a slice boundary candidate
    return np.concatenate([iv, np.tile(np.array([[0, 10 + k + r, 50]], dtype=np.int32), (500, 1)),
                           np.tile(np.array([[0, 8190, 8200 + r]], dtype=np.int32), (40, 1)),
                           np.tile(np.array([[1, 16380 + r, 16390 + r]], dtype=np.int32), (30, 1))])
_rg = np.random.default_rng(99)
_tid = _rg.integers(0, len(LENS), 4000)
_first = (_rg.random(4000) * np.asarray(LENS)[_tid]).astype(np.int64) + 1
REGS = np.stack([_tid, _first, np.minimum(_first + _rg.integers(0, 60000, 4000), np.asarray(LENS)[_tid])], axis=1).astype(np.int32)
REGS[0] = [0, 1, LENS[0]]; REGS[1] = [1, 8192, 8193]; REGS[2] = [0, 8193, 16384]
_b = np.repeat(np.arange(1000, 451000, 3, dtype=np.int32), 10)
OVER = np.ascontiguousarray(np.stack([np.zeros_like(_b), _b, _b + 100], axis=1))
assert int(LENS[0]) > 452000
results, errors = {}, []
def rank_main(rank):
    try:
        with pda.Engine(LENS, device=0) as e:
            c = make_comm(e, rank)
            res = []
            if mode == "pipeline":
                for k in range(K):                        # software pipeline: start(k), then finish(k - 1)
                    e.reset(); e.push_intervals(sample(rank, k)); c.start(k % 2)
                    if k: res.append(c.finish((k - 1) % 2, 10000, 1, 18, world - 1))
                res.append(c.finish((K - 1) % 2, 10000, 1, 18, world - 1))
                # the deferred form: sorted batches stay pending, pd_export_i4 packs the tile windows straight from LDS
                e.reset(); e.keep_deferred(True)
                iv = sample(rank, 7); iv = iv[np.lexsort((iv[:, 1], iv[:, 0]))]
                e.push_intervals(iv, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
                res.append(c.run(8192, 2, 18, 0))
            elif mode == "cells":
                # statistics that need the summed CELLS: every rank makes its slice's depth and reduces what lies in it
                # (pd_sliced_window_sum below 8192 cells, pd_sliced_interval_sum); root 0, then root on the last rank
                e.push_intervals(sample(rank, 3))
                for w in (100, 149, 1000, 4096, 8191):
                    res.append(c.window_sum(w, 2 if w == 149 else 1, 18, 0))
                res.append(c.interval_sum(REGS, 1, 18, world - 1))
                res.append(c.interval_sum(REGS[:7], 3, 0, 0))
                # ... and from a deferred (sorted, pending) sample: the export comes straight from the tile windows
                e.reset(); e.keep_deferred(True)
                iv = sample(rank, 5); iv = iv[np.lexsort((iv[:, 1], iv[:, 0]))]
                e.push_intervals(iv, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
                res.append(c.window_sum(100, 1, 18, 0))
            elif mode == "overflow":
                # more cells outside the 4-bit range than the exception block holds (2^18): ten reads start on every third base
                # of 450 000 bases on rank 0 (+10 at 150 000 cells, -10 at 150 000 others: the ends fall on another residue, nothing
                # cancels) -> every rank gets PD_ERANGE, no sample is consumed, the contexts still add up
                e.keep_deferred(True)
                if rank == 0:
                    iv = OVER
                else:
                    iv = sample(rank, 0); iv = iv[np.lexsort((iv[:, 1], iv[:, 0]))]
                e.push_intervals(np.ascontiguousarray(iv), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
                try:
                    c.run(8192, 1, 18, 0)
                    res.append("no error")
                except pda.PdError as ex:
                    res.append(ex.code)
                res.append(e.scan_reduce_windows(8192, 1, 18))      # the sample is still whole on this context
            results[rank] = res
            if transport != "rccl": bar.wait()            # (a local communicator's buffers are read by the peers until every rank is done)
            c.close()
    except Exception:                                       # noqa: BLE001
        errors.append("rank %d: %s" % (rank, traceback.format_exc()))
th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in th: t.start()
for t in th: t.join(240)
assert not errors, "\n".join(errors)
assert len(results) == world
if mode == "pipeline":
    for k in range(K + 1):
        kk, w, md, rootk = (k, 10000, 1, world - 1) if k < K else (7, 8192, 2, 0)
        parts = [sample(r, kk) for r in range(world)]
        d, off = oracle_depth(LENS, np.concatenate(parts), True)
        cov, tot = windows_ref(LENS, d, off, w, md)
        for r in range(world):
            got = results[r][k]
            if r == rootk:
                assert np.array_equal(got[1], cov) and np.array_equal(got[2], tot), (k, r)
            else:
                assert got is None
elif mode == "cells":
    import pd_oracle as O
    d, off = oracle_depth(LENS, np.concatenate([sample(r, 3) for r in range(world)]), True)
    for j, w in enumerate((100, 149, 1000, 4096, 8191)):
        cov, tot = windows_ref(LENS, d, off, w, 2 if w == 149 else 1)
        for r in range(world):
            got = results[r][j]
            if r == 0:
                assert np.array_equal(got[1], cov) and np.array_equal(got[2], tot), (w, int((got[1] != cov).sum()), int((got[2] != tot).sum()))
            else:
                assert got is None
    ec, es = O.stat_regions(d, off, REGS, 1)
    got = results[world - 1][5]
    assert np.array_equal(got[0], ec) and np.array_equal(got[1], es), (int((got[0] != ec).sum()), int((got[1] != es).sum()))
    d0, off0 = oracle_depth(LENS, np.concatenate([sample(r, 3) for r in range(world)]), False)
    ec, es = O.stat_regions(d0, off0, REGS[:7], 3)
    got = results[0][6]
    assert np.array_equal(got[0], ec) and np.array_equal(got[1], es)
    d, off = oracle_depth(LENS, np.concatenate([sample(r, 5) for r in range(world)]), True)
    cov, tot = windows_ref(LENS, d, off, 100, 1)
    got = results[0][7]
    assert np.array_equal(got[1], cov) and np.array_equal(got[2], tot)
else:
    for r in range(world):
        assert results[r][0] == -6, (r, results[r][0])            # PD_ERANGE on every rank
    for r in range(world):
        iv = OVER if r == 0 else sample(r, 0)
        d, off = oracle_depth(LENS, iv, True)
        cov, tot = windows_ref(LENS, d, off, 8192, 1)
        assert np.array_equal(results[r][1][1], cov) and np.array_equal(results[r][1][2], tot), r
print("LOOPBACK-OK", world, mode)
'''


TRANSPORTS = ["rccl", "local"]       # rccl: the library's RCCL calls through the loopback stand-in; local: the in-process peer-copy transport (pd_local_comm.h)


def _run(world, mode, shim, transport="rccl"):
    env = dict(os.environ, PANDEPTH_RCCL_LIB=shim) if transport == "rccl" else {k: v for k, v in os.environ.items() if k != "PANDEPTH_RCCL_LIB"}
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(world), mode, transport], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=360, env=env)
    assert p.returncode == 0 and "LOOPBACK-OK %d %s" % (world, mode) in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sliced_sum_n_ranks_through_the_library(world, shim, transport):
    """pd_comm_init (unique id, one thread per rank) + pd_sliced_sum_start / _finish: three samples pipelined through the two
    slots with the root on the LAST rank, then the deferred (direct export) form with the root on rank 0."""
    _run(world, "pipeline", shim, transport)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_cell_level_statistics_n_ranks(world, shim, transport):
    """pd_sliced_window_sum for windows below 8192 cells and pd_sliced_interval_sum: every rank turns its slice of the exchanged
    images into depth cells and reduces the windows / region stretches that lie in it — against the oracle on the summed samples
    (windows of 100, 149, 1000, 4096 and 8191 cells incl. the ones across tile and slice boundaries; 4000 regions incl. a whole
    contig and regions that start or end on a tile edge; with and without the 18-bit wrap; a deferred sample)."""
    _run(world, "cells", shim, transport)


@pytest.mark.parametrize("transport", TRANSPORTS)
def test_exception_overflow_is_a_clean_decline_on_every_rank(shim, transport):
    """A sample whose 4-bit image has more out-of-range cells than the exception block holds: PD_ERANGE everywhere, and every
    context still holds its sample (the executable then adds the contexts up with pd_accumulate_from)."""
    _run(3, "overflow", shim, transport)


LIST_WIDE = [e for e in MANIFEST if ".list" in e["args"][1] and e["fixture"] in ("f1", "f2") and "-a" not in e["args"]]


@pytest.mark.parametrize("transport", ["peer", "rccl"])
@pytest.mark.parametrize("gpus", ["2", "3"])
@pytest.mark.parametrize("case", LIST_WIDE, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_cli_list_mode_over_n_contexts_through_the_communicator(case, gpus, transport, shim, tmp_path):
    """`#.list` inputs: one context per (stand-in) GPU, pd_comm_init_local (the executable's default: in-process peer copies) or
    pd_comm_preinit + pd_comm_init_all (-X transport=rccl: here the loopback stand-in) + pd_sliced_window_sum / pd_sliced_interval_sum
    with N = 2 and 3 ranks — the same bytes as the reference for whole-chromosome tables, `-w` tables of any width and `-g` / `-b` tables."""
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, PANDEPTH_TIMING="1", PANDEPTH_TUNE="gpus=" + gpus + (",transport=rccl" if transport == "rccl" else ""), PANDEPTH_RCCL_LIB=shim)
    p = subprocess.run([CLI] + case["args"] + ["-o", str(tmp_path / "o"), "-t", "4"], cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=180, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-800:]
    # whole-chromosome bins, narrow windows and annotation intervals alike: nobody adds the contexts into one GPU
    how = b"in slices (RCCL)" if transport == "rccl" else b"in slices (in-process peer copies)"
    assert how in p.stderr and b"added into GPU" not in p.stderr and b"communicator unavailable" not in p.stderr, p.stderr.decode()[-800:]
    if transport == "rccl":
        assert b"comm ahead" in p.stderr, p.stderr.decode()[-800:]      # made at process entry, adopted by the contexts
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gzip.decompress(gz)).hexdigest() == meta["text_sha256"], suffix
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix + " (gz bytes)"


def test_messages_are_chunked_at_full_size(shim):
    """BASELINE's genome (3.0e9 cells) over 2 ranks: a slice is 750 MB, so the exchange goes in three grouped send/recv rounds of
    <= 256 MiB (RCCL 2.26 drops the second half of a message above 1 GiB: profiles/r01_rccl_large_message.txt).  Two thin samples;
    the sliced sum must equal one context holding both."""
    code = r'''
import os, sys, threading, traceback, numpy as np
root = sys.argv[1]
sys.path.insert(0, root)
import pandepth_amd as pda
from tools import synth
names, lens = synth.genome_c2()
uid = pda.comm_unique_id()
recs = [synth.gen_records_numpy(lens, 1500000, seed=60 + r) for r in range(2)]
runs = [np.concatenate(synth.records_to_runs(x)) for x in recs]
out, errors = {}, []
def rank_main(r):
    try:
        with pda.Engine(lens.astype(np.uint32), device=0) as e:
            c = pda.Comm(e, uid, r, 2)
            e.push_intervals(runs[r])
            out[r] = c.run(10000000, 1, 18, 0)
            c.close()
    except Exception:
        errors.append(traceback.format_exc())
th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
for t in th: t.start()
for t in th: t.join(900)
assert not errors, "\n".join(errors)
with pda.Engine(lens.astype(np.uint32), device=0) as e:
    e.push_intervals(np.concatenate(runs))
    want = e.scan_reduce_windows(10000000, 1, 18)
assert out[1] is None
assert np.array_equal(out[0][1], want[1]) and np.array_equal(out[0][2], want[2])
assert int(want[2].sum()) > 4.0e8
print("CHUNKED-OK")
'''
    p = subprocess.run([sys.executable, "-c", code, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, PANDEPTH_RCCL_LIB=shim))
    assert p.returncode == 0 and "CHUNKED-OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
