"""The multi-sample sum with RCCL issued inside the library (include/pandepth_amd.h: pd_comm_*, pd_sliced_sum_start /
_finish, pd_sliced_window_sum — the successor of SURVEY §8(b)'s pd_allreduce_diff) against the depth oracle:
one rank on any box; two ranks as two processes, and the executable's `#.list` mode over two GPUs, where the box has them."""
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
MANIFEST = json.load(open(os.path.join(HERE, "golden", "manifest.json")))

CHILD = r'''
import os, sys, time, numpy as np
root, rank, world, idfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import pandepth_amd as pda
from test_gpu_engine import LENS, rand_intervals, oracle_depth, windows_ref
if rank == 0:
    uid = pda.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f: f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120: raise SystemExit("no unique id")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
K = 3                                                     # samples per rank, two in flight
def sample(r, k):
    rng = np.random.default_rng(1000 + 10 * r + k)
    iv = rand_intervals(rng, LENS, 40000)
    # pile-ups beyond the 4-bit range (exceptions), one across a tile edge
    return np.concatenate([iv, np.tile(np.array([[0, 10 + k + r, 50]], dtype=np.int32), (500, 1)),
                           np.tile(np.array([[0, 8190, 8200 + r]], dtype=np.int32), (40, 1))])
with pda.Engine(LENS, device=rank) as e:
    c = pda.Comm(e, uid, rank, world)
    res = []
    for k in range(K):                                    # software pipeline: start(k), then finish(k - 1)
        e.reset(); e.push_intervals(sample(rank, k)); c.start(k % 2)
        if k: res.append(c.finish((k - 1) % 2, 10000, 1, 18, 0))
    res.append(c.finish((K - 1) % 2, 10000, 1, 18, 0))
    # the deferred form: sorted batches stay pending, pd_export_i4 packs the tile windows straight from LDS
    e.reset(); e.keep_deferred(True)
    iv = sample(rank, 7); iv = iv[np.lexsort((iv[:, 1], iv[:, 0]))]
    e.push_intervals(iv, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    res.append(c.run(8192, 2, 18, 0))
    if rank == 0:
        for k, got in enumerate(res):
            kk, w, md = (k, 10000, 1) if k < K else (7, 8192, 2)
            parts = [sample(r, kk) for r in range(world)]
            d, off = oracle_depth(LENS, np.concatenate(parts), True)
            cov, tot = windows_ref(LENS, d, off, w, md)
            assert np.array_equal(got[1], cov) and np.array_equal(got[2], tot), k
    else:
        assert all(r is None for r in res)
    c.close()
print("COMM-OK", rank)
'''


def _run_ranks(world, tmp_path):
    idfile = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(r), str(world), idfile], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "COMM-OK %d" % r in so, so[-1500:] + se[-3000:]


def test_comm_one_rank(tmp_path):
    """world = 1: export, events between the context's stream and the communicator's, all-reduce / all-gather, sweep, gather."""
    _run_ranks(1, tmp_path)


def _n_gpus():
    import pandepth_amd as pda
    import ctypes
    n = ctypes.c_int(0)
    return n.value if pda.load().pd_device_count(ctypes.byref(n)) == 0 else 0


def test_comm_two_ranks_two_processes(tmp_path):
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs")
    _run_ranks(2, tmp_path)


LIST_WIDE = [e for e in MANIFEST if ".list" in e["args"][1] and e["fixture"] in ("f1", "f2") and
             not any(a in e["args"] for a in ("-g", "-b", "-w", "-a"))]


@pytest.mark.parametrize("transport", ["peer", "rccl"])
@pytest.mark.parametrize("case", LIST_WIDE, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_cli_list_mode_sums_through_the_communicator(case, transport, tmp_path):
    """`#.list` inputs in whole-chromosome mode: one context per GPU (all of them when the box has several; one, with
    comm=force in PANDEPTH_TUNE, otherwise), statistics through the in-process communicator (default) or REAL RCCL made ahead of
    the contexts (-X transport=rccl: pd_comm_preinit + pd_comm_init_all) + pd_sliced_window_sum — same bytes as the reference, and
    the same stdout: RCCL's version banner must not reach it and none of our lines may be lost."""
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, PANDEPTH_TIMING="1", PANDEPTH_TUNE="comm=force" + (",transport=rccl" if transport == "rccl" else ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([CLI] + case["args"] + ["-o", str(tmp_path / "o"), "-t", "4"], cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=180, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-800:]
    assert (b"in slices (RCCL)" if transport == "rccl" else b"in slices (in-process peer copies)") in p.stderr, p.stderr.decode()[-800:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gzip.decompress(gz)).hexdigest() == meta["text_sha256"], suffix
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix + " (gz bytes)"
