"""A genome of MORE than 2^32 cells (PanDepth is a plant-pangenome tool: per-contig arrays of any total size, PD:4129-4145): 20 chromosomes of
250 Mb + 300 scaffolds = 5.08e9 cells, 2e8 records, on the GPU through the C-ABI.  Flat cell indices pass 2^32 inside chromosome 18; the
compact form keeps 32 bits of a run's flat begin, so everything below must hold on BOTH sides of that line:

  * mass conservation (sum of TotalDepth == sum of clipped run lengths) on the atomic path, the owner-tile path and the compact / direct path
  * the three paths give identical tables (10 Mb bins with and without the 18-bit wrap; 1 kb windows)
  * 1 kb windows add up to the 10 Mb bins on contigs before, across and behind the 2^32nd cell
  * per-base depth against the CPU oracle on sampled scaffolds (the last one lies behind 5.0e9 cells) and on a stretch of the last chromosome
The executable on a generated 4.8 Gb / 2e8-record BAM (its decoder's compact session, and the 12-byte sessions of narrower windows) is
compared with the reference's output by SHA-256 (the reference ran builder-side on the identical file: profiles/r05_above4g_reference.json)."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
BIN = 10000000


@pytest.fixture(scope="module")
def big():
    import torch
    import pandepth_amd as pda
    from tools import synth
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    lens = np.array([250000000] * 20 + list(rng.integers(10000, 500001, 300)), dtype=np.int64)
    assert lens.sum() > (1 << 32) + 5e8
    eng = pda.Engine(lens.astype(np.uint32), device=0)
    first, near, far = synth.gen_runs_torch(lens, int(2e8), dev, seed=77, split=True)
    other = torch.cat([near, far], 0)
    other = other[torch.argsort((other[:, 0].long() << 32) | other[:, 1].long())].contiguous()
    del near, far
    torch.cuda.synchronize()
    L = torch.from_numpy(lens).to(dev)

    def mass(r):
        t = r[:, 0].long()
        b = torch.minimum(r[:, 1].long().clamp_min(0), L[t])
        e = torch.minimum(r[:, 2].long().clamp_min(0), L[t])
        return int((e - b).clamp_min(0).sum().item())
    yield {"eng": eng, "first": first, "other": other, "lens": lens, "mass": mass(first) + mass(other), "pda": pda, "synth": synth, "torch": torch}
    eng.close()


def test_above_4g_cells(big):
    import pd_oracle as O
    s = big
    eng, pda, synth, torch, lens = s["eng"], s["pda"], s["synth"], s["torch"], s["lens"]
    first, other = s["first"], s["other"]
    n_cells, _ = eng.device_layout()
    assert n_cells > (1 << 32)
    # ---- the atomic path: the yardstick ----
    eng.keep_deferred(False)
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_DEFAULT)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_DEFAULT)
    woff, cov_a, tot_a = eng.scan_reduce_windows(BIN, 1, 0)
    assert int(tot_a.sum()) == s["mass"]
    _, cov18, tot18 = eng.scan_reduce_windows(BIN, 1, 18)
    assert np.array_equal(cov18, cov_a) and np.array_equal(tot18, tot_a)
    w1, c1_a, t1_a = eng.scan_reduce_windows(1000, 1, 0)
    for t in (0, 16, 17, 18, 19, 20, len(lens) - 1):                     # 1 kb windows add up to the 10 Mb bins (17 holds the 2^32nd cell)
        n10, per = int(woff[t + 1] - woff[t]), BIN // 1000
        a = c1_a[w1[t]:w1[t + 1]].astype(np.int64); b = t1_a[w1[t]:w1[t + 1]].astype(np.int64)
        pad = n10 * per - a.size
        a = np.concatenate([a, np.zeros(pad, dtype=np.int64)]).reshape(n10, per).sum(1)
        b = np.concatenate([b, np.zeros(pad, dtype=np.int64)]).reshape(n10, per).sum(1)
        assert np.array_equal(a, cov_a[woff[t]:woff[t + 1]].astype(np.int64)) and np.array_equal(b, tot_a[woff[t]:woff[t + 1]].astype(np.int64)), t
    # ---- per-base depth against the oracle: three scaffolds (the last lies behind 5.0e9 cells) and 3 Mb of the last chromosome ----
    eng.scan(0)
    allr = torch.cat([first, other], 0)
    for t in (20, 170, len(lens) - 1):
        sel = allr[allr[:, 0] == t].cpu().numpy().copy()
        sel[:, 0] = 0
        d, off = O.depth_from_intervals([int(lens[t])], sel)
        assert np.array_equal(eng.read_depth(t, 0, int(lens[t])), d[off[0]:off[0] + int(lens[t])]), "contig %d" % t
    t, a, n = 19, 120000000, 3000000
    sel = allr[(allr[:, 0] == t) & (allr[:, 2] > a) & (allr[:, 1] < a + n)].cpu().numpy().astype(np.int64)
    sel[:, 0] = 0
    sel[:, 1] = np.clip(sel[:, 1] - a, 0, n); sel[:, 2] = np.clip(sel[:, 2] - a, 0, n)
    d, off = O.depth_from_intervals([n], sel.astype(np.int32))
    assert np.array_equal(eng.read_depth(t, a, n), d[off[0]:off[0] + n])
    del allr
    # ---- the owner-tile path (sorted pushes) ----
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
    _, cov_t, tot_t = eng.scan_reduce_windows(BIN, 1, 0)
    assert np.array_equal(cov_t, cov_a) and np.array_equal(tot_t, tot_a), "owner-tile path differs from the atomic path"
    _, c1_t, t1_t = eng.scan_reduce_windows(1000, 1, 0)
    assert np.array_equal(c1_t, c1_a) and np.array_equal(t1_t, t1_a)
    # ---- the compact sample (what the decoder leaves) through the direct kernel: no arrays ----
    eng.reset()
    runs = eng.runs_create(first.data_ptr(), int(first.shape[0]), other.data_ptr(), int(other.shape[0]))
    eng.keep_deferred(True)
    try:
        for wrap in (0, 18):
            eng.reset()
            eng.push_runs(runs, pda.PD_PUSH_MORE)
            _, cov, tot = eng.scan_reduce_windows(BIN, 1, wrap)
            assert int(tot.sum()) == s["mass"]
            bad = np.nonzero((cov != cov_a) | (tot != tot_a))[0]
            assert bad.size == 0, "compact / direct path (wrap %d) differs from the atomic path in %d bins, first %s" % (wrap, bad.size, bad[:8].tolist())
        # ... and expanded back to 12-byte runs for a mode that needs the arrays (narrow windows): the same 1 kb table
        eng.reset()
        eng.push_runs(runs, pda.PD_PUSH_MORE)
        _, c1, t1 = eng.scan_reduce_windows(1000, 1, 0)
        assert np.array_equal(c1, c1_a) and np.array_equal(t1, t1_a)
    finally:
        eng.keep_deferred(False)
        eng.reset()
        eng.runs_destroy(runs)


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def test_cli_above_4g_cells(tmp_path):
    """the executable end to end on a 4.8 Gb genome (bamgen -s 1.6: chromosomes of up to 533 Mb, flat cells up to 4.8e9) with 2e8 records:
    the decoder's compact session (mode 0, -w 10000) and its 12-byte sessions + arrays (-w 1000), SHA-256 against the reference's files"""
    want = json.load(open(os.path.join(ROOT, "profiles", "r05_above4g_reference.json")))
    free = os.statvfs(str(tmp_path)).f_bavail * os.statvfs(str(tmp_path)).f_frsize
    if free < 16e9:
        pytest.skip("needs 16 GB under %s for the generated BAM" % tmp_path)
    gen = os.path.join(ROOT, "tools", "bamgen")
    assert os.access(gen, os.X_OK), "tools/bamgen not built (__graft_entry__.build)"
    bam = str(tmp_path / "big48.bam")
    subprocess.run([gen, "-o", bam] + want["bamgen_args"], check=True, stderr=subprocess.PIPE, timeout=1800)
    assert os.path.getsize(bam) == want["bam_bytes"], "bamgen is deterministic: same arguments, same file"
    cli = os.path.join(ROOT, "pandepth_amd", "pandepth")
    try:
        for case in want["cases"]:
            out = str(tmp_path / ("o_" + case["name"]))
            p = subprocess.run([cli, "-i", bam] + case["args"] + ["-o", out, "-t", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                               env=dict(os.environ, PANDEPTH_TIMING="1"))
            assert p.returncode == 0, p.stderr.decode()[-600:]
            assert _sha(out + "." + case["file"]) == case["sha256"], "%s: %s differs from the reference's" % (case["name"], case["file"])
            if case["name"] in ("chr", "w10000"):
                assert "runs (compact session)" in p.stderr.decode(), "the compact session did not run above 2^32 cells"
            os.remove(out + "." + case["file"])
    finally:
        # 13.5 GB: the box's /tmp is what bench.py's full-size file has to fit into afterwards
        for f in (bam, bam + ".bai"):
            if os.path.exists(f):
                os.remove(f)
