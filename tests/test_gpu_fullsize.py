"""Full-size (BASELINE.json configs[1]: 3.0 Gb reference, 1e9 records) parity through
size-independent properties, on the GPU through the C-ABI — the oracle cannot run at this size:

  * mass conservation: sum of TotalDepth over all bins == sum of clipped run lengths (exact, int64)
  * the owner-tile path and the atomic path give identical tables on the same runs
  * linearity: pushing the sample twice doubles every TotalDepth and leaves CoveredSite unchanged
  * window additivity: 1 kb windows add up to the 10 Mb bins; interval statistics over those
    windows' coordinates equal the window statistics; -d 3 never covers more than -d 1
  * the 18-bit wrap is the identity while depth < 2^18, and read-back cells add up to the bins
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = 10000000


@pytest.fixture(scope="module")
def sample():
    import torch
    import pandepth_amd as pda
    from tools import synth
    dev = torch.device("cuda", 0)
    names, lens = synth.genome_c2()
    eng = pda.Engine(lens.astype(np.uint32), device=0)
    # one generation for both forms of the sample (the generator's float cumsum is not bit-reproducible from call to call:
    # a second call may round one position differently): first runs + the later runs as the decoder's near / far streams,
    # and the same later runs as ONE stream in position order
    first, near, far = synth.gen_runs_torch(lens, int(1e9), dev, seed=4242, split=True)
    other = torch.cat([near, far], 0)
    other = other[torch.argsort((other[:, 0].long() << 32) | other[:, 1].long())].contiguous()
    torch.cuda.synchronize()
    L = torch.from_numpy(lens.astype(np.int64)).to(dev)

    def mass(r):
        t = r[:, 0].long()
        b = torch.minimum(r[:, 1].long().clamp_min(0), L[t])
        e = torch.minimum(r[:, 2].long().clamp_min(0), L[t])
        return int((e - b).clamp_min(0).sum().item())
    yield {"eng": eng, "first": first, "other": other, "near": near, "far": far, "lens": lens, "mass": mass(first) + mass(other), "pda": pda,
           "synth": synth, "torch": torch}
    eng.close()


def load(s, flags_sorted=True, times=1):
    eng, pda, synth = s["eng"], s["pda"], s["synth"]
    eng.reset()
    for _ in range(times):
        if flags_sorted:
            eng.push_intervals_device(s["first"].data_ptr(), int(s["first"].shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            eng.push_intervals_device(s["other"].data_ptr(), int(s["other"].shape[0]),
                                      pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
        else:
            eng.push_intervals_device(s["first"].data_ptr(), int(s["first"].shape[0]), pda.PD_PUSH_DEFAULT)
            eng.push_intervals_device(s["other"].data_ptr(), int(s["other"].shape[0]), pda.PD_PUSH_DEFAULT)


def test_fullsize_properties(sample):
    s = sample
    eng, lens = s["eng"], s["lens"]
    load(s)
    woff, cov, tot = eng.scan_reduce_windows(BIN, 1, 0)
    assert int(tot.sum()) == s["mass"]                                    # mass conservation
    assert int(cov.sum()) <= int(lens.sum()) and int(cov.sum()) > 0.9 * lens.sum()
    _, cov18, tot18 = eng.scan_reduce_windows(BIN, 1, 18)                  # wrap is the identity below 2^18
    assert np.array_equal(cov18, cov) and np.array_equal(tot18, tot)
    _, cov3, _ = eng.scan_reduce_windows(BIN, 3, 0)
    assert np.all(cov3 <= cov)
    # 1 kb windows add up to the 10 Mb bins
    w1off, c1, t1 = eng.scan_reduce_windows(1000, 1, 0)
    for t in (0, 5, 11, 12, len(lens) - 1):
        n10 = int(woff[t + 1] - woff[t])
        per = BIN // 1000
        a = c1[w1off[t]:w1off[t + 1]].astype(np.int64)
        b = t1[w1off[t]:w1off[t + 1]].astype(np.int64)
        pad = n10 * per - a.size
        a = np.concatenate([a, np.zeros(pad, dtype=np.int64)]).reshape(n10, per).sum(1)
        b = np.concatenate([b, np.zeros(pad, dtype=np.int64)]).reshape(n10, per).sum(1)
        assert np.array_equal(a, cov[woff[t]:woff[t + 1]].astype(np.int64))
        assert np.array_equal(b, tot[woff[t]:woff[t + 1]].astype(np.int64))
    # materialise the depth: interval statistics over window coordinates == window statistics
    eng.scan(0)
    _, c2, t2 = eng.reduce_windows(1000, 1)
    assert np.array_equal(c2, c1) and np.array_equal(t2, t1)
    rng = np.random.default_rng(1)
    ks = rng.integers(0, int(w1off[1]) - 1, 2000)
    regs = np.stack([np.zeros_like(ks), ks * 1000 + 1, (ks + 1) * 1000], axis=1).astype(np.int32)
    rc, rs = eng.reduce_intervals(regs, 1)
    assert np.array_equal(rc.astype(np.int64), c1[ks].astype(np.int64)) and np.array_equal(rs, t1[ks])
    cells = eng.read_depth(3, 1000000, 2000000)
    k0 = int(w1off[3]) + 1000
    assert np.array_equal(cells.reshape(-1, 1000).sum(1).astype(np.uint64), t1[k0:k0 + 2000])
    # the atomic path gives the same tables
    load(s, flags_sorted=False)
    _, cova, tota = eng.scan_reduce_windows(BIN, 1, 0)
    assert np.array_equal(cova, cov) and np.array_equal(tota, tot)
    # linearity
    load(s, times=2)
    _, covd, totd = eng.scan_reduce_windows(BIN, 1, 0)
    assert np.array_equal(covd, cov) and np.array_equal(totd, 2 * tot)


def load3(s):
    """three deferred streams (first / near / far runs): what the product's GPU decoder emits with "decode_near_span" set"""
    eng, pda, synth = s["eng"], s["pda"], s["synth"]
    eng.reset()
    eng.push_intervals_device(s["first"].data_ptr(), int(s["first"].shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(s["near"].data_ptr(), int(s["near"].shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.NEAR_SPAN))
    eng.push_intervals_device(s["far"].data_ptr(), int(s["far"].shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))


def test_fullsize_direct_path(sample):
    """The direct window path (k_direct_tiles: no difference arrays in HBM; what the pandepth executable runs in
    whole-chromosome mode) at full size: mass conservation, and the same tables as the atomic path, for 10 Mb bins,
    1 kb windows (narrow instantiation) and the 18-bit wrap."""
    s = sample
    eng = s["eng"]
    load(s, flags_sorted=False)
    eng.keep_deferred(False)
    _, cov_a, tot_a = eng.scan_reduce_windows(BIN, 1, 0)
    w1off, c1_a, t1_a = eng.scan_reduce_windows(1000, 1, 0)
    eng.keep_deferred(True)
    try:
        for w, wrap, ec, et in ((BIN, 0, cov_a, tot_a), (BIN, 18, cov_a, tot_a), (1000, 0, c1_a, t1_a)):
            load3(s)
            _, cov, tot = eng.scan_reduce_windows(w, 1, wrap)
            # the direct call read the sample and left it deferred: asking again gives the same tables
            _, cov_again, tot_again = eng.scan_reduce_windows(w, 1, wrap)
            assert np.array_equal(cov, cov_again) and np.array_equal(tot, tot_again)
            assert int(tot.sum()) == s["mass"]
            if not (np.array_equal(cov, ec) and np.array_equal(tot, et)):
                # which side is off?  a third computation (sorted pushes, arrays path) on the same engine
                eng.keep_deferred(False)
                load(s)
                _, c3, t3 = eng.scan_reduce_windows(w, 1, wrap)
                bad = np.nonzero(tot != et)[0]
                raise AssertionError("w=%d wrap=%d: %d windows differ (first %s: direct %s, atomic path %s, tile path %s); direct==tile %s, atomic==tile %s; "
                                     "atomic-path mass %d vs %d" % (w, wrap, bad.size, bad[:6].tolist(), tot[bad[:6]].tolist(), et[bad[:6]].tolist(), t3[bad[:6]].tolist(),
                                                                    bool(np.array_equal(tot, t3)), bool(np.array_equal(et, t3)), int(et.sum()), s["mass"]))
    finally:
        eng.keep_deferred(False)


def test_fullsize_compact_path(sample):
    """What the executable runs in whole-chromosome mode, at full size: the sample as ONE compact sample (pd_runs_create: 8-byte runs, the
    sorted stream in file order + the later runs bucketed; the GPU decoder leaves the same object) -> pd_push_runs -> k_direct_c8.
    Mass conservation; the same tables as the atomic path for 10 Mb bins with and without the 18-bit wrap; the call READS the sample
    (asking twice gives the same); the 4-bit image straight from the compact sample equals the image of the arrays."""
    s = sample
    eng, pda, torch = s["eng"], s["pda"], s["torch"]
    load(s, flags_sorted=False)
    eng.keep_deferred(False)
    _, cov_a, tot_a = eng.scan_reduce_windows(BIN, 1, 0)
    _, cov_a18, tot_a18 = eng.scan_reduce_windows(BIN, 1, 18)
    n_cells, n_sums = eng.device_layout()
    img_a = torch.empty(n_cells // 2, dtype=torch.uint8, device="cuda")
    exc = torch.zeros(1 << 18, 2, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    eng.export_i4(img_a.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr())
    torch.cuda.synchronize()
    n_exc_a = int(cnt[0].item())
    exc_a = exc[:n_exc_a].clone()
    eng.reset()
    runs = eng.runs_create(s["first"].data_ptr(), int(s["first"].shape[0]), s["other"].data_ptr(), int(s["other"].shape[0]))
    eng.keep_deferred(True)
    try:
        for wrap, ec, et in ((0, cov_a, tot_a), (18, cov_a18, tot_a18)):
            eng.reset()
            eng.push_runs(runs, pda.PD_PUSH_MORE)
            _, cov, tot = eng.scan_reduce_windows(BIN, 1, wrap)
            _, cov2, tot2 = eng.scan_reduce_windows(BIN, 1, wrap)
            assert np.array_equal(cov, cov2) and np.array_equal(tot, tot2)
            assert int(tot.sum()) == s["mass"]
            assert np.array_equal(cov, ec) and np.array_equal(tot, et), "compact path differs from the atomic path (wrap %d)" % wrap
        eng.reset()
        eng.push_runs(runs, pda.PD_PUSH_MORE)
        img_c = torch.empty(n_cells // 2, dtype=torch.uint8, device="cuda")
        exc.zero_(); cnt.zero_()
        eng.export_i4(img_c.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr())
        torch.cuda.synchronize()
        assert int(cnt[0].item()) == n_exc_a
        assert bool(torch.equal(img_c, img_a))
        ka = exc_a[torch.argsort(exc_a[:, 0])]; kc = exc[:n_exc_a][torch.argsort(exc[:n_exc_a, 0])]
        assert bool(torch.equal(ka, kc))
    finally:
        eng.keep_deferred(False)
        eng.reset()
        eng.runs_destroy(runs)


def test_fullsize_gff_config(sample):
    """configs[2] at full size: a synthetic annotation of 33 688 transcripts / 175 274 CDS entries (README:128) on the 1e9-record
    sample: pd_reduce_intervals against sums of pd_read_depth slices for sampled entries, and against window statistics for
    entries laid exactly on windows; -d 5 against a threshold applied to the same cells."""
    s = sample
    eng, lens = s["eng"], s["lens"]
    rng = np.random.default_rng(33688)
    n_tr, n_cds = 33688, 175274
    per = rng.multinomial(n_cds - n_tr, np.ones(n_tr) / n_tr) + 1
    tid = rng.choice(12, n_tr, p=lens[:12] / lens[:12].sum())
    regs = np.zeros((n_cds, 3), dtype=np.int32)
    k = 0
    for t in range(n_tr):
        start = int(rng.integers(1, lens[tid[t]] - 40000))
        for _ in range(per[t]):
            ln = int(np.clip(rng.lognormal(np.log(150), 0.7), 30, 3000))
            regs[k] = (tid[t], start, start + ln - 1)
            start += ln + int(rng.integers(50, 1500))
            k += 1
    load(s)
    eng.scan(18)
    c, t = eng.reduce_intervals(regs, 1)
    c5, t5 = eng.reduce_intervals(regs, 5)
    assert np.all(c5 <= c) and np.all(t5 <= t) and int(c.sum()) > 0
    for i in rng.choice(n_cds, 400, replace=False):
        tt, a, b = int(regs[i, 0]), int(regs[i, 1]), int(regs[i, 2])
        cells = eng.read_depth(tt, a - 1, b - a + 1).astype(np.int64)
        assert int(c[i]) == int((cells >= 1).sum()) and int(t[i]) == int(cells.sum())
        assert int(c5[i]) == int((cells >= 5).sum()) and int(t5[i]) == int(cells[cells >= 5].sum())
    # total over all entries == total over the same cells counted with multiplicity
    assert int(t.sum()) == sum(int(x) for x in t)


def test_fullsize_w100_config(sample):
    """configs[3] at full size: -w 100 fused (from the difference arrays) and from the depth arrays give the same 3.0e7
    windows; they add up to the -w 1000 windows and to the 10 Mb bins; the per-site read-back of a whole chromosome adds
    up to its windows (the -a stream is those cells, one line each)."""
    s = sample
    eng, lens = s["eng"], s["lens"]
    load(s)
    woff, cov, tot = eng.scan_reduce_windows(BIN, 1, 18)
    w1, c1, t1 = eng.scan_reduce_windows(1000, 1, 18)
    w100, c100, t100 = eng.scan_reduce_windows(100, 1, 18)
    eng.scan(18)
    _, d100, e100 = eng.reduce_windows(100, 1)
    assert np.array_equal(d100, c100) and np.array_equal(e100, t100)
    assert int(t100.sum()) == s["mass"] == int(tot.sum())
    for t in (0, 7, 11, 40):
        a = c100[w100[t]:w100[t + 1]].astype(np.int64); b = t100[w100[t]:w100[t + 1]].astype(np.int64)
        n1 = int(w1[t + 1] - w1[t])
        pad = n1 * 10 - a.size
        a = np.concatenate([a, np.zeros(pad, dtype=np.int64)]).reshape(n1, 10).sum(1)
        b = np.concatenate([b, np.zeros(pad, dtype=np.int64)]).reshape(n1, 10).sum(1)
        assert np.array_equal(a, c1[w1[t]:w1[t + 1]].astype(np.int64)) and np.array_equal(b, t1[w1[t]:w1[t + 1]].astype(np.int64))
    t = 7                                                                 # a whole chromosome's per-site cells (-a)
    cells = eng.read_depth(t, 0, int(lens[t])).astype(np.int64)
    n = cells.size // 100 * 100
    assert np.array_equal(cells[:n].reshape(-1, 100).sum(1), t100[w100[t]:w100[t] + n // 100].astype(np.int64))
    assert np.array_equal((cells[:n].reshape(-1, 100) >= 1).sum(1), c100[w100[t]:w100[t] + n // 100].astype(np.int64))
