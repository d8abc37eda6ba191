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
    first, other = synth.gen_runs_torch(lens, int(1e9), dev, seed=4242)
    torch.cuda.synchronize()
    L = torch.from_numpy(lens.astype(np.int64)).to(dev)

    def mass(r):
        t = r[:, 0].long()
        b = torch.minimum(r[:, 1].long().clamp_min(0), L[t])
        e = torch.minimum(r[:, 2].long().clamp_min(0), L[t])
        return int((e - b).clamp_min(0).sum().item())
    yield {"eng": eng, "first": first, "other": other, "lens": lens, "mass": mass(first) + mass(other), "pda": pda,
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
