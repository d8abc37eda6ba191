"""debug: full-size direct path at w=1000 vs the arrays path: where do the depth sums differ, and is it deterministic?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pandepth_amd as pda
from tools import synth
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
R = int(float(os.environ.get("R", "1e9")))
first, other = synth.gen_runs_torch(lens, R, dev, seed=4242)
first3, near, far = synth.gen_runs_torch(lens, R, dev, seed=4242, split=True)
del first3
torch.cuda.synchronize()
W = 1000
def load(direct, streams):
    eng.reset()
    eng.keep_deferred(bool(direct))
    for k, (t, dis) in enumerate(streams):
        last = k == len(streams) - 1
        eng.push_intervals_device(t.data_ptr(), int(t.shape[0]), pda.PD_PUSH_SORTED | (pda.PD_PUSH_DISORDER(dis) if dis else 0) | (pda.PD_PUSH_MORE if (direct or not last) else 0))
two = [(first, 0), (other, synth.MAX_SPAN)]
three = [(first, 0), (near, synth.NEAR_SPAN), (far, synth.MAX_SPAN)]
load(False, two)
woff, c_a, t_a = eng.scan_reduce_windows(W, 1, 0)
off = np.concatenate([[0], np.cumsum(((lens.astype(np.int64) + 8191) // 8192) * 8192)])
def report(name, c, t):
    bad = np.nonzero(t != t_a)[0]
    print(name, "sum diffs", bad.size, "cover diffs", int((c != c_a).sum()))
    for k in bad[:6]:
        tid = int(np.searchsorted(woff, k, side="right") - 1)
        wi = int(k - woff[tid])
        cell = int(off[tid]) + wi * W
        print("  window", int(k), "contig", tid, "win", wi, "cell", wi * W, "global cell", cell, "tile", cell // 8192, "+", cell % 8192, "got", int(t[k]), "want", int(t_a[k]))
for rep in range(3):
    load(True, three); _, c, t = eng.scan_reduce_windows(W, 1, 0); report("rep %d: 3 streams narrow" % rep, c, t)
    load(True, two); _, c, t = eng.scan_reduce_windows(W, 1, 0); report("rep %d: 2 streams narrow" % rep, c, t)
    load(True, three); eng.scan_reduce_windows(10000000, 1, 0)
    load(True, three); eng.scan_reduce_windows(10000000, 1, 18)
    load(True, three); _, c, t = eng.scan_reduce_windows(W, 1, 0); report("rep %d: 3 streams narrow after two wide calls" % rep, c, t)
# where is the run that lands wrong?  runs touching the first bad window's boundary
