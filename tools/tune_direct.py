#!/usr/bin/env python3
"""Times the direct window path (k_direct_tiles) on the bench workload for several kernel variants
(pd_set_param "direct_un" = loads in flight per thread + 100 x waves-per-SIMD target) and grids."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandepth_amd as pda
from tools import synth
R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
torch.cuda.synchronize()
eng.keep_deferred(True)
out = {}
ref = None
for grid in (256,):
    for un in (504, 508, 404):
        eng.set_param("direct_un", un)
        eng.set_param("sample", grid)
        def step():
            eng.reset()
            eng.push_intervals_device(first.data_ptr(), first.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            eng.push_intervals_device(other.data_ptr(), other.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
            return eng.scan_reduce_windows(10000000, 1, 0)
        r = step()
        if ref is None:
            ref = r
        assert np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2])
        eng.profile(True)
        for _ in range(4):
            step()
        ms, n = eng.profile_get("direct_tiles")
        ms2, n2 = eng.profile_get("scatter_index")
        eng.profile(False)
        out["sample%d_un%d" % (grid, un)] = (round(ms / n, 3), round(ms2 / n * 1.0, 3))
        if grid and un not in (504, 404, 604):
            continue
print(json.dumps(out))
eng.close()
