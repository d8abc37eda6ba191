#!/usr/bin/env python3
"""The direct window path only (k_direct_tiles), for rocprofv3 --pmc passes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import pandepth_amd as pda
from tools import synth
R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
torch.cuda.synchronize()
eng.keep_deferred(True)
if len(sys.argv) > 2:
    eng.set_param("direct_un", int(sys.argv[2]))
for it in range(2):
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
    eng.scan_reduce_windows(10000000, 1, 0)
    eng.synchronize()
print("runs", int(first.shape[0]) + int(other.shape[0]))
