#!/usr/bin/env python3
"""Times the receiving side of the sliced multi-GPU sum (pd_slice_sweep_i4) on ONE GPU for the slice
a rank would own in a world of N: N nibble images of 1/N of the C2 genome's tiles (random content —
the kernel's work does not depend on the values), plus pd_export_i4 of a full context."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandepth_amd as pda          # noqa: E402
from tools import synth             # noqa: E402

TILE = 8192
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
n_cells, n_sums = eng.device_layout()
n_tiles = n_cells // TILE
sums = torch.zeros(n_sums, dtype=torch.int32, device=dev)
out = {}
for world in (1, 2, 4, 8):
    st = -(-n_tiles // world)
    sb = st * (TILE // 2)
    recv = torch.randint(0, 256, (world * sb,), dtype=torch.uint8, device=dev)
    part = torch.zeros(st * 24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(5):
        eng.slice_sweep_i4(recv.data_ptr(), world, sb, 0, st, sums.data_ptr(), 0, 0, 0, 10000000, 1, 18, part.data_ptr())
    eng.synchronize()
    ms, n = eng.profile_get("slice_sweep")
    eng.profile(False)
    out["slice_sweep_world%d" % world] = {"avg_ms": round(ms / n, 4), "tiles": st, "bytes_read": world * sb,
                                          "GB/s": round(world * sb / (ms / n * 1e-3) / 1e9, 1)}
    del recv, part
print(json.dumps(out))
eng.close()
