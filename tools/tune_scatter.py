#!/usr/bin/env python3
"""tools/tune_scatter.py — sweep the owner-tile scatter's knobs on the bench workload (GPU box)."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pandepth_amd as pda  # noqa: E402
from tools import synth  # noqa: E402


def main():
    R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
    dev = torch.device("cuda", 0)
    names, lens = synth.genome_c2()
    eng = pda.Engine(lens.astype(np.uint32), device=0)
    first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
    torch.cuda.synchronize()
    nf, no = int(first.shape[0]), int(other.shape[0])

    def run(tag, **params):
        for k, v in params.items():
            eng.set_param(k, v)
        for it in range(3):
            if it == 1:
                eng.profile(True)
            eng.reset()
            eng.push_intervals_device(first.data_ptr(), nf, pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
            eng.push_intervals_device(other.data_ptr(), no, pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
            eng.synchronize()
        ms_idx, n_idx = eng.profile_get("scatter_index")
        ms_t, n_t = eng.profile_get("scatter_tiles")
        ms_f, _ = eng.profile_get("scatter_finish")
        eng.profile(False)
        print("%-44s tiles %.3f ms/step (index %.3f finish %.3f) launches/step %d" % (
            tag, ms_t / 2, ms_idx / 2, ms_f / 2, n_t // 2), flush=True)

    for st in (4096, 8192):
        for grid in (2048, 8192, 65536, 1 << 20):
            run("stile=%d grid=%d" % (st, grid), scatter_tile=st, grid_tiles=grid, sample=64, lmax=512)
    for sample, lmax in ((32, 512), (128, 512), (64, 256)):
        run("stile=4096 grid=1M sample=%d lmax=%d" % (sample, lmax), scatter_tile=4096, grid_tiles=1 << 20,
            sample=sample, lmax=lmax)


if __name__ == "__main__":
    main()
