"""Narrow-window reductions from materialised depth (k_sweep<.., WIN, FROM_DEPTH>) and fused from the difference arrays, by window
width: where the time goes.  min_dep 0 so that every cell counts (the LDS traffic of a deep sample) on cheap, mostly empty arrays."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pandepth_amd.capi as pda

lens = np.array([250_000_000] * 8, dtype=np.uint32)          # 2.0e9 cells, 8 GB
rng = np.random.default_rng(1)
iv = np.stack([rng.integers(0, 8, 200000), rng.integers(0, 249_000_000, 200000), np.zeros(200000, dtype=np.int64)], axis=1).astype(np.int32)
iv[:, 2] = iv[:, 1] + 150
out = {}
with pda.Engine(lens) as e:
    e.push_intervals(iv)
    e.scan(0)
    e.profile(True)
    cells = int(lens.sum())
    for w in (4, 64, 100, 128, 1000, 4096, 8191, 8192, 100000):
        for md in (0, 1):
            e.profile(True)
            for _ in range(3):
                e.reduce_windows(w, md)
            ms, n = e.profile_get("reduce_windows")
            out["w%d_min%d" % (w, md)] = {"ms": round(ms / n, 4), "GBps": round(cells * 4 / (ms / n) / 1e6, 1), "frac": round(cells * 4 / (ms / n) / 1e6 / 8000, 3)}
            print("w %6d min_dep %d: %.3f ms  %.0f GB/s  %.3f" % (w, md, ms / n, cells * 4 / (ms / n) / 1e6, cells * 4 / (ms / n) / 1e6 / 8000), flush=True)
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "win_ab.json"), "w"), indent=1)
