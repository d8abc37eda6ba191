"""The write-back sweep (pd_scan) and the window sweeps on 2.0e9 cells: ms and fraction of 8 TB/s (8 B per cell written back, 4 B read only)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pandepth_amd.capi as pda
lens = np.array([250_000_000] * 8, dtype=np.uint32)
rng = np.random.default_rng(1)
iv = np.stack([rng.integers(0, 8, 2000000), rng.integers(0, 249_000_000, 2000000), np.zeros(2000000, dtype=np.int64)], axis=1).astype(np.int32)
iv[:, 2] = iv[:, 1] + 150
cells = int(lens.sum())
with pda.Engine(lens) as e:
    for rep in range(3):
        e.reset(); e.push_intervals(iv); e.synchronize()
        e.profile(True)
        e.scan(0)
        ms, n = e.profile_get("scan")
        print("write-back sweep: %.3f ms  %.0f GB/s  %.3f" % (ms / n, cells * 8 / (ms / n) / 1e6, cells * 8 / (ms / n) / 1e6 / 8000), flush=True)
    for w in (100, 1000, 100000):
        e.profile(True)
        for _ in range(3): e.reduce_windows(w, 0)
        ms, n = e.profile_get("reduce_windows")
        print("reduce_windows w %d: %.3f ms  %.3f" % (w, ms / n, cells * 4 / (ms / n) / 1e6 / 8000), flush=True)
    for w in (100, 100000):
        for rep in range(2):
            e.reset(); e.push_intervals(iv); e.synchronize()
            e.profile(True)
            e.scan_reduce_windows(w, 0, 0)
            ms, n = e.profile_get("scan_reduce_windows")
            print("scan_reduce_windows w %d: %.3f ms  %.3f" % (w, ms / n, cells * 4 / (ms / n) / 1e6 / 8000), flush=True)
