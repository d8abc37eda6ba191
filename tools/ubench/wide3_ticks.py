"""Per-phase shader-clock share of k_direct_wide3 on the bench sample, from a -DPD_WIDE3_TICKS build of the library
(tools/ubench/libpandepth_ticks.so; see tools/ubench/wide3_ticks.sh)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pandepth_amd.capi as capi
capi.lib_path = lambda: os.path.join(ROOT, "tools", "ubench", "libpandepth_ticks.so")
import torch
import pandepth_amd as pda
from tools import synth
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
first, other = synth.gen_runs_torch(lens, int(1e9), dev, seed=42)
torch.cuda.synchronize()
eng.keep_deferred(True)
L = capi.load()
L.pd_x_wide3_ticks.restype = ctypes.c_int
L.pd_x_wide3_ticks.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
NAMES = ["bounds + zero window", "barrier 1", "candidates", "barrier 2", "window read + scans", "barrier 3", "statistics", "barrier 4 + partial store"]
for v in [int(x) for x in os.environ.get("VARIANTS", "0,3504").split(",")]:
    eng.set_param("direct_un", v)
    for it in range(3):
        eng.reset()
        eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
        eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN) | pda.PD_PUSH_MORE)
        if it == 1: L.pd_x_wide3_ticks(buf)          # discard the warm-up
        eng.scan_reduce_windows(10000000, 1, 0)
    L.pd_x_wide3_ticks(buf)
    t = np.array(buf[:8], dtype=np.float64); tot = t.sum()
    print("variant %d: cycles per tile and wave %.0f (2 launches)" % (v, tot / 2 / 366535 / 4))
    for n, x in zip(NAMES, t): print("   %-28s %5.1f %%" % (n, 100 * x / tot))
