"""pd_export_i4 on a deferred sample at 1e8 and 1e9 records: reports whether the direct export was taken (no scatter_tiles / export_i4 launches) and its time."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pandepth_amd as pda
from tools import synth
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
for R in (int(1e8), int(1e9)):
    eng = pda.Engine(lens.astype(np.uint32), device=0)
    first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
    n_cells, n_sums = eng.device_layout()
    img = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev); exc = torch.zeros((1 << 18, 2), dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.keep_deferred(True)
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), first.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), other.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
    eng.profile(True)
    eng.export_i4(img.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr())
    eng.synchronize()
    print("R", R, "exceptions", int(cnt.item()), "msg:", eng.L.pd_strerror(eng.h), {k: eng.profile_get(k) for k in ("direct_export", "scatter_tiles", "export_i4")}, flush=True)
    eng.close(); del first, other, img
