"""A/B of the direct wide-window kernels on the bench sample (one process, one data set): per variant of the
"direct_un" knob the context's own event time of the direct_tiles section, and equality of the tables with the default's."""
import os, sys, time
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
first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
torch.cuda.synchronize()
eng.keep_deferred(True)
runs8 = eng.runs_create(first.data_ptr(), int(first.shape[0]), other.data_ptr(), int(other.shape[0]))   # the sample in the compact form (variants "c<un>")
compact = False
def scatter():
    eng.reset()
    if compact:
        eng.push_runs(runs8, pda.PD_PUSH_MORE)
        return
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN) | pda.PD_PUSH_MORE)
n_cells, _ = eng.device_layout()
img = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev)
exc = torch.zeros((1 << 18, 2), dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
img_ref = None
torch.cuda.synchronize()
CASES = [(10000000, 1, 0), (10000000, 1, 18), (10000000, 3, 0), (8192, 1, 0), (250000, 2, 18), (10000000, 0, 0)]
ref = {}
variants = [x for x in os.environ.get("VARIANTS", "0").split(",")]
for vs in variants:
    compact = vs.startswith("c")
    if compact: vs = vs[1:]
    v = int(vs.split("@")[0].split("/")[0])
    eng.set_param("grid_tiles", int(vs.split("@")[1]) if "@" in vs else 0)
    opts = vs.split("/")[1:]                                  # /s64 = direct_sample 64, /l160 = lmax 160
    eng.set_param("direct_sample", next((int(o[1:]) for o in opts if o[0] == "s"), 256))
    eng.set_param("lmax", next((int(o[1:]) for o in opts if o[0] == "l"), 512))
    eng.set_param("direct_un", v)
    ok = True
    for c in CASES:
        scatter()
        _, cov, tot = eng.scan_reduce_windows(*c)
        if c not in ref: ref[c] = (cov.copy(), tot.copy())
        elif not (np.array_equal(cov, ref[c][0]) and np.array_equal(tot, ref[c][1])):
            ok = False
            bad = np.nonzero((cov != ref[c][0]) | (tot != ref[c][1]))[0]
            print("variant", v, "case", c, "DIFFERS at", bad.size, "windows, first", bad[:5].tolist(), tot[bad[:3]].tolist(), ref[c][1][bad[:3]].tolist(), cov[bad[:3]].tolist(), ref[c][0][bad[:3]].tolist())
    for _ in range(2):
        scatter(); eng.scan_reduce_windows(10000000, 1, 0)
    eng.synchronize()
    eng.profile(True)
    t0 = time.perf_counter()
    N = 10
    for _ in range(N):
        scatter(); eng.scan_reduce_windows(10000000, 1, 0)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    ms, n = eng.profile_get("direct_tiles")
    ims, inn = eng.profile_get("scatter_index")
    eng.profile(False)
    print("variant %s%5d %s grid %s: direct_tiles %.3f ms/launch (%d), index %.3f ms/step, step wall %.3f ms, tables %s" % ("compact " if compact else "", v, "/".join(opts), vs.split("@")[1] if "@" in vs else "auto", ms / max(n, 1), n, ims / N, dt, "equal" if ok else "DIFFERENT"), flush=True)
    if os.environ.get("EXPORT", "1") == "1":
        # the export instantiation (pd_export_i4 on the deferred sample: what a rank of the multi-GPU sum runs)
        scatter(); eng.export_i4(img.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr()); eng.synchronize()
        if img_ref is None: img_ref = (img.clone(), int(cnt.item()))
        same = bool(torch.equal(img, img_ref[0])) and int(cnt.item()) == img_ref[1]
        eng.profile(True)
        for _ in range(5):
            scatter(); eng.export_i4(img.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr())
        eng.synchronize()
        ems, en = eng.profile_get("direct_export")
        eng.profile(False)
        print("        export: direct_export %.3f ms/launch (%d), image %s (%d exceptions)" % (ems / max(en, 1), en, "equal" if same else "DIFFERENT", int(cnt.item())), flush=True)
