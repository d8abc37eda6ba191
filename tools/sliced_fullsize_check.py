#!/usr/bin/env python3
"""Full-scale cross-check of the 4-bit image and the fused slice sweep against the context's own arrays."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pandepth_amd as pda
from pandepth_amd import multi
from tools import synth
TILE = 8192
R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
torch.cuda.synchronize()
eng.reset()
eng.push_intervals_device(first.data_ptr(), first.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
eng.push_intervals_device(other.data_ptr(), other.shape[0], pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
woff, cov1, tot1 = eng.scan_reduce_windows(10000000, 1, 18)
n_cells, n_sums = eng.device_layout()
n_tiles = n_cells // TILE
B = 1 << 16
send = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev)
exc = torch.zeros((B, 2), dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()          # torch's fills run on torch's stream, the export on the context's
eng.export_i4(send.data_ptr(), exc.data_ptr(), B, cnt.data_ptr())
eng.synchronize()
n_exc = int(cnt.item())
print("exceptions", n_exc, flush=True)
view = multi.buffer_view(eng, dev)
cells = view[:n_cells]
bad_total = 0
CH = 1 << 28
for o in range(0, n_cells, CH):
    n = min(CH, n_cells - o)
    b = send[o // 2:(o + n) // 2]
    img = torch.stack([(b & 0xf).to(torch.int32) - 8, (b >> 4).to(torch.int32) - 8], dim=1).reshape(-1)
    c = cells[o:o + n]
    inr = (c >= -8) & (c <= 7)
    bad = ((img != c) & inr) | ((img != 0) & ~inr)
    nb = int(bad.sum().item())
    if nb and bad_total == 0:
        idx = torch.nonzero(bad)[:8].flatten() + o
        print("first bad cells", idx.tolist(), "cells", cells[idx].tolist(), "img", img[idx - o].tolist(), flush=True)
    bad_total += nb
print("image mismatches", bad_total, "out-of-range cells", flush=True)
sums = view[n_cells:].clone()
meta = torch.cat([sums, torch.tensor([min(n_exc, B)], dtype=torch.int32, device=dev)])
part = torch.zeros(n_tiles * 24, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.slice_sweep_i4(send.data_ptr(), 1, n_cells // 2, 0, n_tiles, meta.data_ptr(), exc.data_ptr(), B,
                   meta.data_ptr() + 4 * n_sums, 10000000, 1, 18, part.data_ptr())
eng.synchronize()
_, cov2, tot2 = eng.gather_windows(part.data_ptr(), 10000000)
print("windows", len(tot1), "cover equal", np.array_equal(cov1, cov2), "sum equal", np.array_equal(tot1, tot2), flush=True)
if not np.array_equal(tot1, tot2):
    d = np.nonzero(tot1 != tot2)[0]
    print("first differing windows", d[:10], tot1[d[:5]], tot2[d[:5]])
    print("total", int(tot1.sum()), int(tot2.sum()))

# SlicedSum itself at full scale, without and with a 1-rank RCCL group
for mode in ("engine", "sync"):
    ss = multi.SlicedSum(eng, dev, stream_mode=mode)
    r = ss.run(10000000, 1, 18, 0)
    print("SlicedSum no-group", mode, np.array_equal(r[1], cov1), np.array_equal(r[2], tot1), flush=True)
    del ss
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
for mode in ("engine", "sync"):
    ss = multi.SlicedSum(eng, dev, stream_mode=mode, self_via_collective=True)
    r = ss.run(10000000, 1, 18, 0)
    print("SlicedSum rccl-1", mode, np.array_equal(r[1], cov1), np.array_equal(r[2], tot1), int(r[2].sum()), flush=True)
    s = ss.slots[0]
    print("  recv == send:", bool(torch.equal(s["recv"], s["send"])), "meta sums == ctx sums:", bool(torch.equal(s["meta"][:ss.n_sums], ss.sums)), flush=True)
    del ss
dist.destroy_process_group()
