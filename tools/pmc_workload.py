#!/usr/bin/env python3
"""tools/pmc_workload.py — the bench step (plus one write-back scan for calibration) without timing,
meant to run under `rocprofv3 --pmc ...` (see tools/pmc_collect.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pandepth_amd as pda  # noqa: E402
from tools import synth  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
dev = torch.device("cuda", 0)
names, lens = synth.genome_c2()
eng = pda.Engine(lens.astype(np.uint32), device=0)
first, other = synth.gen_runs_torch(lens, R, dev, seed=42)
torch.cuda.synchronize()
for it in range(2):
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
    eng.scan_reduce_windows(10000000, 1, 0)
    eng.scan(0)          # write-back sweep: reads AND writes every cell once = the calibration kernel
    eng.reduce_windows(100, 1)      # narrow windows off the depth (k_sweep<false, true, true>: LDS accumulators + k_window_edges)
    eng.synchronize()
# the direct window path (k_direct_tiles): both streams deferred, difference arrays never materialised
eng.keep_deferred(True)
for it in range(2):
    eng.reset()
    eng.push_intervals_device(first.data_ptr(), int(first.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE)
    eng.push_intervals_device(other.data_ptr(), int(other.shape[0]), pda.PD_PUSH_SORTED | pda.PD_PUSH_MORE | pda.PD_PUSH_DISORDER(synth.MAX_SPAN))
    eng.scan_reduce_windows(10000000, 1, 0)
    eng.synchronize()
# the same sample in the compact form (pd_runs_create): k_direct_c8, and its export instantiation (pd_export_i4)
runs8 = eng.runs_create(first.data_ptr(), int(first.shape[0]), other.data_ptr(), int(other.shape[0]))
n_cells, _ = eng.device_layout()
img = torch.zeros(n_cells // 2, dtype=torch.uint8, device=dev)
exc = torch.zeros((1 << 18, 2), dtype=torch.int64, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for it in range(2):
    eng.reset()
    eng.push_runs(runs8, pda.PD_PUSH_MORE)
    eng.scan_reduce_windows(10000000, 1, 0)
    eng.reset()
    eng.push_runs(runs8, pda.PD_PUSH_MORE)
    eng.export_i4(img.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr())
    eng.synchronize()
# the sliced sum's receiving side on the image just exported: one part, the whole genome (k_sweep_i4_wave)
n_tiles = n_cells // 8192
# (the tile sums the collective would all-reduce, here from the image itself — with zeros every tile would start at depth 0, the running depth
# would dip below zero and k_sweep_i4_fast would send every tile to the slow list: not the kernel a real exchange runs)
_t = img[:n_tiles * 4096].view(n_tiles, 4096)
sums = torch.zeros(n_tiles + 16, dtype=torch.int32, device=dev)
sums[:n_tiles] = (_t & 15).sum(1, dtype=torch.int32) + (_t >> 4).sum(1, dtype=torch.int32) - 8 * 8192
del _t
# ... plus what the cells outside the nibble's range (exported as exceptions, their nibble written as 8 = a difference of zero) really hold
_n = int(cnt[0].item())
if _n:
    _cells = exc[:_n, 0]
    _vals = (exc[:_n, 1] << 32) >> 32                      # pd_exc {uint64 cell; int32 value; int32 pad}: the low word, sign-extended
    sums.index_add_(0, (_cells // 8192).to(torch.int64), _vals.to(torch.int32))
part = torch.zeros(n_tiles * 24 + 64, dtype=torch.uint8, device=dev)
for it in range(2):
    eng.slice_sweep_i4(img.data_ptr(), 1, n_cells // 2, 0, n_tiles, sums.data_ptr(), exc.data_ptr(), 1 << 18, cnt.data_ptr(), 10000000, 1, 18, part.data_ptr())
    eng.synchronize()
# zlib's LZ77 parse on the device (pd_deflate_parse), 48 MB of per-site rows: the DEFAULT path of round 5 first — 8 KiB chunks with 2 KiB of
# overlap, sixteen to a workgroup with their text in LDS (k_lz_parse_lds, "lz_group" 16) — then round 4's default for comparison: 16 + 4 KiB
# chunks parsed with the text in memory (k_lz_parse, "lz_group" 0)
rows = b"".join(b"Chr01\t%d\t%d\n" % (j, 30 + (j * 2654435761 >> 7) % 23) for j in range(3200000))
def grid(CH, TAIL):
    return [(s0, min(len(rows), s0 + CH + TAIL), max(0, s0 - 32768)) for s0 in range(0, len(rows) - TAIL, CH)]
chunks = grid(8192, 2048)
for it in range(2):
    eng.deflate_parse(rows, chunks)
eng.set_param("lz_group", 0)
for it in range(2):
    eng.deflate_parse(rows, grid(16384, 4096))
eng.set_param("lz_group", 16)
eng.reset()
eng.runs_destroy(runs8)
eng.keep_deferred(False)
eng.reset()
print("cells", eng.device_buffer()[1], "runs", int(first.shape[0]) + int(other.shape[0]), "lz text", len(rows), "chunks", len(chunks))
