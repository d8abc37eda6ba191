#!/usr/bin/env python3
"""tools/bgzf_gpu_bench.py — GPU BGZF inflate (pd_x_bgzf_inflate) on a generated payload BAM:
checks the first blocks against zlib and reports kernel GB/s for both table placements."""
import os
import struct
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402
from pandepth_amd import capi  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(2e6)
names, lens = synth.genome_c2(scale=R / 1e9)
rec = synth.gen_records_numpy(lens, R, seed=11)
td = tempfile.mkdtemp(prefix="pdbgzf", dir="/tmp")
bam = os.path.join(td, "p.bam")
synth.write_bam(bam, names, lens, rec, procs=16, payload=True, level=int(os.environ.get("BGZF_LEVEL", "6")))
data = open(bam, "rb").read()
print("records %d, BGZF %.1f MB" % (R, len(data) / 1e6), flush=True)
VARIANTS = [(1, "lane per block, tables in global memory")] + [
    (2 + (w << 4), "wave per block, %d waves/CU" % w) for w in (8, 12, 14, 16, 20)] + [
    (2 + (16 << 4) + 0x8000, "wave per block, 16 waves/CU, CRC-32 check off")]
if os.environ.get("BGZF_VARIANTS"):
    keep = set(int(x) for x in os.environ["BGZF_VARIANTS"].split(","))
    VARIANTS = [v for v in VARIANTS if v[0] in keep]
for variant, name in VARIANTS:
    t0 = time.perf_counter()
    out, ms, nb, n = capi.bgzf_inflate(data, variant=variant, reps=3, want_output=True)
    wall = time.perf_counter() - t0
    # verify the first 3000 blocks against zlib
    o = uo = 0; bad = 0
    for k in range(min(nb, 3000)):
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        raw = zlib.decompress(data[o + 12 + xlen:o + bs - 8], -15)
        if out[uo:uo + len(raw)] != raw:
            bad += 1
        uo += len(raw); o += bs
    print("%-40s %d blocks, %.1f MB out: kernel %.2f ms = %.1f GB/s out (%.1f GB/s in), verified-bad %d, call wall %.2f s" % (
        name, nb, n / 1e6, ms, n / ms / 1e6, len(data) / ms / 1e6, bad, wall), flush=True)
