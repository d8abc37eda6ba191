"""tools/ubench/inflate_ab.py <file.bam> — the wave inflate kernel alone (pd_x_bgzf_inflate) on the BGZF members of one file, for the
library named by PANDEPTH_AMD_LIB (tuning builds: tools/ubench/build_inflate_variants.sh) and the waves-per-CU settings in WAVES
(default "16").  Prints GB/s of inflated bytes per setting; the first 2000 members of the output are compared with zlib."""
import os
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pandepth_amd import capi  # noqa: E402

data = open(sys.argv[1], "rb").read()
cap = int(float(os.environ.get("MAX_BYTES", "1.2e9")))
if len(data) > cap:                       # whole members only
    o = 0
    while o + 18 <= cap:
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        if o + bs > cap:
            break
        o += bs
    data = data[:o]
name = os.path.basename(os.environ.get("PANDEPTH_AMD_LIB", "libpandepth_amd.so"))
for w in [int(x) for x in os.environ.get("WAVES", "16").split(",")]:
    variant = 2 + (w << 4)
    try:
        out, ms, nb, n = capi.bgzf_inflate(data, variant=variant, reps=int(os.environ.get("REPS", "5")), want_output=True)
    except Exception as ex:  # noqa: BLE001
        print("%-28s %2d waves/CU: FAILED %r" % (name, w, ex), flush=True)
        continue
    o = uo = 0
    bad = 0
    for k in range(min(nb, 2000)):
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        raw = zlib.decompress(data[o + 12 + xlen:o + bs - 8], -15)
        if bytes(out[uo:uo + len(raw)]) != raw:
            bad += 1
        uo += len(raw); o += bs
    print("%-28s %2d waves/CU: %d members, %.1f MB out, kernel %.3f ms = %.1f GB/s out (%.1f GB/s in), members differing from zlib: %d" % (
        name, w, nb, n / 1e6, ms, n / ms / 1e6, len(data) / ms / 1e6, bad), flush=True)
