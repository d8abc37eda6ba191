#!/usr/bin/env python3
"""tests/e2e_bench.py — END-TO-END wall clock of the CLI on a generated BAM (GPU box): host BAM
decode + PCIe + kernels + table writer, next to the reference binary on the same file, with a
byte comparison of the outputs.  Numbers from here go to DESIGN.md (they are NOT bench.py's value)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402


def main():
    R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(5e7)
    modes = sys.argv[2:] or ["chr"]
    ncpu = os.cpu_count()
    names, lens = synth.genome_c2(scale=R / 1e9)
    t0 = time.perf_counter()
    rec = synth.gen_records_numpy(lens, R, seed=99)
    td = tempfile.mkdtemp(prefix="pde2e", dir="/tmp")
    bam = os.path.join(td, "s.bam")
    synth.write_bam(bam, names, lens, rec, procs=min(96, ncpu), level=1, payload=True)
    t1 = time.perf_counter()
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], check=True)
    t2 = time.perf_counter()
    print("records %d genome %.1f Mb bam %.2f GB  (generate %.1f s, index %.1f s)" % (
        R, lens.sum() / 1e6, os.path.getsize(bam) / 1e9, t1 - t0, t2 - t1), flush=True)
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    cli = os.path.join(ROOT, "pandepth_amd", "pandepth")
    extra = {"chr": [], "w100": ["-w", "100"], "w1000": ["-w", "1000"], "s": ["-s"], "w100a": ["-w", "100", "-a"]}
    if os.environ.get("E2E_DECODE_ONLY"):
        for t in (8, 16, 32, 64, 128):
            subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "mine"), "-t", str(t)], check=True,
                           stdout=subprocess.DEVNULL, env=dict(os.environ, PANDEPTH_TIMING="1", PANDEPTH_TUNE="decode_only=1"))
        return
    if os.environ.get("E2E_DEVICE_DECODE"):
        for bmb, t in (("2048", 2), ("1024", 16), ("2048", 16), ("2048", 32)):
            best = 1e9
            for rep in range(2):
                a = time.perf_counter()
                subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "dd"), "-t", str(t)], check=True, stdout=subprocess.DEVNULL,
                               env=dict(os.environ, PANDEPTH_TUNE="device_decode=1,dd_batch_mb=" + bmb,
                                        **({"PANDEPTH_TIMING": "1"} if rep == 1 else {})))
                best = min(best, time.perf_counter() - a)
            print("pandepth(MI355X, device decode) batch %s MB -t %d  %.2f s  %.3e records/s" % (bmb, t, best, R / best), flush=True)
        subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "hd"), "-t", "32"], check=True, stdout=subprocess.DEVNULL)
        print("device-decode output == host-decode output:",
              open(os.path.join(td, "dd.chr.stat.gz"), "rb").read() == open(os.path.join(td, "hd.chr.stat.gz"), "rb").read(), flush=True)
        return
    for mode in modes:
        suffix = "chr.stat.gz" if mode in ("chr", "s") else "win.stat.gz"
        for t in ([8, 32, 64, 128] if mode == "chr" else [16]):
            if True:
                subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "mine"), "-t", str(t)] + extra[mode],
                               check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, PANDEPTH_TIMING="1"))
            best = 1e9
            for _ in range(2):
                a = time.perf_counter()
                subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "mine"), "-t", str(t)] + extra[mode],
                               check=True, stdout=subprocess.DEVNULL)
                best = min(best, time.perf_counter() - a)
            print("pandepth(MI355X) %-6s -t %-3d  %.2f s  %.3e records/s" % (mode, t, best, R / best), flush=True)
        if os.access(ref, os.X_OK):
            best = 1e9
            for _ in range(2):
                a = time.perf_counter()
                subprocess.run([ref, "-i", bam, "-o", os.path.join(td, "ref"), "-t", "36"] + extra[mode], check=True,
                               stdout=subprocess.DEVNULL)
                best = min(best, time.perf_counter() - a)
            print("pandepth_ref(CPU)  %-6s -t 36   %.2f s  %.3e records/s" % (mode, best, R / best), flush=True)
            same = open(os.path.join(td, "mine." + suffix), "rb").read() == open(os.path.join(td, "ref." + suffix), "rb").read()
            print("outputs byte-identical: %s" % same, flush=True)
            if "-a" in extra[mode]:
                import hashlib
                hs = []
                for who in ("mine", "ref"):
                    h = hashlib.md5()
                    pz = subprocess.Popen(["gzip", "-dc", os.path.join(td, who + ".SiteDepth.gz")], stdout=subprocess.PIPE)
                    for blk in iter(lambda: pz.stdout.read(1 << 22), b""):
                        h.update(blk)
                    pz.wait()
                    hs.append(h.hexdigest())
                gz_same = subprocess.run(["cmp", "-s", os.path.join(td, "mine.SiteDepth.gz"), os.path.join(td, "ref.SiteDepth.gz")]).returncode == 0
                print("SiteDepth.gz sizes %d / %d bytes, decompressed content identical: %s, .gz bytes identical: %s" % (
                    os.path.getsize(os.path.join(td, "mine.SiteDepth.gz")), os.path.getsize(os.path.join(td, "ref.SiteDepth.gz")),
                    hs[0] == hs[1], gz_same), flush=True)


if __name__ == "__main__":
    main()
