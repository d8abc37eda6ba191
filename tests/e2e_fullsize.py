#!/usr/bin/env python3
"""tests/e2e_fullsize.py — the CLI on a FULL-SIZE reference (3.0 Gb, 512 contigs) with a thin read set:
checks the 3 Gb code paths end to end (12 GB context, index arrays, 3e6-window tables) against the
reference binary, byte for byte."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

R = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(3e6)
names, lens = synth.genome_c2()
rec = synth.gen_records_numpy(lens, R, seed=77)
td = tempfile.mkdtemp(prefix="pdfull", dir="/tmp")
bam = os.path.join(td, "f.bam")
synth.write_bam(bam, names, lens, rec, procs=16, payload=False)
subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], check=True)
cli, ref = os.path.join(ROOT, "pandepth_amd", "pandepth"), os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
# configs[2]'s annotation shape: 33 688 transcripts / 175 274 CDS rows, exon length log-normal
import numpy as np  # noqa: E402
rng = np.random.default_rng(3)
per_tx = np.full(33688, 175274 // 33688); per_tx[:175274 - per_tx.sum()] += 1
chrom = rng.choice(12, 33688, p=lens[:12] / lens[:12].sum())
gff = os.path.join(td, "c3.gff")
with open(gff, "w") as fh:
    for t in range(33688):
        c = int(chrom[t]); s0 = int(rng.integers(1, lens[c] - 200000))
        for e in range(int(per_tx[t])):
            el = int(min(5000, max(30, rng.lognormal(np.log(150), 0.7))))
            fh.write("%s\tsynth\tCDS\t%d\t%d\t.\t+\t0\tID=cds%d.%d;Parent=tx%05d\n" % (names[c], s0, s0 + el - 1, t, e, t))
            s0 += el + int(rng.integers(80, 3000))
# PD_E2E_TAGS: a subset of the cases, e.g. "w100,w1000" (w100: the 3e7-row, 1.3 GB table whose rows the engine formats and parses on the device)
only = [x for x in os.environ.get("PD_E2E_TAGS", "chr,w1000,chr_s,gff").split(",") if x]
for tag, extra, suffix in (("chr", [], "chr.stat.gz"), ("w1000", ["-w", "1000"], "win.stat.gz"), ("chr_s", ["-s"], "chr.stat.gz"),
                           ("gff", ["-g", gff], "gene.stat.gz"), ("w100", ["-w", "100"], "win.stat.gz")):
    if tag not in only:
        continue
    t0 = time.perf_counter()
    pm = subprocess.run([cli, "-i", bam, "-o", os.path.join(td, "m_" + tag), "-t", "16"] + extra, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=280,
                        env=dict(os.environ, PANDEPTH_TIMING="1"))
    for ln in pm.stderr.decode().splitlines():
        if any(k in ln for k in ("window table", "scan + ", "totals + table", "table gzip", "table close", "decode + scatter", "engine create")):
            print("      " + ln[:200], flush=True)
    t1 = time.perf_counter()
    subprocess.run([ref, "-i", bam, "-o", os.path.join(td, "r_" + tag), "-t", "16"] + extra, check=True, stdout=subprocess.DEVNULL, timeout=280)
    t2 = time.perf_counter()
    same = open(os.path.join(td, "m_%s.%s" % (tag, suffix)), "rb").read() == open(os.path.join(td, "r_%s.%s" % (tag, suffix)), "rb").read()
    print("%-6s genome %.2f Gb, %d records: pandepth %.2f s, pandepth_ref %.2f s, byte-identical %s" % (
        tag, lens.sum() / 1e9, R, t1 - t0, t2 - t1, same), flush=True)
