#!/usr/bin/env python3
"""Generate the parity fixtures and golden outputs under tests/golden/.

TEST INFRASTRUCTURE.  Runs only in the dev container, where the reference checkout is mounted
at /root/reference and `make -C oracle ref` has produced

    oracle/_ref/pandepth_ref   the reference CLI compiled from its own sources
    oracle/_ref/sam2bam        SAM -> BAM+BAI helper linked against the reference's libhts.a

What is committed is data only: seeded synthetic inputs (SAM/BAM/BAI/GFF/GTF/BED/list) and the
reference's outputs on them (the `.stat.gz` / `.SiteDepth.gz` files, byte for byte, plus a
manifest with sha256 of the gz bytes and of the decompressed text).  No reference source is
copied.  Re-running this script must reproduce the committed files bit for bit.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py f5         # regenerate the named fixtures only
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
SAM2BAM = os.path.join(ROOT, "oracle", "_ref", "sam2bam")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def write(path, text):
    with open(path, "w") as f:
        f.write(text)


def cigar_ref_span(cig):
    n, span = "", 0
    for ch in cig:
        if ch.isdigit():
            n += ch
        else:
            if ch in "MDN=X":
                span += int(n)
            n = ""
    return span


def sam_header(contigs, sorted_=True):
    h = "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if sorted_ else "unsorted")
    for name, ln in contigs:
        h += "@SQ\tSN:%s\tLN:%d\n" % (name, ln)
    return h


def sam_line(i, flag, rname, pos1, mapq, cigar):
    return "r%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t*\t*\n" % (i, flag, rname, pos1, mapq, cigar)


def to_bam(sam, bam, index=True):
    cmd = [SAM2BAM, sam, bam] + ([] if index else ["noindex"])
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


# ---------------------------------------------------------------------------------------------
# F1: small multi-mode fixture
# ---------------------------------------------------------------------------------------------
F1_CONTIGS = [("chrA", 1001), ("chrB", 500), ("chrC", 1)]
F1_CIGARS = ["50M", "20M5D30M", "10S40M", "25M100N25M", "30M2I18M", "50=", "20=1X29=", "5H45M3S",
             "2D48M", "10M1D10M1D10M1D20M"]


def f1_reads(seed, n=600):
    rng = random.Random(seed)
    recs = []
    for i in range(n):
        ci = 0 if rng.random() < 0.6 else 1
        name, ln = F1_CONTIGS[ci]
        cig = rng.choice(F1_CIGARS)
        span = cigar_ref_span(cig)
        # allow a small overhang past the contig end (reference pads its arrays); keep < 90
        pos1 = rng.randint(1, ln - span + 60) if ln - span + 60 >= 1 else 1
        pos1 = max(1, pos1)
        flag = rng.choice([0, 0, 0, 16, 16, 256, 1024, 512, 99, 147, 2048])
        mapq = rng.choice([0, 10, 20, 60, 60])
        if rng.random() < 0.03:      # unmapped read placed at its mate's position (no CIGAR)
            flag, cig = 4 | 1 | 64, "*"
        recs.append((ci, pos1, flag, mapq, cig))
    return recs


def build_f1(d):
    os.makedirs(d, exist_ok=True)
    recs = f1_reads(42)
    srt = sorted(range(len(recs)), key=lambda k: (recs[k][0], recs[k][1], k))
    sam = sam_header(F1_CONTIGS, True)
    for k in srt:
        ci, pos1, flag, mapq, cig = recs[k]
        sam += sam_line(k, flag, F1_CONTIGS[ci][0], pos1, mapq, cig)
    write(os.path.join(d, "f1.sam"), sam)
    to_bam(os.path.join(d, "f1.sam"), os.path.join(d, "f1.bam"), True)
    # the same records with no index next to them (forces the sorted no-index stream)
    shutil.copy(os.path.join(d, "f1.bam"), os.path.join(d, "f1_noidx.bam"))
    # unsorted: shuffled order, header says unsorted
    rng = random.Random(7)
    order = list(range(len(recs)))
    rng.shuffle(order)
    usam = sam_header(F1_CONTIGS, False)
    for k in order:
        ci, pos1, flag, mapq, cig = recs[k]
        usam += sam_line(k, flag, F1_CONTIGS[ci][0], pos1, mapq, cig)
    write(os.path.join(d, "f1_unsorted.sam"), usam)
    to_bam(os.path.join(d, "f1_unsorted.sam"), os.path.join(d, "f1_unsorted.bam"), False)
    # two more samples for the list mode
    for s, seed in (("f1_s2", 43), ("f1_s3", 44)):
        r2 = f1_reads(seed, 400)
        srt2 = sorted(range(len(r2)), key=lambda k: (r2[k][0], r2[k][1], k))
        t = sam_header(F1_CONTIGS, True)
        for k in srt2:
            ci, pos1, flag, mapq, cig = r2[k]
            t += sam_line(k, flag, F1_CONTIGS[ci][0], pos1, mapq, cig)
        write(os.path.join(d, s + ".sam"), t)
        to_bam(os.path.join(d, s + ".sam"), os.path.join(d, s + ".bam"), True)
        os.remove(os.path.join(d, s + ".sam"))
    write(os.path.join(d, "f1_3.list"), "f1.bam\nf1_s2.bam\n\nf1_s3.bam\n")
    write(os.path.join(d, "f1_mixed.list"), "f1.bam\nf1_noidx.bam\nf1_unsorted.bam\n")

    gff = "\n".join([
        "##gff-version 3",
        "# comment line",
        "chrA\tsrc\tgene\t100\t700\t.\t+\t.\tID=g1",
        "chrA\tsrc\tmRNA\t100\t700\t.\t+\t.\tID=g1.t1;Parent=g1",
        "chrA\tsrc\texon\t100\t260\t.\t+\t.\tID=g1.t1.e1;Parent=g1.t1",
        "chrA\tsrc\tCDS\t120\t260\t.\t+\t0\tID=g1.t1.c1;Parent=g1.t1",
        "chrA\tsrc\texon\t400\t520\t.\t+\t.\tID=g1.t1.e2;Parent=g1.t1",
        "chrA\tsrc\tCDS\t400\t520\t.\t+\t0\tID=g1.t1.c2;Parent=g1.t1",
        "chrA\tsrc\texon\t600\t700\t.\t+\t.\tID=g1.t1.e3;Parent=g1.t1",
        "chrA\tsrc\tCDS\t600\t690\t.\t+\t0\tID=g1.t1.c3;Parent=g1.t1",
        "chrA\tsrc\tCDS\t120\t200\t.\t-\t0\tID=g0.t1.c1;Parent=g0.t1",
        "chrA\tsrc\tCDS\t120\t180\t.\t-\t0\tID=g0.t0.c1;Parent=g0.t0",
        "chrA\tsrc\tCDS\t900\t1001\t.\t-\t0\tParent=g2.t1",
        "chrA\tsrc\tCDS\t650\t720\t.\t-\t0\tID=ov1;Parent=g3.t1,g3.t2",
        "chrA\tsrc\tCDS\t721\t760\t.\t-\t0\tID=adj1;Parent=g4.t1",
        "chrB\tsrc\tCDS\t1\t90\t.\t+\t0\tID=b1.c1;Parent=b1.t1",
        "chrB\tsrc\tCDS\t200\t500\t.\t+\t0\tID=b2.c1;Parent=b2.t1;Note=x",
        "chrB\tsrc\tCDS\t250\t300\t.\t+\t0\tID=b2.c2;Parent=b2.t1",
        "chrB\tsrc\texon\t10\t40\t.\t+\t.\tID=onlyid",
        "chrZ\tsrc\tCDS\t5\t50\t.\t+\t0\tID=z1;Parent=z.t1",
        "",
    ]) + "\n"
    write(os.path.join(d, "f1.gff"), gff)

    gtf = "\n".join([
        "# gtf comment",
        'chrA\tsrc\tCDS\t120\t260\t.\t+\t0\tgene_id "g1"; transcript_id "g1.t1";',
        'chrA\tsrc\tCDS\t400\t520\t.\t+\t0\tgene_id "g1"; transcript_id "g1.t1";',
        'chrA\tsrc\texon\t100\t260\t.\t+\t.\tgene_id "g1"; transcript_id "g1.t1";',
        'chrA\tsrc\tCDS\t120\t200\t.\t-\t0\tgene_id "g0"; transcript_id "g0.t1";',
        'chrA\tsrc\tCDS\t900\t1001\t.\t-\t0\tgene_id "g2"; transcript_id "g2.t1"; extra "y";',
        'chrB\tsrc\tCDS\t200\t500\t.\t+\t0\tgene_id "b2"; transcript_id "b2.t1";',
        'chrB\tsrc\tCDS\t1\t90\t.\t+\t0\tgene_id "b1"; transcript_id "b1.t1";',
        'chrZ\tsrc\tCDS\t5\t50\t.\t+\t0\tgene_id "z"; transcript_id "z.t1";',
    ]) + "\n"
    write(os.path.join(d, "f1.gtf"), gtf)

    bed3 = "\n".join([
        "#chr\tstart\tend",
        "chrA\t100\t200",
        "chrA\t150\t400",
        "chrA\t0900\t1001",
        "chrB\t1\t500",
        "chrB\t300\t250",
        "chrQ\t1\t10",
        "chrA\t100\t200",
        "chrA\t401\t401",
    ]) + "\n"
    write(os.path.join(d, "f1.bed3"), bed3)

    bed4 = "\n".join([
        "chrA\t100\t200\tr1",
        "chrA\t150\t400\tr2",
        "chrA\t500\t600\tr1",
        "chrB\t10\t260\tq2",
        "chrB\t10\t60\tq1",
        "chrB\t300\t250\tbad",
        "chrA\t700\t1001\ttail",
    ]) + "\n"
    write(os.path.join(d, "f1.bed4"), bed4)


F1_CASES = [
    # name, args (inputs are relative to the fixture dir)
    ("chr", ["-i", "f1.bam"]),
    ("chr_t1", ["-i", "f1.bam", "-t", "1"]),
    ("chr_q20", ["-i", "f1.bam", "-q", "20"]),
    ("chr_x0", ["-i", "f1.bam", "-x", "0"]),
    ("chr_d3", ["-i", "f1.bam", "-d", "3"]),
    ("chr_s", ["-i", "f1.bam", "-s"]),
    ("chr_a", ["-i", "f1.bam", "-a"]),
    ("chr_sam", ["-i", "f1.sam"]),
    ("chr_noidx", ["-i", "f1_noidx.bam"]),
    ("chr_unsorted", ["-i", "f1_unsorted.bam"]),
    ("chr_unsorted_sam_a", ["-i", "f1_unsorted.sam", "-a"]),
    ("w100", ["-i", "f1.bam", "-w", "100"]),
    ("w100_a", ["-i", "f1.bam", "-w", "100", "-a"]),
    ("w100_s", ["-i", "f1.bam", "-w", "100", "-s"]),
    ("w1", ["-i", "f1.bam", "-w", "1"]),
    ("w149_d2", ["-i", "f1.bam", "-w", "149", "-d", "2"]),
    ("w150", ["-i", "f1.bam", "-w", "150"]),
    ("w200", ["-i", "f1.bam", "-w", "200"]),
    ("w200_a", ["-i", "f1.bam", "-w", "200", "-a"]),
    ("w200_s", ["-i", "f1.bam", "-w", "200", "-s"]),
    ("w1000", ["-i", "f1.bam", "-w", "1000"]),
    ("gff", ["-i", "f1.bam", "-g", "f1.gff"]),
    ("gff_exon", ["-i", "f1.bam", "-g", "f1.gff", "-f", "exon"]),
    ("gff_a", ["-i", "f1.bam", "-g", "f1.gff", "-a"]),
    ("gff_s_a", ["-i", "f1.bam", "-g", "f1.gff", "-s", "-a"]),
    ("gff_unsorted_a", ["-i", "f1_unsorted.bam", "-g", "f1.gff", "-a"]),
    ("gff_w", ["-i", "f1.bam", "-g", "f1.gff", "-w", "100"]),
    ("gtf", ["-i", "f1.bam", "-g", "f1.gtf"]),
    ("bed3", ["-i", "f1.bam", "-b", "f1.bed3"]),
    ("bed3_a", ["-i", "f1.bam", "-b", "f1.bed3", "-a"]),
    ("bed4_d10", ["-i", "f1.bam", "-b", "f1.bed4", "-d", "10"]),
    ("bed4_s", ["-i", "f1.bam", "-b", "f1.bed4", "-s"]),
    ("list3", ["-i", "f1_3.list"]),
    ("list3_a", ["-i", "f1_3.list", "-a"]),
    ("list3_w100", ["-i", "f1_3.list", "-w", "100"]),
    ("list3_w200", ["-i", "f1_3.list", "-w", "200"]),
    ("list3_gff", ["-i", "f1_3.list", "-g", "f1.gff"]),
    ("list3_bed4", ["-i", "f1_3.list", "-b", "f1.bed4"]),
    ("list_mixed_gff_a", ["-i", "f1_mixed.list", "-g", "f1.gff", "-a"]),
]


# ---------------------------------------------------------------------------------------------
# F2: 18-bit wrap fixture (262150 stacked reads)
# ---------------------------------------------------------------------------------------------
F2_CONTIGS = [("w1", 400), ("w2", 300)]


def build_f2(d):
    os.makedirs(d, exist_ok=True)
    sam = [sam_header(F2_CONTIGS, True)]
    k = 0
    for _ in range(262150):
        sam.append("r\t0\tw1\t101\t60\t10M\t*\t0\t0\t*\t*\n")
        k += 1
    for _ in range(7):
        sam.append("r\t0\tw1\t105\t60\t20M\t*\t0\t0\t*\t*\n")
    for _ in range(5):
        sam.append("r\t0\tw2\t11\t60\t30M\t*\t0\t0\t*\t*\n")
    write(os.path.join(d, "f2.sam"), "".join(sam))
    to_bam(os.path.join(d, "f2.sam"), os.path.join(d, "f2.bam"), True)
    os.remove(os.path.join(d, "f2.sam"))
    write(os.path.join(d, "f2_2.list"), "f2.bam\nf2.bam\n")
    write(os.path.join(d, "f2.bed4"), "w1\t90\t140\thot\nw2\t1\t300\tcold\n")


F2_CASES = [
    ("chr", ["-i", "f2.bam"]),                 # uint32 cells
    ("chr_s", ["-i", "f2.bam", "-s"]),         # 18-bit cells
    ("chr_a", ["-i", "f2.bam", "-a"]),         # 18-bit cells
    ("w50", ["-i", "f2.bam", "-w", "50"]),     # mode 6, 18-bit
    ("w150", ["-i", "f2.bam", "-w", "150"]),   # mode 5, uint32
    ("bed4", ["-i", "f2.bam", "-b", "f2.bed4"]),
    ("bed4_a", ["-i", "f2.bam", "-b", "f2.bed4", "-a"]),
    ("list2", ["-i", "f2_2.list"]),            # 2 x 262150 = 524300 -> mod 2^18
]


# ---------------------------------------------------------------------------------------------
# F3: config-1 "tiny.bam": ~10k reads on a 1 000 003 bp reference
# ---------------------------------------------------------------------------------------------
F3_CONTIGS = [("chr1", 700001), ("chr2", 250000), ("chr3", 50001), ("chr4", 1)]


def build_f3(d):
    os.makedirs(d, exist_ok=True)
    rng = random.Random(42)
    recs = []
    total = sum(l for _, l in F3_CONTIGS)
    for i in range(10000):
        x = rng.randrange(total - 1)
        ci = 0
        while x >= F3_CONTIGS[ci][1]:
            x -= F3_CONTIGS[ci][1]
            ci += 1
        ln = F3_CONTIGS[ci][1]
        u = rng.random()
        if u < 0.85:
            cig = "150M"
        elif u < 0.90:
            a = rng.randint(10, 140)
            cig = "%dM%dD%dM" % (a, rng.randint(1, 20), 150 - a)
        elif u < 0.95:
            a = rng.randint(10, 130)
            ins = rng.randint(1, 10)
            cig = "%dM%dI%dM" % (a, ins, 150 - a - ins)
        elif u < 0.99:
            s = rng.randint(1, 40)
            cig = "%dS%dM" % (s, 150 - s)
        else:
            a = rng.randint(20, 130)
            cig = "%dM%dN%dM" % (a, rng.randint(100, 5000), 150 - a)
        span = cigar_ref_span(cig)
        pos1 = min(x + 1, max(1, ln - span + 1))
        v = rng.random()
        flag = 0 if rng.random() < 0.5 else 16
        if v < 0.02:
            flag |= 1024
        elif v < 0.03:
            flag |= 256
        elif v < 0.04:
            flag |= 512
        mapq = rng.choice([0, 20, 60, 60, 60])
        recs.append((ci, pos1, flag, mapq, cig))
    srt = sorted(range(len(recs)), key=lambda k: (recs[k][0], recs[k][1], k))
    sam = [sam_header(F3_CONTIGS, True)]
    for k in srt:
        ci, pos1, flag, mapq, cig = recs[k]
        sam.append(sam_line(k, flag, F3_CONTIGS[ci][0], pos1, mapq, cig))
    for k in range(50):     # unmapped, unplaced reads at the end of the file
        sam.append("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*\n" % k)
    write(os.path.join(d, "tiny.sam"), "".join(sam))
    to_bam(os.path.join(d, "tiny.sam"), os.path.join(d, "tiny.bam"), True)
    os.remove(os.path.join(d, "tiny.sam"))
    # synthetic annotation: 40 transcripts with 1-6 CDS each
    g = ["##gff-version 3"]
    rg = random.Random(5)
    for t in range(40):
        ci = rg.choice([0, 0, 0, 1, 1, 2])
        name, ln = F3_CONTIGS[ci]
        s = rg.randint(1, ln - 40000)
        for e in range(rg.randint(1, 6)):
            el = rg.randint(60, 600)
            g.append("%s\tsyn\tCDS\t%d\t%d\t.\t+\t0\tID=t%d.c%d;Parent=t%d" % (name, s, s + el - 1, t, e, t))
            s += el + rg.randint(50, 5000)
    write(os.path.join(d, "tiny.gff"), "\n".join(g) + "\n")


F3_CASES = [
    ("chr", ["-i", "tiny.bam"]),
    ("chr_s", ["-i", "tiny.bam", "-s"]),
    ("gff", ["-i", "tiny.bam", "-g", "tiny.gff"]),
    ("w1000", ["-i", "tiny.bam", "-w", "1000"]),
    ("w100", ["-i", "tiny.bam", "-w", "100"]),
    ("w100_a", ["-i", "tiny.bam", "-w", "100", "-a"]),
]

# ---------------------------------------------------------------------------------------------
# F4: last-base targets.  The reference's indexed path (ProDealChrBambai, PD:676-786) hands a window's statistics
# to the genes with GeneStart < MeMEnd (PD:299-303), and MeMEnd is clipped to the contig length: a target that STARTS
# on the last base of its contig gets no statistics there (found by tests/fuzz_vs_ref.py), while the SiteInfo paths
# (-a, no index, list) count it.
# ---------------------------------------------------------------------------------------------
F4_CONTIGS = [("k2", 2), ("k201", 201), ("k37", 37), ("k500", 500)]


def build_f4(d):
    os.makedirs(d, exist_ok=True)
    rng = random.Random(404)
    recs = []
    for i in range(900):
        ci = rng.randrange(len(F4_CONTIGS))
        ln = F4_CONTIGS[ci][1]
        span = rng.randint(1, min(ln, 60))
        pos1 = rng.randint(1, ln - span + 1)
        if rng.random() < 0.3:
            pos1 = ln - span + 1                        # pile reads onto the contig ends
        recs.append((ci, pos1, rng.choice([0, 16, 0, 16, 1024]), rng.choice([0, 30, 60]), "%dM" % span))
    srt = sorted(range(len(recs)), key=lambda k: (recs[k][0], recs[k][1], k))
    sam = sam_header(F4_CONTIGS, True)
    for k in srt:
        ci, pos1, flag, mapq, cig = recs[k]
        sam += sam_line(k, flag, F4_CONTIGS[ci][0], pos1, mapq, cig)
    write(os.path.join(d, "e.sam"), sam)
    to_bam(os.path.join(d, "e.sam"), os.path.join(d, "e.bam"), True)
    shutil.copy(os.path.join(d, "e.bam"), os.path.join(d, "e_noidx.bam"))
    write(os.path.join(d, "e.list"), "e.bam\ne_noidx.bam\n")
    write(os.path.join(d, "e.bed"), "\n".join([
        "k2\t2\t2\tlastbase", "k2\t1\t2\tboth", "k201\t201\t201\tend201", "k201\t150\t201\ttail", "k201\t1\t10\thead",
        "k37\t37\t37\talone",                       # the only target of its contig: its window starts there, it is counted
        "k500\t10\t40\ta", "k500\t500\t500\tz", "k500\t499\t500\ty"]) + "\n")
    write(os.path.join(d, "e.gff"), "\n".join([
        "k2\ts\tCDS\t2\t2\t.\t+\t0\tID=a;Parent=g_last", "k201\ts\tCDS\t201\t201\t.\t+\t0\tID=b;Parent=g_end",
        "k201\ts\tCDS\t20\t80\t.\t+\t0\tID=c;Parent=g_mid", "k500\ts\tCDS\t500\t500\t.\t-\t0\tID=d;Parent=g_z",
        "k500\ts\tCDS\t100\t200\t.\t-\t0\tID=e;Parent=g_z", "k37\ts\tCDS\t37\t37\t.\t+\t0\tID=f;Parent=g_alone"]) + "\n")


F4_CASES = [
    ("bed4", ["-i", "e.bam", "-b", "e.bed"]),
    ("bed4_a", ["-i", "e.bam", "-b", "e.bed", "-a"]),
    ("bed4_noidx", ["-i", "e_noidx.bam", "-b", "e.bed"]),
    ("bed4_s", ["-i", "e.bam", "-b", "e.bed", "-s"]),
    ("bed4_list", ["-i", "e.list", "-b", "e.bed"]),
    ("gff", ["-i", "e.bam", "-g", "e.gff"]),
    ("gff_sam", ["-i", "e.sam", "-g", "e.gff"]),
    ("chr", ["-i", "e.bam"]),
    ("w10", ["-i", "e.bam", "-w", "10"]),
    ("w200", ["-i", "e.bam", "-w", "200"]),
]

# ---------------------------------------------------------------------------------------------
# F5: the GC(%) column (-c -r): FASTA with lower case / N / comments / CRLF / blank lines / a gzip copy, a sequence
# name the BAM header does not know (it becomes contig 0 in the reference's name table), multi-entry ids (only the
# entry that creates an id is counted)
# ---------------------------------------------------------------------------------------------
F5_CONTIGS = [("g1", 1500), ("g2", 700), ("g3", 1), ("g4", 301)]


def build_f5(d):
    os.makedirs(d, exist_ok=True)
    rng = random.Random(505)
    recs = []
    for i in range(1200):
        ci = rng.choice([0, 0, 1, 3])
        ln = F5_CONTIGS[ci][1]
        cig = rng.choice(["40M", "15M3D25M", "8S32M", "20M60N20M", "25M2I13M", "40="])
        span = cigar_ref_span(cig)
        pos1 = rng.randint(1, ln - span + 1)
        recs.append((ci, pos1, rng.choice([0, 16, 0, 16, 1024, 256]), rng.choice([0, 20, 60]), cig))
    for nm, sel in (("c", recs), ("c2", recs[:500])):
        srt = sorted(range(len(sel)), key=lambda k: (sel[k][0], sel[k][1], k))
        sam = sam_header(F5_CONTIGS, True)
        for k in srt:
            ci, pos1, flag, mapq, cig = sel[k]
            sam += sam_line(k, flag, F5_CONTIGS[ci][0], pos1, mapq, cig)
        write(os.path.join(d, nm + ".sam"), sam)
        to_bam(os.path.join(d, nm + ".sam"), os.path.join(d, nm + ".bam"), True)
    os.remove(os.path.join(d, "c2.sam"))
    shutil.copy(os.path.join(d, "c.bam"), os.path.join(d, "c_noidx.bam"))
    write(os.path.join(d, "c.list"), "c.bam\nc2.bam\n")
    fa = ""
    for name, ln in F5_CONTIGS + [("extra", 90)]:
        seq = "".join(rng.choice("ACGTACGTacgtNnRY") for _ in range(ln + (25 if name == "g2" else 0)))   # g2: longer than its contig
        fa += ">%s%s\n" % (name, {"g1": " primary assembly", "g2": "\tlen=700"}.get(name, ""))
        width = {"g1": 60, "g2": 70, "g3": 60, "g4": 50, "extra": 30}[name]
        for k in range(0, len(seq), width):
            fa += seq[k:k + width] + ("\r\n" if name == "g4" else "\n")
            if name == "g1" and k == 600:
                fa += "\n"                                # an empty line inside a record
    with open(os.path.join(d, "c.fa"), "wb") as f:
        f.write(fa.encode())
    with open(os.path.join(d, "c.fa.gz"), "wb") as f:
        f.write(gzip.compress(fa.encode(), mtime=0))
    write(os.path.join(d, "c.gff"), "\n".join([
        "g1\ts\tCDS\t100\t180\t.\t+\t0\tID=a;Parent=t1", "g1\ts\tCDS\t300\t420\t.\t+\t0\tID=b;Parent=t1",
        "g1\ts\tCDS\t1400\t1500\t.\t+\t0\tID=c;Parent=t2", "g1\ts\texon\t90\t200\t.\t+\t.\tID=x;Parent=t1",
        "g2\ts\tCDS\t1\t700\t.\t-\t0\tID=d;Parent=u1", "g2\ts\tCDS\t650\t700\t.\t-\t0\tID=e;Parent=u1",
        "g4\ts\tCDS\t10\t300\t.\t+\t0\tID=f;Parent=v1", "extra\ts\tCDS\t5\t60\t.\t+\t0\tID=g;Parent=w1",
        "nowhere\ts\tCDS\t5\t60\t.\t+\t0\tID=h;Parent=w2"]) + "\n")
    write(os.path.join(d, "c.gtf"), "\n".join([
        'g1\ts\tCDS\t100\t180\t.\t+\t0\tgene_id "G1"; transcript_id "T1";',
        'g1\ts\tCDS\t300\t420\t.\t+\t0\tgene_id "G1"; transcript_id "T1";',
        'g2\ts\tCDS\t20\t90\t.\t-\t0\tgene_id "G2"; transcript_id "T2";']) + "\n")
    write(os.path.join(d, "c.bed3"), "g1\t1\t1500\ng1\t700\t800\ng2\t350\t350\nextra\t10\t20\ng4\t1\t301\n")
    write(os.path.join(d, "c.bed4"), "g1\t1\t100\tA\ng1\t500\t900\tA\ng1\t50\t60\tB\ng2\t1\t700\tC\ng4\t100\t200\tD\n")


F5_CASES = [
    ("chr", ["-i", "c.bam", "-c", "-r", "c.fa"]),
    ("chr_gz", ["-i", "c.bam", "-c", "-r", "c.fa.gz"]),
    ("chr_sam", ["-i", "c.sam", "-c", "-r", "c.fa"]),
    ("chr_noidx", ["-i", "c_noidx.bam", "-c", "-r", "c.fa"]),
    ("chr_s_a", ["-i", "c.bam", "-s", "-a", "-c", "-r", "c.fa"]),
    ("chr_list", ["-i", "c.list", "-c", "-r", "c.fa"]),
    ("w150", ["-i", "c.bam", "-w", "150", "-c", "-r", "c.fa"]),
    ("w400_d5", ["-i", "c.bam", "-w", "400", "-d", "5", "-c", "-r", "c.fa.gz"]),
    ("w250_list", ["-i", "c.list", "-w", "250", "-c", "-r", "c.fa"]),
    ("gff", ["-i", "c.bam", "-g", "c.gff", "-c", "-r", "c.fa"]),
    ("gff_exon", ["-i", "c.bam", "-g", "c.gff", "-f", "exon", "-c", "-r", "c.fa"]),
    ("gff_list_a", ["-i", "c.list", "-g", "c.gff", "-a", "-c", "-r", "c.fa"]),
    ("gtf", ["-i", "c.bam", "-g", "c.gtf", "-c", "-r", "c.fa"]),
    ("bed3", ["-i", "c.bam", "-b", "c.bed3", "-c", "-r", "c.fa"]),
    ("bed4", ["-i", "c.bam", "-b", "c.bed4", "-c", "-r", "c.fa", "-q", "20"]),
    ("bed4_noidx", ["-i", "c_noidx.bam", "-b", "c.bed4", "-c", "-r", "c.fa"]),
    ("r_only", ["-i", "c.bam", "-r", "c.fa"]),                    # -r without -c: no GC column
]

# ---------------------------------------------------------------------------------------------
# F6: PAF input (paf_main, PD:852-2024): targets from the first file or from -r, cg:Z: walks vs plain target spans,
# reversed coordinates, tp:A:S lines and -x, -q on column 12, a list of files, gzip input, a name only a later file knows
# ---------------------------------------------------------------------------------------------
F6_TARGETS = [("ctgA", 3000), ("ctgB", 800), ("ctgC", 1), ("ctgD", 151)]


def f6_paf(seed, n, targets):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        name, L = rng.choice([t for t in targets if t[1] >= 2])
        a = rng.randrange(0, L); b = min(L, a + rng.randrange(1, 700))
        tags = ["tp:A:%s" % rng.choice("PPPS"), "cm:i:%d" % rng.randrange(100)]
        if i % 2:
            rem, ops = b - a, []
            while rem > 0:
                k = min(rem, rng.randrange(1, 250)); ops.append("%d%s" % (k, rng.choice("MMM=XDN"))); rem -= k
                if rng.random() < 0.3:
                    ops.append("%d%s" % (rng.randrange(1, 30), rng.choice("IS")))
            tags.append("cg:Z:" + "".join(ops))
            s, e = a, b
        else:
            s, e = max(a, 1), max(b, 1)
        if i % 17 == 0:
            s, e = e, s
        out.append("\t".join(["q%d" % i, "4000", "5", "800", rng.choice("+-"), name, str(L), str(s), str(e),
                               str(abs(e - s)), str(abs(e - s) + 7), str(rng.choice([0, 10, 60]))] + tags))
    return "\n".join(out) + "\n"


def build_f6(d):
    os.makedirs(d, exist_ok=True)
    rng = random.Random(606)
    write(os.path.join(d, "p.paf"), f6_paf(61, 260, F6_TARGETS))
    with open(os.path.join(d, "p2.paf.gz"), "wb") as f:
        # the second file also names a target the first never mentions: it is looked up with operator[] -> target 0
        extra = "qx\t900\t0\t300\t+\tnovel\t500\t100\t400\t300\t300\t60\ttp:A:P\n"
        f.write(gzip.compress((f6_paf(62, 120, F6_TARGETS[:2]) + extra).encode(), mtime=0))
    write(os.path.join(d, "p.list"), "p.paf\np2.paf.gz\n")
    fa = ""
    for name, ln in [F6_TARGETS[1], F6_TARGETS[0], F6_TARGETS[3], F6_TARGETS[2]]:       # ids follow the FASTA order
        seq = "".join(rng.choice("ACGTacgtNn") for _ in range(ln))
        fa += ">%s\n" % name + "".join(seq[k:k + 80] + "\n" for k in range(0, ln, 80))
    write(os.path.join(d, "p.fa"), fa)
    write(os.path.join(d, "p.gff"), "\n".join([
        "ctgA\ts\tCDS\t100\t400\t.\t+\t0\tID=a;Parent=m1", "ctgA\ts\tCDS\t900\t1500\t.\t+\t0\tID=b;Parent=m1",
        "ctgA\ts\tCDS\t2990\t3000\t.\t+\t0\tID=c;Parent=m2", "ctgB\ts\tCDS\t1\t800\t.\t-\t0\tID=d;Parent=n1",
        "ctgD\ts\tCDS\t151\t151\t.\t-\t0\tID=e;Parent=o1", "ctgD\ts\tCDS\t5\t60\t.\t-\t0\tID=f;Parent=o2"]) + "\n")
    write(os.path.join(d, "p.bed3"), "ctgA\t1\t3000\nctgA\t1200\t1300\nctgB\t400\t400\nctgD\t1\t151\n")
    write(os.path.join(d, "p.bed4"), "ctgA\t1\t100\tA\nctgA\t2000\t2500\tA\nctgB\t1\t800\tB\nctgD\t100\t151\tD\n")


F6_CASES = [
    ("chr", ["-i", "p.paf"]),
    ("chr_gz_x0", ["-i", "p2.paf.gz", "-x", "0"]),
    ("chr_list", ["-i", "p.list"]),
    ("chr_q20_a", ["-i", "p.paf", "-q", "20", "-a"]),
    ("w100", ["-i", "p.paf", "-w", "100"]),
    ("w500_d3", ["-i", "p.paf", "-w", "500", "-d", "3"]),
    ("w150_list_x0", ["-i", "p.list", "-w", "150", "-x", "0"]),
    ("gff", ["-i", "p.paf", "-g", "p.gff"]),
    ("gff_list_a", ["-i", "p.list", "-g", "p.gff", "-a"]),
    ("bed3", ["-i", "p.paf", "-b", "p.bed3"]),
    ("bed4", ["-i", "p.paf", "-b", "p.bed4", "-d", "2"]),
    ("gc_chr", ["-i", "p.paf", "-r", "p.fa", "-c"]),
    ("gc_w300", ["-i", "p.paf", "-r", "p.fa", "-c", "-w", "300"]),
    ("gc_gff", ["-i", "p.paf", "-r", "p.fa", "-c", "-g", "p.gff"]),
    ("gc_bed3_list", ["-i", "p.list", "-r", "p.fa", "-c", "-b", "p.bed3"]),
    ("c_without_r", ["-i", "p.paf", "-c"]),
]

# ---------------------------------------------------------------------------------------------
# F7: CRAM 3.0 input written by the reference's own htslib (oracle/_ref/sam2bam ... out.cram): reference-free and
# reference-based files, with and without .crai, multi-reference slices (tiny contigs), bases + qualities + tags +
# mate fields in the records, a list mixing CRAM and BAM, and -r next to a CRAM (htslib then takes the contig
# lengths from the FASTA)
# ---------------------------------------------------------------------------------------------
F7_CONTIGS = [("k1", 4000), ("k2", 900), ("k3", 1), ("k4", 250)]


def build_f7(d):
    os.makedirs(d, exist_ok=True)
    rng = random.Random(707)
    ref = {nm: "".join(rng.choice("ACGT") for _ in range(ln)) for nm, ln in F7_CONTIGS}
    recs = []
    for i in range(1500):
        ci = rng.choice([0, 0, 0, 1, 3])
        nm, L = F7_CONTIGS[ci]
        ops = []
        k = rng.randrange(1, 5)
        for j in range(k):
            ops.append((rng.randrange(5, 50), "M"))
            if j < k - 1:
                ops.append((rng.randrange(1, 25), rng.choice("IDNX=")))
        if rng.random() < 0.3:
            ops.insert(0, (rng.randrange(1, 15), "S"))
        if rng.random() < 0.2:
            ops.append((rng.randrange(1, 15), "S"))
        if rng.random() < 0.1:
            ops.insert(0, (rng.randrange(1, 9), "H"))
        span = sum(n for n, o in ops if o in "MDN=X")
        if span >= L:
            continue
        pos = rng.randrange(1, L - span + 1)
        seq, rp = [], pos - 1
        for n, o in ops:
            if o in "M=X":
                for t in range(n):
                    b = ref[nm][rp + t]
                    if o == "X" or (o == "M" and rng.random() < 0.04):
                        b = rng.choice([x for x in "ACGT" if x != b])
                    seq.append(b)
                rp += n
            elif o in "IS":
                seq += [rng.choice("ACGTN") for _ in range(n)]
            elif o in "DN":
                rp += n
        seq = "".join(seq)
        qual = "".join(chr(33 + rng.randrange(2, 41)) for _ in seq)
        flag = rng.choice([0, 16, 99, 147, 83, 163, 256, 1024, 0, 16])
        mate = ("=", pos + rng.randrange(0, 200), rng.randrange(-300, 300)) if flag & 1 else ("*", 0, 0)
        tags = rng.choice(["", "\tNM:i:%d" % rng.randrange(9), "\tRG:Z:g1\tAS:i:%d" % rng.randrange(200)])
        recs.append((ci, pos, "r%d\t%d\t%s\t%d\t%d\t%s\t%s\t%d\t%d\t%s\t%s%s" % (
            i, flag, nm, pos, rng.choice([0, 10, 60]), "".join("%d%s" % x for x in ops), mate[0], mate[1], mate[2], seq, qual, tags)))
    recs.append((99, 0, "u1\t77\t*\t0\t0\t*\t*\t0\t0\tACGTACGT\tIIIIIIII"))
    recs.sort(key=lambda r: (r[0], r[1]))
    hdr = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in F7_CONTIGS) + "@RG\tID:g1\tSM:x\n"
    write(os.path.join(d, "m.sam"), hdr + "\n".join(r[2] for r in recs) + "\n")
    # the FASTA handed to -r is LONGER than k2 says: with CRAM input htslib rewrites the header's LN from it
    fa = ""
    for nm, ln in F7_CONTIGS:
        sq = ref[nm] + ("ACGTT" * 5 if nm == "k2" else "")
        fa += ">%s\n" % nm + "".join(sq[k:k + 60] + "\n" for k in range(0, len(sq), 60))
    write(os.path.join(d, "m.fa"), fa)
    write(os.path.join(d, "m_exact.fa"), "".join(">%s\n" % nm + "".join(ref[nm][k:k + 60] + "\n" for k in range(0, ln, 60)) for nm, ln in F7_CONTIGS))
    subprocess.run([SAM2BAM, "m.sam", "m.cram"], cwd=d, check=True, stderr=subprocess.DEVNULL)                       # reference-free + .crai
    subprocess.run([SAM2BAM, "m.sam", "m_noidx.cram", "noindex"], cwd=d, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([SAM2BAM, "m.sam", "m_ref.cram", "ref=m_exact.fa"], cwd=d, check=True, stderr=subprocess.DEVNULL)  # reference-based
    subprocess.run([SAM2BAM, "m.sam", "m_v31.cram", "fmt=cram,version=3.1"], cwd=d, check=True, stderr=subprocess.DEVNULL)  # CRAM 3.1 (rANS Nx16)
    subprocess.run([SAM2BAM, "m.sam", "m_v21.cram", "noindex", "fmt=cram,version=2.1"], cwd=d, check=True, stderr=subprocess.DEVNULL)  # CRAM 2.1 framing
    for junk in ("m_exact.fa.fai",):
        if os.path.exists(os.path.join(d, junk)):
            os.remove(os.path.join(d, junk))
    to_bam(os.path.join(d, "m.sam"), os.path.join(d, "m.bam"), True)
    write(os.path.join(d, "m.list"), "m.cram\nm.bam\nm_noidx.cram\n")
    write(os.path.join(d, "m.gff"), "\n".join([
        "k1\ts\tCDS\t100\t700\t.\t+\t0\tID=a;Parent=t1", "k1\ts\tCDS\t1500\t1900\t.\t+\t0\tID=b;Parent=t1",
        "k1\ts\tCDS\t3990\t4000\t.\t+\t0\tID=c;Parent=t2", "k2\ts\tCDS\t1\t900\t.\t-\t0\tID=d;Parent=u1",
        "k4\ts\tCDS\t20\t250\t.\t-\t0\tID=e;Parent=v1"]) + "\n")
    write(os.path.join(d, "m.bed4"), "k1\t1\t400\tA\nk1\t2000\t3000\tA\nk2\t1\t900\tB\nk4\t100\t250\tD\n")


F7_CASES = [
    ("chr", ["-i", "m.cram"]),
    ("chr_noidx", ["-i", "m_noidx.cram"]),
    ("chr_refbased", ["-i", "m_ref.cram"]),
    ("chr_s", ["-i", "m.cram", "-s"]),
    ("chr_q20_x0", ["-i", "m.cram", "-q", "20", "-x", "0"]),
    ("w100_a", ["-i", "m.cram", "-w", "100", "-a"]),
    ("w500", ["-i", "m_noidx.cram", "-w", "500", "-d", "3"]),
    ("gff", ["-i", "m.cram", "-g", "m.gff"]),
    ("gff_noidx", ["-i", "m_noidx.cram", "-g", "m.gff"]),
    ("bed4_a", ["-i", "m_ref.cram", "-b", "m.bed4", "-a"]),
    ("list", ["-i", "m.list"]),
    ("list_gff", ["-i", "m.list", "-g", "m.gff"]),
    ("r_overrides_ln", ["-i", "m.cram", "-r", "m.fa"]),
    ("gc", ["-i", "m.cram", "-r", "m.fa", "-c", "-w", "300"]),
    ("chr_v31", ["-i", "m_v31.cram"]),
    ("bed4_v31", ["-i", "m_v31.cram", "-b", "m.bed4"]),
    ("chr_v21", ["-i", "m_v21.cram"]),
]

BIG_OUTPUT = 400000   # decompressed bytes above which only hashes are committed


def run_cases(fx, d, cases, manifest):
    outd = os.path.join(d, "expected")
    if os.path.isdir(outd):
        shutil.rmtree(outd)
    os.makedirs(outd)
    for name, args in cases:
        prefix = os.path.join("expected", name)
        cmd = [REF] + args + ["-o", prefix]
        p = subprocess.run(cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        entry = {"fixture": fx, "name": name, "args": args, "returncode": p.returncode,
                 "stdout": p.stdout.decode(), "outputs": {}}
        for fn in sorted(os.listdir(outd)):
            if not fn.startswith(name + "."):
                continue
            path = os.path.join(outd, fn)
            gzb = open(path, "rb").read()
            txt = gzip.decompress(gzb)
            suffix = fn[len(name) + 1:]
            keep = len(txt) <= BIG_OUTPUT
            entry["outputs"][suffix] = {"gz_sha256": sha(gzb), "text_sha256": sha(txt),
                                        "text_bytes": len(txt), "committed": keep}
            if not keep:
                os.remove(path)
        manifest.append(entry)
        print("%-4s %-22s rc=%d %s" % (fx, name, p.returncode, " ".join(sorted(entry["outputs"]))))


def main():
    if not (os.path.exists(REF) and os.path.exists(SAM2BAM)):
        sys.exit("build the reference oracle first: make -C oracle ref")
    manifest = []
    only = sys.argv[1:]                                  # e.g. `make_golden.py f5`: rebuild these, keep the other entries
    if only:
        manifest = [e for e in json.load(open(os.path.join(HERE, "manifest.json"))) if e["fixture"] not in only]
    for fx, build, cases in (("f1", build_f1, F1_CASES), ("f2", build_f2, F2_CASES),
                             ("f3", build_f3, F3_CASES), ("f4", build_f4, F4_CASES), ("f5", build_f5, F5_CASES),
                             ("f6", build_f6, F6_CASES), ("f7", build_f7, F7_CASES)):
        if only and fx not in only:
            continue
        d = os.path.join(HERE, fx)
        build(d)
        run_cases(fx, d, cases, manifest)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
