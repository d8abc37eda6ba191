#!/usr/bin/env python3
"""tests/fuzz_vs_ref.py — differential fuzzing of the host pipeline (on the CPU oracle engine, tests/harness) against the
compiled reference (oracle/_ref/pandepth_ref): random small SAM inputs, region files with the quirks real files have, random
option mixes.  Compares exit code, stdout and every output file byte for byte.  Needs /root/reference's build (dev container).
usage: fuzz_vs_ref.py [seed] [cases] [big | args | messy | oracle]"""
import glob
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
CLI = os.environ.get("FUZZ_CLI") or os.path.join(ROOT, "tests", "harness", "pandepth_oracle_cli")     # FUZZ_CLI: a copy of the harness binary, so that rebuilding the tree does not disturb a running campaign


MESSY = False       # fuzz_vs_ref.py <seed> <cases> messy: truncated lines, CRLF, extra columns, runs of spaces in the target files
BIG = False         # fuzz_vs_ref.py <seed> <cases> big: contigs of 11-32 Mb, reads and targets clustered around the 10 Mb steps
                    # of the reference's indexed window walk (PD:676-786)


def gen_sam(rng, sorted_hdr, lie=False):
    ncontig = rng.randrange(1, 5)
    lens = [rng.choice([1, 2, 37, 101, 201, 500, 1001, 2500, 10001]) for _ in range(ncontig)]
    if BIG:
        ncontig = rng.randrange(1, 3)
        lens = [rng.choice([11000000, 20000001, 25000000, 32000000, 10000150]) for _ in range(ncontig)]
    names = ["c%d" % i if rng.random() < 0.8 else "chr_%s" % "xyzw"[i] for i in range(ncontig)]
    out = ["@HD\tVN:1.6\tSO:%s" % ("coordinate" if sorted_hdr else rng.choice(["unsorted", "queryname", "unknown"]))]
    for n, l in zip(names, lens):
        out.append("@SQ\tSN:%s\tLN:%d" % (n, l))
    reads = []
    for i in range(rng.randrange(0, 400)):
        t = rng.randrange(ncontig)
        L = lens[t]
        pos = rng.randrange(1, L + 1)
        if BIG and rng.random() < 0.85:                   # near a 10 Mb step (or the contig end)
            c = rng.choice([10000000, 20000000, 30000000, L, 10000000 + rng.randrange(-400, 400)])
            pos = min(L, max(1, c + rng.randrange(-700, 700)))
        ops = []
        for _ in range(rng.randrange(1, 6)):
            op = rng.choice("MMMMIDNS=XHP")
            ops.append((rng.randrange(1, 40 if op != "N" else 300), op))
        if BIG and rng.random() < 0.02 and L > 300000:
            # more than 65535 CIGAR operations: BAM keeps the CIGAR in the CG:B,I tag (SAM spec 4.2.2)
            ops = [(1, "M"), (1, "D")] * rng.randrange(33000, 36000)
            pos = min(pos, L - 80000)
        if ops[0][1] in "DN" and rng.random() < 0.7:
            ops[0] = (ops[0][0], "M")
        # alignments stay inside the contig, as every aligner's do (the reference lets overhanging reads write into the
        # +500 cells of padding behind each contig and reports them for regions that reach beyond the contig end: invalid
        # input on both counts, not reproduced)
        rlen = sum(n for n, o in ops if o in "MDN=X")
        if pos + rlen - 1 > L:
            if rng.random() < 0.5 and rlen <= L:
                pos = rng.randrange(1, L - rlen + 2)
            else:
                pos = min(pos, L)
                ops = [(rng.randrange(1, L - pos + 2), "M")]
        cigar = "".join("%d%s" % o for o in ops)
        qlen = sum(n for n, o in ops if o in "MIS=X")
        flag = rng.choice([0, 16, 0, 16, 256, 512, 1024, 2048, 4, 99, 147, 1040])
        mapq = rng.choice([0, 1, 10, 20, 30, 60, 255])
        if flag & 4:
            # an unmapped read carries no alignment (SAM spec: CIGAR "*").  With a CIGAR, and -x letting it through, the
            # reference's indexed path counts it only inside the 10 Mb window that fetched it by its first base
            # (htslib's end position of an unmapped read is pos + 1): an inconsistent record, not generated, not reproduced
            cigar, qlen = "*", 4
        seq = "A" * qlen if qlen else "*"
        reads.append((t, pos, "r%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t%s\t%s" % (i, flag, names[t], pos, mapq, cigar, seq, "I" * qlen if qlen else "*")))
    if sorted_hdr:
        reads.sort(key=lambda r: (r[0], r[1]))
        if lie and len(reads) > 8:
            # the header keeps saying SO:coordinate, but two blocks of records change places (what concatenating two sorted
            # files gives): only possible without an index; the reference reads such a file with its no-index cursor
            a = rng.randrange(0, len(reads) // 2); b = rng.randrange(len(reads) // 2, len(reads) - 1)
            k = rng.randrange(1, min(len(reads) // 2 - a, len(reads) - b, 40) + 1)
            reads[a:a + k], reads[b:b + k] = reads[b:b + k], reads[a:a + k]
    else:
        rng.shuffle(reads)
    if rng.random() < 0.3:
        reads.append((99, 0, "u1\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII"))
    return "\n".join(out + [r[2] for r in reads]) + "\n", names, lens


def gen_regions(rng, names, lens, kind):
    lines = []
    n = rng.randrange(1, 40)
    for i in range(n):
        t = rng.randrange(len(names))
        nm = names[t] if rng.random() < 0.93 else "nochr"
        # inside the contig: beyond its end the reference reads its padding (zeros for one input, uninitialised heap in
        # list mode) — undefined input, not reproduced
        a = rng.randrange(1, lens[t] + 1); b = min(lens[t], a + rng.randrange(0, 400))
        if BIG and rng.random() < 0.9:
            c = rng.choice([10000000, 20000000, 30000000, lens[t], rng.choice([1, 50, 200])])
            a = min(lens[t], max(1, c + rng.randrange(-900, 900))); b = min(lens[t], a + rng.randrange(0, 700))
        if rng.random() < 0.05 and kind.startswith("bed"):
            a, b = b, a                                  # start > end: BED rows are warned about and skipped (PD:3754-3758).  GFF/GTF
                                                         # rows are taken as they are and turn into reversed merged spans whose region
                                                         # fetch is garbage-in: not generated, not reproduced
        if kind == "gff":
            feat = rng.choice(["CDS", "CDS", "CDS", "exon", "gene", "mRNA"])
            gid = "g%d" % rng.randrange(0, max(1, n // 3))
            attr = rng.choice(["ID=c%d;Parent=%s" % (i, gid), "Parent=%s;ID=c%d" % (gid, i), "ID=%s" % gid, "Parent=%s,%sb" % (gid, gid),
                               "Name=x;Parent=%s;Note=a=b" % gid, "%s" % gid, "ID=c%d;Parent=%s;" % (i, gid)])
            lines.append("%s\tsrc\t%s\t%d\t%d\t.\t%s\t%s\t%s" % (nm, feat, a, b, rng.choice("+-."), rng.choice("012."), attr))
        elif kind == "gtf":
            feat = rng.choice(["CDS", "CDS", "exon", "gene"])
            gid = "G%d" % rng.randrange(0, max(1, n // 3))
            lines.append('%s\tsrc\t%s\t%d\t%d\t.\t+\t0\tgene_id "%s"; transcript_id "%s.1";' % (nm, feat, a, b, gid, gid))
        elif kind == "bed3":
            sep = rng.choice(["\t", "\t", " "])
            lines.append(sep.join([nm, ("%d" % a) if rng.random() < 0.9 else "0%d" % a, "%d" % b]))
        else:
            lines.append("\t".join([nm, "%d" % a, "%d" % b, "id%d" % rng.randrange(0, max(1, n // 2))]))
        if MESSY:
            u = rng.random()
            if u < 0.04 and kind.startswith("bed"):      # a truncated BED line (the reference re-uses what the previous line left;
                                                         # a GFF/GTF line without its attribute column indexes past the end of a vector)
                cols = lines[-1].split("\t")
                keep = 3 if kind.startswith("bed") else 5     # the coordinates stay on the line: values re-used from another
                lines[-1] = "\t".join(cols[:rng.randrange(min(keep, len(cols)), len(cols) + 1)])   # contig's line reach beyond this one's end
            elif u < 0.08:
                lines[-1] += "\r"                        # a Windows line ending
            elif u < 0.12:
                lines[-1] += "\textra\t7"                # more columns than needed
            elif u < 0.15:
                lines[-1] = lines[-1].replace("\t", "  ", 1)
        if rng.random() < 0.05:
            lines.append("")
        if rng.random() < 0.05:
            lines.append("# a comment")
    return "\n".join(lines) + ("\n" if rng.random() < 0.9 else "")


def gen_fasta(rng, names, lens, plain=False):
    """-r for -c: every contig present and at least as long as the header says (beyond a sequence's end, and for a contig
    without a sequence, the reference indexes past its string — undefined, not generated); lower case, N / IUPAC codes,
    header comments, CRLF, blank lines, FASTQ-style records, a second record of the same name (the first wins), names
    the alignment header does not know (they become contig 0 in the name table — fixture f5 pins what target rows naming them do; here they are names no target row uses, placed after contig 0's own record, or
    contig 0 would get a foreign, possibly shorter sequence)."""
    recs = []
    order = list(range(len(names)))
    if rng.random() < 0.5:
        rng.shuffle(order)
    for t in order:
        L = lens[t] + rng.choice([0, 0, 0, 1, 37])
        alpha = rng.choice(["ACGT", "ACGTacgt", "ACGTNacgtnRYKM", "GGCCAT", "at"])
        if L > 200000:
            blocks = ["".join(rng.choice(alpha) for _ in range(1000)) for _ in range(8)]
            seq = "".join(rng.choice(blocks) for _ in range(L // 1000 + 1))[:L]
        else:
            seq = "".join(rng.choice(alpha) for _ in range(L))
        recs.append((names[t], seq))
        if rng.random() < 0.1:
            recs.append((names[t], "G" * min(L, 50)))           # a later record of the same name is ignored
    if rng.random() < 0.3:
        k0 = [i for i, r in enumerate(recs) if r[0] == names[0]][0]
        recs.insert(rng.randrange(k0 + 1, len(recs) + 1), (rng.choice(["alien", "other"]), "ACGT" * rng.randrange(1, 30)))
    eol = "\n" if plain else rng.choice(["\n", "\n", "\n", "\r\n"])
    out = []
    for name, seq in recs:
        if not plain and len(seq) < 5000 and rng.random() < 0.1:
            out.append("@%s%s%s%s+%s%s%s" % (name, rng.choice(["", " q"]), eol, seq + eol if seq else eol, eol,
                                         "".join(rng.choice("@>+IIIF#") for _ in seq), eol))
            continue
        w = rng.choice([50, 60, 70, 80, 1000, 10 ** 9])
        body = eol.join(seq[k:k + w] for k in range(0, len(seq), w))
        if not plain and rng.random() < 0.1 and len(seq) > w:
            body = body.replace(eol, eol + "\n", 1)              # an empty line inside the record
        out.append(">%s%s%s%s%s" % (name, rng.choice(["", "", " some comment", "\tx=1"]), eol, body, eol))
    text = "".join(out)
    if rng.random() < 0.1 and text.endswith(eol):
        text = text[:-len(eol)]                             # no line end after the last line of the file
    return text


def gen_paf(rng, targets, n):
    """PAF lines on the given (name, length) targets.  Inside what the reference defines: at least 12 columns, target
    coordinates within the target (+ its 500 cells of padding are never relied on), tstart >= 1 for lines without a
    cg:Z: tag (it starts counting one cell before tstart), well-formed cg:Z: values."""
    out = []
    big = [t for t in targets if t[1] >= 2]
    for i in range(n):
        name, L = rng.choice(big)
        a = rng.randrange(0, L); b = min(L, a + rng.randrange(1, 900))
        cols = ["q%d" % i, "5000", "10", "900", rng.choice("+-"), name, str(L)]
        tags = ["tp:A:%s" % rng.choice("PPPSI"), "cm:i:%d" % rng.randrange(100)]
        if rng.random() < 0.5:
            rem, ops = b - a, []
            while rem > 0:
                k = min(rem, rng.randrange(1, 300)); ops.append("%d%s" % (k, rng.choice("MMMM=XDN"))); rem -= k
                if rng.random() < 0.3:
                    ops.append("%d%s" % (rng.randrange(1, 40), rng.choice("IISH")))
            tags.insert(rng.randrange(len(tags) + 1), "cg:Z:" + "".join(ops))
            s, e = a, b
        else:
            s, e = max(a, 1), max(b, 1)
        if rng.random() < 0.05:
            s, e = e, s                                       # reversed coordinates are swapped back (PD:1570-1575)
        cols += [str(s), str(e), str(abs(e - s)), str(abs(e - s) + rng.randrange(50)), str(rng.choice([0, 1, 20, 60, 255]))]
        sep = "\t" if rng.random() < 0.95 else " "
        out.append(sep.join(cols + tags))
        if rng.random() < 0.03:
            out.append("")
    return "\n".join(out) + ("\n" if rng.random() < 0.9 else "")


def paf_case(rng, td):
    import gzip
    nt = rng.randrange(1, 5)
    targets = [("t%d" % k if rng.random() < 0.8 else "ctg_%d" % k, rng.choice([1, 2, 37, 150, 151, 600, 2500, 5000])) for k in range(nt)]
    if all(L < 2 for _, L in targets):
        targets.append(("tx", 800))
    names, lens = [t[0] for t in targets], [t[1] for t in targets]

    def put(fn, text):
        if rng.random() < 0.25:
            fn += ".gz"
            gzip.open(os.path.join(td, fn), "wb", compresslevel=1).write(text.encode())
        else:
            open(os.path.join(td, fn), "w").write(text)
        return fn
    first = put("a.paf", gen_paf(rng, targets, rng.randrange(0, 300)))
    args = ["-i", first]
    if rng.random() < 0.25:
        files = [first] + [put("b%d.paf" % k, gen_paf(rng, targets, rng.randrange(1, 200))) for k in range(rng.randrange(1, 3))]
        open(os.path.join(td, "p.list"), "w").write("\n".join(files) + "\n")
        args = ["-i", "p.list"]
    use_ref = rng.random() < 0.35
    if use_ref:
        # targets then come from the FASTA records (ids in file order); every PAF target is present with its length
        order = list(range(nt + (len(targets) - nt)))
        rng.shuffle(order)
        fa = "".join(">%s\n%s\n" % (names[t], "".join(rng.choice("ACGTacgtN") for _ in range(lens[t]))) for t in order)
        if rng.random() < 0.3:
            fa += ">spare\nACGTACGTAC\n"
        put_name = "genome.fa"
        open(os.path.join(td, put_name), "w").write(fa)
        args += ["-r", put_name, "-c"]                        # (-r without -c: its GC column reads strings it never filled)
        names = [names[t] for t in order]; lens = [lens[t] for t in order]
    mode = rng.choice(["chr", "chr", "w", "w", "gff", "bed3", "bed4"])
    if mode == "w":
        ws = [150, 151, 200, 1000, 5000] if use_ref else [1, 7, 50, 100, 149, 150, 151, 200, 1000, 0]
        args += ["-w", str(rng.choice(ws))]
    elif mode in ("gff", "bed3", "bed4"):
        # every target keeps at least one target row: a target without rows gets a 500-cell array in the reference and
        # its alignments run off the end of it
        text = gen_regions(rng, names, lens, mode).rstrip("\n").split("\n")
        for nm, L in zip(names, lens):
            a = rng.randrange(1, L + 1); b = rng.randrange(a, L + 1)
            if mode == "gff":
                text.append("%s\tsrc\tCDS\t%d\t%d\t.\t+\t0\tID=k;Parent=keep_%s" % (nm, a, b, nm))
            elif mode == "bed3":
                text.append("%s\t%d\t%d" % (nm, a, b))
            else:
                text.append("%s\t%d\t%d\tkeep_%s" % (nm, a, b, nm))
        rng.shuffle(text)
        if mode == "bed4":
            text = [l for l in text if len(l.split()) == 4] or text
        fn = put("r." + ("gff" if mode == "gff" else "bed"), "\n".join(text) + "\n")
        args += ["-g" if mode == "gff" else "-b", fn]
    if rng.random() < 0.3:
        args += ["-a"]
    if rng.random() < 0.3:
        args += ["-q", str(rng.choice([0, 1, 20, 61]))]
    if rng.random() < 0.3:
        args += ["-d", str(rng.choice([0, 1, 2, 5]))]
    if rng.random() < 0.4:
        args += ["-x", str(rng.choice([0, 256, 1796, 4]))]
    if rng.random() < 0.3:
        args += ["-t", str(rng.choice([1, 4]))]
    return args


S2B = os.path.join(ROOT, "oracle", "_ref", "sam2bam")


def one_case(rng, td):
    sorted_hdr = rng.random() < 0.75
    lie = sorted_hdr and rng.random() < 0.2
    sam, names, lens = gen_sam(rng, sorted_hdr, lie)
    open(os.path.join(td, "x.sam"), "w").write(sam)
    args = ["-i", "x.sam"]
    form = rng.choice(["sam", "bam+bai", "bam+bai", "bam", "list", "cram", "cram+crai"])
    if form.startswith("cram") and os.access(S2B, os.X_OK):
        # CRAM 3.0 written reference-free by the reference's own htslib; with a .crai next to it the reference takes its
        # indexed path
        indexed = form == "cram+crai" and sorted_hdr and not lie
        r = subprocess.run([S2B, "x.sam", "x.cram"] + ([] if indexed else ["noindex"]) + ([rng.choice(["fmt=cram,version=3.1", "fmt=cram,version=3.1", "fmt=cram,version=2.1"])] if rng.random() < 0.5 else [])
                           + (["sps=%d" % rng.choice([3, 50])] if rng.random() < 0.3 and not indexed else []), cwd=td, capture_output=True)
        # (tiny slices only without a .crai: htslib's slice lookup, cram_index_query, walks back only while the PREVIOUS slice
        # still reaches the target, so with a .crai and GFF / BED targets the reference misses a read from an earlier slice
        # that spans into the target — it finds that read in the same data written as BAM, and so does this reader in the
        # CRAM; tests/test_cram.py pins that)
        if r.returncode == 0:
            args = ["-i", "x.cram"]
    elif form.startswith("bam") and os.access(S2B, os.X_OK):
        indexed = form == "bam+bai" and sorted_hdr and not lie
        r = subprocess.run([S2B, "x.sam", "x.bam"] + ([] if indexed else ["noindex"]), cwd=td, capture_output=True)
        if r.returncode == 0:
            args = ["-i", "x.bam"]
    elif form == "list" and os.access(S2B, os.X_OK):
        # two or three files with the same contigs (list mode takes them from the first): other reads, mixed formats
        files = ["x.sam"]
        hdr = [l for l in sam.split("\n") if l.startswith("@")]
        for k in range(rng.randrange(1, 3)):
            other, _, _ = gen_sam(rng, sorted_hdr)
            body = [l for l in other.split("\n") if l and not l.startswith("@")]
            ok = []
            for l in body:                               # keep the reads that fit THESE contigs
                f = l.split("\t")
                if f[2] in names:
                    L = lens[names.index(f[2])]
                    import re
                    rl = sum(int(n) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", f[5]) if o in "MDN=X")
                    if int(f[3]) + rl - 1 <= L:
                        ok.append(l)
            if sorted_hdr:
                ok.sort(key=lambda l: (names.index(l.split("\t")[2]), int(l.split("\t")[3])))
            fn = "y%d.sam" % k
            open(os.path.join(td, fn), "w").write("\n".join(hdr + ok) + "\n")
            if rng.random() < 0.5:
                ext = rng.choice(["bam", "bam", "cram"])
                r = subprocess.run([S2B, fn, "y%d.%s" % (k, ext)] + ([] if sorted_hdr and rng.random() < 0.7 else ["noindex"]), cwd=td, capture_output=True)
                if r.returncode == 0:
                    fn = "y%d.%s" % (k, ext)
            files.append(fn)
        open(os.path.join(td, "in.list"), "w").write("\n".join(files) + "\n")
        args = ["-i", "in.list"]
    mode = rng.choice(["chr", "chr", "w", "w", "gff", "gtf", "bed3", "bed4"])
    if mode == "w":
        args += ["-w", str(rng.choice([1, 2, 7, 50, 100, 149, 150, 151, 200, 1000, 0, -3]))]
    elif mode in ("gff", "gtf"):
        fn = "r." + mode
        text = gen_regions(rng, names, lens, mode)
        if rng.random() < 0.2:                            # the reference reads its target files through gzstream
            import gzip
            fn += ".gz"
            gzip.open(os.path.join(td, fn), "wb").write(text.encode())
        else:
            open(os.path.join(td, fn), "w").write(text)
        args += ["-g", fn]
        if rng.random() < 0.3:
            args += ["-f", rng.choice(["exon", "CDS", "gene"])]
    elif mode in ("bed3", "bed4"):
        fn = "r.bed"
        text = gen_regions(rng, names, lens, mode)
        if rng.random() < 0.2:
            import gzip
            fn += ".gz"
            gzip.open(os.path.join(td, fn), "wb").write(text.encode())
        else:
            open(os.path.join(td, fn), "w").write(text)
        args += ["-b", fn]
    if rng.random() < 0.3:
        args += ["-a"]
    if rng.random() < 0.3:
        args += ["-q", str(rng.choice([0, 1, 10, 20, 61]))]
    if rng.random() < 0.3:
        args += ["-d", str(rng.choice([0, 1, 2, 5, 50]))]
    if rng.random() < 0.3:
        args += ["-x", str(rng.choice([0, 4, 1024, 1796, 3844]))]
    if rng.random() < 0.15:
        args += ["-s"]
    if rng.random() < 0.5:
        args += ["-t", str(rng.choice([1, 2, 5]))]
    # -c -r: the GC(%) column.  Not with a window below 150: the reference frees its sequences before that sweep reads them
    # (PD:4097 / PD:4327) and prints what the heap holds.
    small_w = "-w" in args and int(args[args.index("-w") + 1]) < 150
    if rng.random() < 0.25 and not small_w:
        import gzip
        # CRAM input: -r also goes to htslib's faidx, which wants an indexable file (uniform lines, no FASTQ records, not
        # gzip) and then lets the FASTA's sequence lengths override the header's LN values
        cram = args[1].endswith(".cram") or (args[1].endswith(".list") and open(os.path.join(td, args[1])).read().split("\n")[0].endswith(".cram"))
        fa = gen_fasta(rng, names, lens, plain=cram)
        fn = "genome.fa"
        if rng.random() < 0.25 and not cram:
            fn += ".gz"
            gzip.open(os.path.join(td, fn), "wb", compresslevel=1).write(fa.encode())
        else:
            open(os.path.join(td, fn), "wb").write(fa.encode())
        args += ["-r", fn] + (["-c"] if rng.random() < 0.9 else [])
    elif rng.random() < 0.03:
        args += ["-c"]                                    # -c without -r: the message, rc 0, no output
    return args


def run(exe, args, td, prefix):
    p = subprocess.run([exe] + args + ["-o", prefix], cwd=td, capture_output=True, timeout=120)
    files = {}
    for f in sorted(glob.glob(os.path.join(td, prefix + ".*"))):
        files[os.path.basename(f)[len(prefix):]] = open(f, "rb").read()
    return p.returncode, p.stdout, files


def args_mode(seed, cases):
    """random argv over the golden fixture f1: flags in any order, missing values, unknown flags, unreadable files; compares
    exit code, stderr, stdout (the usage text counts as equal: this build does not list cram/paf) and every output file"""
    rng = random.Random(seed)
    # a scratch COPY of the fixture: `-o <value>` writes next to the inputs
    d = tempfile.mkdtemp(prefix="af1", dir="/tmp")
    src = os.path.join(ROOT, "tests", "golden", "f1")
    for fn in os.listdir(src):
        if os.path.isfile(os.path.join(src, fn)):
            shutil.copy(os.path.join(src, fn), d)
    flags = ["-i", "-o", "-g", "-f", "-b", "-w", "-a", "-q", "-d", "-x", "-t", "-h", "-s", "-c", "-r", "-z", "--help", "-W", "-I", "-O"]
    vals = ["f1.bam", "f1.sam", "f1.gff", "f1.gtf", "f1.bed3", "f1.bed4", "f1_3.list", "missing.bam", "100", "0", "-5", "abc", "1e3", "CDS",
            "exon", "", "3.7", "200", "1796", "60"]
    bad = 0
    for k in range(cases):
        args = []
        if rng.random() < 0.8:
            args += ["-i", rng.choice(["f1.bam", "f1.sam", "f1_3.list", "missing.bam"])]
        for _ in range(rng.randrange(0, 7)):
            args.append(rng.choice(flags))
            if rng.random() < 0.75:
                args.append(rng.choice(vals))
        td = tempfile.mkdtemp(prefix="af", dir="/tmp")
        res = []
        for exe, pre in ((REF, "r"), (CLI, "m")):
            full = [exe] + args + ([] if "-o" in args else ["-o", os.path.join(td, pre)])
            try:
                p = subprocess.run(full, cwd=d, capture_output=True, timeout=60)
            except subprocess.TimeoutExpired:
                res.append(None); continue
            files = {os.path.basename(f)[1:]: open(f, "rb").read() for f in glob.glob(os.path.join(td, pre + ".*"))}
            out = b"<usage>" if p.stdout.startswith(b"Usage: pandepth") else p.stdout
            res.append((p.returncode, out, p.stderr.replace(pre.encode() + b".", b"X."), files))
        shutil.rmtree(td)
        if res[0] is None or res[0][0] < 0:
            continue
        if res[0] != res[1]:
            bad += 1
            print("MISMATCH argv case %d: %s\n  ref %r\n  mine %r" % (k, " ".join(args), res[0][:3], res[1][:3] if res[1] else None), flush=True)
    shutil.rmtree(d, ignore_errors=True)
    print("seed %d: %d argv cases, %d mismatches" % (seed, cases, bad))
    return 1 if bad else 0


def oracle_mode(seed, cases, paf=False):
    """the Python/C restatement (oracle/pd_oracle.py: run()) against the reference binary on the same random inputs: every
    output file's TEXT (the oracle does not deflate)"""
    import gzip
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pd_oracle as O
    rng = random.Random(seed)
    bad = skipped = 0
    for k in range(cases):
        td = tempfile.mkdtemp(prefix="fo", dir="/tmp")
        args = paf_case(rng, td) if paf else one_case(rng, td)
        try:
            p = subprocess.run([REF] + args + ["-o", "ref"], cwd=td, capture_output=True, timeout=120)
        except subprocess.TimeoutExpired:
            skipped += 1; shutil.rmtree(td); continue
        if p.returncode < 0:
            skipped += 1; shutil.rmtree(td); continue
        ref = {os.path.basename(x)[4:]: gzip.decompress(open(x, "rb").read()).decode() for x in glob.glob(os.path.join(td, "ref.*"))}
        try:
            mine = O.run(args + ["-o", "x"], cwd=td)
        except Exception as e:                                   # noqa: BLE001 - reported as a mismatch
            mine = {"error": repr(e)}
        if mine != ref:
            bad += 1
            keep = "/tmp/oraclebad_%d_%d" % (seed, k)
            shutil.rmtree(keep, ignore_errors=True)
            shutil.copytree(td, keep)
            print("ORACLE MISMATCH case %d: %s -> %s" % (k, " ".join(args), keep), flush=True)
        shutil.rmtree(td)
    print("seed %d: %d oracle cases, %d mismatches, %d skipped" % (seed, cases, bad, skipped))
    return 1 if bad else 0


def main():
    global BIG, MESSY
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    PAF = len(sys.argv) > 3 and sys.argv[3] == "paf"
    BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
    MESSY = len(sys.argv) > 3 and sys.argv[3] == "messy"
    if len(sys.argv) > 3 and sys.argv[3] == "args":
        return args_mode(seed, cases)
    if len(sys.argv) > 3 and sys.argv[3] in ("oracle", "oracle-paf"):
        return oracle_mode(seed, cases, sys.argv[3] == "oracle-paf")
    rng = random.Random(seed)
    bad = skipped = 0
    for k in range(cases):
        td = tempfile.mkdtemp(prefix="fz", dir="/tmp")
        args = paf_case(rng, td) if PAF else one_case(rng, td)
        try:
            r_rc, r_out, r_files = run(REF, args, td, "ref")
        except subprocess.TimeoutExpired:
            skipped += 1; shutil.rmtree(td); continue
        if r_rc < 0:                                    # the reference itself crashed on this input
            skipped += 1; shutil.rmtree(td); continue
        m_rc, m_out, m_files = run(CLI, args, td, "mine")
        same = r_rc == m_rc and r_out == m_out and r_files == m_files
        if not same:
            bad += 1
            keep = "/tmp/fuzzbad_%d_%d" % (seed, k)
            shutil.rmtree(keep, ignore_errors=True)
            shutil.copytree(td, keep)
            what = "rc %d/%d" % (r_rc, m_rc) if r_rc != m_rc else "stdout" if r_out != m_out else "files " + ",".join(
                sorted(set(r_files) ^ set(m_files)) or [f for f in r_files if r_files[f] != m_files[f]])
            print("MISMATCH case %d: %s  args %s  -> %s" % (k, what, " ".join(args), keep), flush=True)
        shutil.rmtree(td)
    print("seed %d: %d cases, %d mismatches, %d skipped (reference crashed / hung)" % (seed, cases, bad, skipped))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
