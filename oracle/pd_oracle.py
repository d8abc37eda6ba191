"""oracle/pd_oracle.py — executable restatement of the reference's depth path (CPU).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The hot loops are the C functions of oracle/pd_oracle.c (loaded with ctypes);
this module adds what is needed to replay a whole reference invocation on a small fixture —
a BAM/SAM record reader, the region model, the read-selection rules and the table text — so
that the restatement can be pinned against the compiled reference's golden outputs
(tests/golden/manifest.json).  Citations "PD:n" are /root/reference/src/PanDepth.cpp:n.
"""
import ctypes
import gzip
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpd_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle port`")
        L = ctypes.CDLL(path)
        P = ctypes.c_void_p
        L.pdo_walk_records.restype = ctypes.c_int64
        L.pdo_walk_records.argtypes = [ctypes.c_int64, P, P, P, P, P, P, ctypes.c_uint32,
                                       ctypes.c_int32, P, P]
        L.pdo_add_intervals.restype = None
        L.pdo_add_intervals.argtypes = [ctypes.c_int64, P, P, P]
        L.pdo_wrap18.restype = None
        L.pdo_wrap18.argtypes = [P, ctypes.c_int64]
        L.pdo_stat_regions.restype = None
        L.pdo_stat_regions.argtypes = [P, P, ctypes.c_int64, P, ctypes.c_uint32, P, P]
        L.pdo_sweep_windows.restype = ctypes.c_int64
        L.pdo_sweep_windows.argtypes = [P, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32,
                                        P, P, P, P]
        L.pdo_endpos.restype = ctypes.c_int32
        L.pdo_endpos.argtypes = [ctypes.c_int32, ctypes.c_uint16, ctypes.c_int64, P]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------------------------
# numpy-level wrappers over the C restatement (used directly by the GPU parity tests)
# --------------------------------------------------------------------------------------------
PAD = 512   # cells after each contig (the reference pads +500, PD:4132)


def contig_offsets(lens, pad=PAD):
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    for i, l in enumerate(lens):
        off[i + 1] = off[i] + int(l) + pad
    return off


def depth_from_intervals(lens, iv, wrap18=False):
    """iv: int32 array (n,3) of {tid,beg,end}.  Returns (flat uint32 depth, offsets)."""
    off = contig_offsets(lens)
    d = np.zeros(int(off[-1]), dtype=np.uint32)
    iv = np.ascontiguousarray(iv, dtype=np.int32)
    lib().pdo_add_intervals(iv.shape[0], _p(iv), _p(d), _p(off))
    if wrap18:
        lib().pdo_wrap18(_p(d), d.size)
    return d, off


def stat_regions(depth, off, reg, min_dep):
    """reg: int32 (n,3) of {tid, first(1-based), second(inclusive)} -> (cover int32, sum uint64)"""
    reg = np.ascontiguousarray(reg, dtype=np.int32)
    n = reg.shape[0]
    cover = np.zeros(n, dtype=np.int32)
    tot = np.zeros(n, dtype=np.uint64)
    lib().pdo_stat_regions(_p(depth), _p(off), n, _p(reg), int(min_dep), _p(cover), _p(tot))
    return cover, tot


def sweep_windows(depth_contig, length, w, min_dep):
    nmax = max(1, (int(length) + int(w) - 1) // int(w) + 1)
    s1 = np.zeros(nmax, dtype=np.int32)
    e = np.zeros(nmax, dtype=np.int32)
    c = np.zeros(nmax, dtype=np.int32)
    t = np.zeros(nmax, dtype=np.int32)
    d = np.ascontiguousarray(depth_contig, dtype=np.uint32)
    k = lib().pdo_sweep_windows(_p(d), int(length), int(w), int(min_dep), _p(s1), _p(e), _p(c), _p(t))
    return s1[:k], e[:k], c[:k], t[:k]


# --------------------------------------------------------------------------------------------
# record input (test-side reader; BGZF is multi-member gzip so the gzip module inflates it)
# --------------------------------------------------------------------------------------------
class Records:
    def __init__(self):
        self.names, self.lens, self.text = [], [], ""
        self.tid, self.pos, self.flag, self.mapq = [], [], [], []
        self.cigars = []       # list of lists of packed uint32

    def arrays(self, sel=None):
        idx = range(len(self.tid)) if sel is None else sel
        tid = np.array([self.tid[i] for i in idx], dtype=np.int32)
        pos = np.array([self.pos[i] for i in idx], dtype=np.int32)
        flag = np.array([self.flag[i] for i in idx], dtype=np.uint16)
        mapq = np.array([self.mapq[i] for i in idx], dtype=np.uint8)
        off = np.zeros(len(tid) + 1, dtype=np.int64)
        flat = []
        for k, i in enumerate(idx):
            flat.extend(self.cigars[i])
            off[k + 1] = len(flat)
        cig = np.array(flat if flat else [0], dtype=np.uint32)
        return tid, pos, flag, mapq, off, cig


_CIG = {c: i for i, c in enumerate("MIDNSHP=XB")}


def _cram_text(path):
    """CRAM decoding is NOT restated here (the product's reader is pinned against the SAM text by tests/test_cram.py and
    against the reference's outputs by the golden cases): a .cram input is replaced by the SAM it was written from, which
    the fixtures keep next to it as <stem>.sam (m_noidx.cram, m_ref.cram, m_v31.cram, m_v21.cram -> m.sam)."""
    stem = path[:-5]
    for cand in (stem + ".sam", os.path.join(os.path.dirname(stem), os.path.basename(stem).split("_")[0] + ".sam")):
        if os.path.exists(cand):
            return cand
    raise NotImplementedError("no SAM text next to " + path)


def read_alignments(path):
    if path.endswith(".cram"):
        path = _cram_text(path)
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    r = Records()
    if raw[:4] == b"BAM\x01":
        l_text, = struct.unpack_from("<i", raw, 4)
        r.text = raw[8:8 + l_text].decode().rstrip("\0")
        o = 8 + l_text
        n_ref, = struct.unpack_from("<i", raw, o)
        o += 4
        for _ in range(n_ref):
            l_name, = struct.unpack_from("<i", raw, o)
            r.names.append(raw[o + 4:o + 4 + l_name - 1].decode())
            l_ref, = struct.unpack_from("<i", raw, o + 4 + l_name)
            r.lens.append(l_ref)
            o += 8 + l_name
        while o < len(raw):
            bs, tid, pos, l_rn, mq, _bin, n_cig, flag = struct.unpack_from("<iiiBBHHH", raw, o)
            co = o + 36 + l_rn
            r.tid.append(tid); r.pos.append(pos); r.flag.append(flag); r.mapq.append(mq)
            r.cigars.append(list(struct.unpack_from("<%dI" % n_cig, raw, co)))
            o += 4 + bs
    else:
        hdr = []
        for line in raw.decode().split("\n"):
            if not line:
                continue
            if line[0] == "@":
                hdr.append(line)
                if line.startswith("@SQ"):
                    f = dict(x.split(":", 1) for x in line.split("\t")[1:])
                    r.names.append(f["SN"]); r.lens.append(int(f["LN"]))
                continue
            f = line.split("\t")
            r.tid.append(r.names.index(f[2]) if f[2] != "*" else -1)
            r.pos.append(int(f[3]) - 1); r.flag.append(int(f[1])); r.mapq.append(int(f[4]))
            cg, n = [], ""
            if f[5] != "*":
                for ch in f[5]:
                    if ch.isdigit():
                        n += ch
                    else:
                        cg.append((int(n) << 4) | _CIG[ch]); n = ""
            r.cigars.append(cg)
        r.text = "\n".join(hdr) + "\n"
    return r


def endpos(r, i):
    cg = np.array(r.cigars[i] if r.cigars[i] else [0], dtype=np.uint32)
    return lib().pdo_endpos(r.pos[i], r.flag[i], len(r.cigars[i]), _p(cg))


# --------------------------------------------------------------------------------------------
# region model  (PD:3547-4051)
# --------------------------------------------------------------------------------------------
class Gene:
    __slots__ = ("start", "end", "length", "cds", "cover", "depth", "gc")

    def __init__(self, s, e):
        self.start, self.end, self.length, self.cds = s, e, e - s + 1, [(s, e)]
        self.cover, self.depth = 0, 0
        self.gc = 0


_GC = None                                                  # {tid: bytes} while a -c run builds its regions


def read_fasta(path, chrmap, keep_all=False):
    """PD:3506-3529 (klib kseq records): '>' or '@' starts a record, the name ends at the first white space, sequence
    lines are joined until a line starts with '>', '@' or '+'; '+' opens a quality block as long as the sequence.
    Returns {tid: sequence}; a name the header does not know becomes contig 0 IN `chrmap` (map::operator[]), and the
    first sequence claiming an id wins."""
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    lines = raw.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    seqs, k, n = {}, 0, len(lines)
    while k < n and not (lines[k][:1] in (b">", b"@")):    # (the reader would also accept a marker in mid-line here)
        k += 1
    while k < n:
        name = lines[k][1:].split()[0].decode() if lines[k][1:].split() else ""
        k += 1
        parts = []
        while k < n and lines[k][:1] not in (b">", b"@", b"+"):
            ln = lines[k]
            if len(ln) > 1 and ln.endswith(b"\r"):
                ln = ln[:-1]
            parts.append(ln)
            k += 1
        seq = b"".join(parts)
        if k < n and lines[k][:1] == b"+":
            k += 1
            q = 0
            while k < n and q < len(seq):
                ln = lines[k]
                if len(ln) > 1 and ln.endswith(b"\r"):
                    ln = ln[:-1]
                q += len(ln)
                k += 1
            if q != len(seq):
                break                                       # truncated quality: the reader stops here
            while k < n and not (lines[k][:1] in (b">", b"@")):
                k += 1
        if keep_all:
            seqs.setdefault("all", []).append((name, seq))
            continue
        tid = chrmap.setdefault(name, 0)
        seqs.setdefault(tid, seq)
    return seqs.get("all", []) if keep_all else seqs


def _gc_count(tid, s, e):                                   # PD:3536-3538 + the loops `for (ii = Start-1; ii < End; ii++)`
    b = _GC.get(tid, b"")[max(s - 1, 0):max(e, 0)]
    return b.count(b"C") + b.count(b"c") + b.count(b"G") + b.count(b"g")


def _add(genes, tid, gid, s, e):
    g = genes.setdefault(tid, {})
    if gid not in g:
        g[gid] = Gene(s, e)
        if _GC is not None:                                 # only the entry that creates the id is counted (PD:3605-3611)
            g[gid].gc = _gc_count(tid, s, e)
    else:
        x = g[gid]
        x.start = min(x.start, s); x.end = max(x.end, e)
        x.length += e - s + 1; x.cds.append((s, e))


def _lines(path):
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return raw.decode().split("\n")


def parse_regions(path, mode, names, feature, chrmap=None):
    """mode 1 GFF, 2 GTF, 3 BED3, 4 BED4.  Plain well-formed inputs only."""
    genes = {}
    if chrmap is None:
        chrmap = {}
        for i, n in enumerate(names):
            chrmap.setdefault(n, i)
    for line in _lines(path):
        if not line or line[0] == "#":
            continue
        if mode == 1:                                     # PD:3557-3647
            f = line.split()
            if len(f) < 9 or f[2] != feature:
                continue
            s, e = int(f[3]), int(f[4])
            inf = [x for x in f[8].replace(",", ";").split(";") if x]
            gid = inf[0].split("=")[-1]
            for a in inf[1:]:
                kv = [x for x in a.split("=") if x]
                if kv[0] == "Parent":
                    gid = kv[-1]
            if f[0] in chrmap:
                _add(genes, chrmap[f[0]], gid, s, e)
        elif mode == 2:                                   # PD:3649-3740
            line = line.replace('"', "").replace(";", "")
            f = line.split()
            if len(f) < 10 or f[2] != feature:
                continue
            if f[0] in chrmap:
                _add(genes, chrmap[f[0]], f[9], int(f[3]), int(f[4]))
        elif mode == 3:                                   # PD:3741-3819
            f = line.split()
            s, e = int(f[1]), int(f[2])
            if s > e:
                continue
            if f[0] in chrmap:
                _add(genes, chrmap[f[0]], f[0] + "_" + f[1] + "_" + f[2], s, e)
        elif mode == 4:                                   # PD:3821-3898
            f = line.split()
            s, e = int(f[1]), int(f[2])
            if s > e:
                continue
            if f[0] in chrmap:
                _add(genes, chrmap[f[0]], f[3], s, e)
    return genes


def merge_regions(genes):                                  # PD:3912-3972
    merged = {}
    for tid, g in genes.items():
        spans = {}
        for x in g.values():
            spans[x.start] = max(spans.get(x.start, x.end), x.end)
        out = []
        for s in sorted(spans):
            e = spans[s]
            if not out or s > out[-1][1]:
                out.append([s, e])
            elif e > out[-1][1]:
                out[-1][1] = e
        merged[tid] = [tuple(x) for x in out]
    return merged


def synth_bins(names, lens, width):                        # PD:3995-4049
    genes = {}
    for tid, ln in enumerate(lens):
        start, end = 1, 2
        while end <= ln:
            end = min(start + width - 1, ln)
            _add(genes, tid, names[tid] + str(start), start, end)
            end += 2
            start += width
    return genes


# --------------------------------------------------------------------------------------------
# one whole reference invocation
# --------------------------------------------------------------------------------------------
def _sniff_gff(path):                                      # PD:162-181
    mode = 0
    for k, line in enumerate(_lines(path)[:167]):
        if len(line) < 2 or line[0] == "#":
            continue
        if "Parent" in line:
            mode = 1
        elif "transcript_id" in line:
            mode = 2
    return mode


def _sniff_bed(path):                                      # PD:268-288
    ls = _lines(path) + ["", ""]
    return 4 if (len(ls[0].split()) == 4 or len(ls[1].split()) == 4) else 3


def _sorted_header(text):                                  # PD:4537-4549
    p = text.find("\tSO:")
    if p < 0:
        return False
    p += 4
    q = len(text)
    for ch in "\n\t":
        k = text.find(ch, p)
        if k >= 0:
            q = min(q, k)
    return text[p:q] == "coordinate"


def _select_indexed(r, merged, lens):                      # PD:419-434 (htslib multi-region fetch)
    sel = []
    for i in range(len(r.tid)):
        t = r.tid[i]
        if t not in merged:
            continue
        p, q = r.pos[i], endpos(r, i)
        for s, e in merged[t]:
            beg = max(s - 1, 1); end = min(e + 1, lens[t])
            if p < end and q > beg - 1:
                sel.append(i)
                break
    return sel


def _select_sorted_stream(r, merged, flag_mask, min_mapq):  # PD:4608-4646
    sel = []
    it = {t: 0 for t in merged}
    done = {t: False for t in range(len(r.lens))}
    for t in range(len(r.lens)):
        if t not in merged:
            done[t] = True
    for i in range(len(r.tid)):
        t = r.tid[i]
        if t < 0 or done[t]:
            continue
        if r.mapq[i] < min_mapq or (r.flag[i] & flag_mask):
            continue
        regs = merged[t]
        if endpos(r, i) < regs[it[t]][0]:
            continue
        if r.pos[i] > regs[it[t]][1]:
            it[t] += 1
            while it[t] < len(regs) and not (r.pos[i] <= regs[it[t]][1]):
                it[t] += 1
            if it[t] == len(regs):
                done[t] = True
                if all(done.values()):
                    break
                # the reference still counts THIS read (PD:4648 runs after the region advance)
                it[t] = len(regs) - 1
                sel.append(i)
                continue
        sel.append(i)
    return sel


def run(args, cwd="."):
    """Replay `pandepth <args>`; returns {suffix: text} for every file the reference writes."""
    o = {"i": None, "g": None, "b": None, "f": "CDS", "w": None, "a": False, "q": -1, "d": 1,
         "x": 1796, "s": False, "c": False, "r": None}
    k = 0
    while k < len(args):
        f = args[k].replace("-", "")
        if f in ("a", "s", "c"):
            o[f] = True
        elif f in ("w", "q", "d", "x", "t"):
            o[f] = int(args[k + 1]); k += 1
        else:
            o[f] = args[k + 1]; k += 1
        k += 1
    # PD:211-215: a given -w below 1 is warned about and becomes 1 (so `-w 0` is still window mode); not given = 0
    o["w"] = 0 if o["w"] is None else max(o["w"], 1)
    o["d"] = max(o["d"], 1)
    path = os.path.join(cwd, o["i"])
    is_list = path.endswith(".list") or path.endswith(".List")
    files = [os.path.join(cwd, l) for l in _lines(path) if l] if is_list else [path]
    is_list = len(files) > 1
    if is_paf(files[0]):                                    # PD:3466-3479: the first input's extension decides
        return _run_paf_front(o, cwd, files, is_list)
    first = read_alignments(files[0])
    names, lens = first.names, list(first.lens)
    if files[0].endswith(".cram") and o["r"]:
        # htslib gets -r for CRAM input (PD:3486-3492) and rewrites every @SQ LN that differs from the FASTA's sequence
        fa = {}
        for nm, sq in read_fasta(os.path.join(cwd, o["r"]), {}, keep_all=True):
            fa.setdefault(nm, len(sq))
        lens = [fa.get(n, l) for n, l in zip(names, lens)]

    mode = 0
    if o["g"]:
        mode = _sniff_gff(os.path.join(cwd, o["g"]))
    elif o["b"]:
        mode = _sniff_bed(os.path.join(cwd, o["b"]))
    global _GC
    chrmap = {}
    for i, n in enumerate(names):
        chrmap.setdefault(n, i)
    if o["c"] and not o["r"]:
        return {}                                           # PD:3530-3533: "lack reference sequence", exit 0, no output
    gc = bool(o["c"])
    _GC = read_fasta(os.path.join(cwd, o["r"]), chrmap) if gc else None
    try:
        return _run(o, cwd, files, is_list, first, names, lens, mode, chrmap, gc)
    finally:
        _GC = None


def is_paf(path):
    if path.endswith(".gz"):
        path = path[:-3]
    return path.endswith(".paf") or path.endswith(".PAF")


def _run_paf_front(o, cwd, files, is_list):
    """paf_main (PD:852-2024): targets from -r's records (ids in file order, the GC column then always on) or from
    columns 6/7 of the first file; depth from [tstart-1, tend) or the cg:Z: walk; 18-bit cells; the shared tables."""
    global _GC
    if o["c"] and not o["r"]:
        return {}                                           # PD:909-913
    names, lens, chrmap, seqs = [], [], {}, {}
    if o["r"]:
        tmp = {}
        raw = read_fasta(os.path.join(cwd, o["r"]), tmp, keep_all=True)
        for tid, (nm, sq) in enumerate(raw):
            chrmap[nm] = tid; names.append(nm); lens.append(len(sq)); seqs[tid] = sq
    else:
        for line in _lines(files[0]):
            f = line.split()
            if len(f) >= 7 and f[5] not in chrmap:          # (well-formed lines only)
                chrmap[f[5]] = len(names); names.append(f[5]); lens.append(int(f[6]))
    mode = 0
    if o["g"]:
        mode = _sniff_gff(os.path.join(cwd, o["g"]))
    elif o["b"]:
        mode = _sniff_bed(os.path.join(cwd, o["b"]))
    gc = bool(o["r"])
    if gc and not o["c"]:
        raise NotImplementedError("PAF with -r but without -c: the reference counts G/C in strings it never filled")
    _GC = seqs if gc else None
    try:
        return _run(o, cwd, files, is_list, None, names, lens, mode, chrmap, gc, paf=True)
    finally:
        _GC = None


def _paf_depth(files, o, chrmap, depth, off, lens):           # PD:1532-1616
    for fp in files:
        for line in _lines(fp):
            if not line:
                continue
            if (o["x"] & 0x100) and "tp:A:S" in line:
                continue
            f = line.replace("\t", " ").split(" ")
            f = [x for x in f if x]
            tid = chrmap.setdefault(f[5], 0)
            if int(f[11]) < o["q"]:
                continue
            s, e = int(f[7]), int(f[8])
            if s > e:
                s, e = e, s
            cg = [k for k, x in enumerate(f) if x.startswith("cg:Z:")]
            base = int(off[tid])
            if cg and cg[0] > 1:
                n = ""
                for ch in f[cg[0]][5:]:
                    if ch.isdigit():
                        n += ch
                        continue
                    k = int(n); n = ""
                    if ch in "M=X":
                        depth[base + s:base + s + k] += 1       # (inside the target for the inputs replayed here)
                        s += k
                    elif ch in "DN":
                        s += k
            else:
                depth[base + max(s - 1, 0):base + e] += 1


def _run(o, cwd, files, is_list, first, names, lens, mode, chrmap, gc, paf=False):
    genes = parse_regions(os.path.join(cwd, o["g"] or o["b"]), mode, names, o["f"], chrmap) if mode else {}
    merged = merge_regions(genes)
    if not merged:
        if o["w"] == 0:
            mode, width = 0, 10000000
        elif o["w"] < 150:
            mode, width = 6, 10000000
        else:
            mode, width = 5, o["w"]
        genes = synth_bins(names, lens, width)
        merged = merge_regions(genes)

    off = contig_offsets(lens)
    depth = np.zeros(int(off[-1]), dtype=np.uint32)
    wrap = is_list
    if paf:
        _paf_depth(files, o, chrmap, depth, off, lens)
        wrap = True
    for fp in ([] if paf else files):
        r = first if fp == files[0] else read_alignments(fp)
        has_index = any(os.path.exists(fp + e) for e in (".bai", ".csi", ".crai")) and not o["s"]
        if has_index:
            sel = _select_indexed(r, merged, lens)
            if o["a"] or mode == 6:
                wrap = True
        elif _sorted_header(r.text):
            sel = _select_sorted_stream(r, merged, o["x"], o["q"])
            wrap = True
        else:
            sel = [i for i in range(len(r.tid)) if r.tid[i] >= 0]
            wrap = True
        tid, pos, flag, mapq, coff, cig = r.arrays(sel)
        lib().pdo_walk_records(len(tid), _p(tid), _p(pos), _p(flag), _p(mapq), _p(coff), _p(cig),
                               o["x"], o["q"], _p(depth), _p(off))
    if wrap:
        lib().pdo_wrap18(_p(depth), depth.size)

    out = {}
    if o["a"]:
        rows = []
        for t in sorted(merged):
            d = depth[off[t]:off[t] + lens[t]]
            rows.append("".join("%s\t%d\t%d\n" % (names[t], j, d[j]) for j in range(lens[t])))
        out["SiteDepth.gz"] = "".join(rows)

    def footer(L, C, D, G=0):
        if not L:      # PD:5122 divides 0.0 by 0.0: x86 SSE gives the NaN with the sign bit set, printf prints "-nan"
            return "##RegionLength: 0\tCoveredSite: %d\t%sCoverage(%%): -nan\tMeanDepth: -nan\n" % (
                C, "GC(%): -nan\t" if gc else "")
        cov = C * 100.0 / L
        mean = D * 1.0 / L
        return "##RegionLength: %d\tCoveredSite: %d\t%sCoverage(%%): %.2f\tMeanDepth: %.2f\n" % (
            L, C, "GC(%%): %.2f\t" % (G * 100.0 / L) if gc else "", cov, mean)      # PD:5005

    def gccol(G, L):                                        # the extra column sits after TotalDepth (PD:4095-4114)
        return "%.2f\t" % (G * 100.0 / L) if gc else ""

    hgc = "GC(%)\t" if gc else ""
    SL = SC = SD = SG = 0
    if mode == 6 and gc:
        # PD:4097 drops the sequences before PD:4327 indexes them: the reference prints whatever memory follows
        raise NotImplementedError("-c with -w < 150: the reference's GC column is read from freed memory")
    if mode == 6:
        txt = "#Chr\tStart\tEnd\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n"
        for t in sorted(merged):
            s1, e, c, d = sweep_windows(depth[off[t]:off[t] + lens[t]], lens[t], o["w"], o["d"])
            for k in range(len(s1)):
                L = int(e[k]) - int(s1[k]) + 1
                txt += "%s\t%d\t%d\t%d\t%d\t%d\t%.2f\t%.2f\n" % (names[t], s1[k], e[k], L, c[k], d[k],
                                                               c[k] * 100.0 / L, d[k] * 1.0 / L)
                SL += L; SC += int(c[k]); SD += int(d[k])
        out["win.stat.gz"] = txt + footer(SL, SC, SD)
        return out

    reg, owner = [], []
    for t in sorted(genes):
        for gid in sorted(genes[t], key=lambda s: s.encode()):
            for (s, e) in genes[t][gid].cds:
                reg.append((t, s, e)); owner.append((t, gid))
    cover, tot = stat_regions(depth, off, np.array(reg, dtype=np.int32).reshape(-1, 3), o["d"])
    for k, (t, gid) in enumerate(owner):
        genes[t][gid].cover += int(cover[k]); genes[t][gid].depth += int(tot[k])
    if not wrap:
        # PD:676-786 + PD:295-309: the indexed uint32 path walks the merged spans of a contig in windows
        # [MeMStart, MeMEnd] (10 Mb of spans each; MeMEnd = last span end + 1, clipped to the contig length) and gives a
        # window's cells only to the genes with GeneStart < MeMEnd and GeneEnd >= MeMStart ("<=" when the window is a single
        # position).  A gene whose span starts on the contig's last base is selected by no window unless its window
        # starts there too: it keeps cover 0 / depth 0.
        for t in sorted(genes):
            spans, clen = merged[t], lens[t]
            wins = []
            ms = max(1, spans[0][0]); me = min(ms + 10000000 - 1, clen)
            for k, (s0, e0) in enumerate(spans):
                end = min(e0 + 1, clen)
                last = k + 1 == len(spans)
                if end >= me or last:
                    me = end
                    wins.append((ms, me))
                    if not last:
                        ms = spans[k + 1][0]
                        if ms - 150 > me:
                            ms -= 150
                    me = min(ms + 10000000, clen)
            for g in genes[t].values():
                sel = False
                for (a, b) in wins:
                    out_ = (g.start >= b or g.end < a) if b != a else (g.start > b or g.end < a)
                    sel = sel or not out_
                if not sel:
                    g.cover = 0; g.depth = 0

    if mode == 0:
        txt = "#Chr\tLength\tCoveredSite\tTotalDepth\t%sCoverage(%%)\tMeanDepth\n" % hgc
        for t in sorted(genes):
            L = sum(g.length for g in genes[t].values())
            C = sum(g.cover for g in genes[t].values())
            D = sum(g.depth for g in genes[t].values())
            G = sum(g.gc for g in genes[t].values())
            txt += "%s\t%d\t%d\t%d\t%s%.2f\t%.2f\n" % (names[t], L, C, D, gccol(G, L), C * 100.0 / L, D * 1.0 / L)
            SL += L; SC += C; SD += D; SG += G
        out["chr.stat.gz"] = txt + footer(SL, SC, SD, SG)
        return out

    if mode == 5:
        txt = "#Chr\tStart\tEnd\tLength\tCoveredSite\tTotalDepth\t%sCoverage(%%)\tMeanDepth\n" % hgc
        suffix = "win.stat.gz"
    else:
        txt = "#Chr\tStart\tEnd\t%s\tLength\tCoveredSite\tTotalDepth\t%sCoverage(%%)\tMeanDepth\n" % (
            "RegionID" if mode == 3 else "GeneID", hgc)
        suffix = "bed.stat.gz" if mode in (3, 4) else "gene.stat.gz"
    for t in sorted(genes):
        ids = sorted(genes[t], key=lambda s: s.encode())
        ids.sort(key=lambda gid: genes[t][gid].start)          # stable: ties stay in id order
        for gid in ids:
            g = genes[t][gid]
            mid = "" if mode == 5 else gid + "\t"
            txt += "%s\t%d\t%d\t%s%d\t%d\t%d\t%s%.2f\t%.2f\n" % (
                names[t], g.start, g.end, mid, g.length, g.cover, g.depth, gccol(g.gc, g.length),
                g.cover * 100.0 / g.length, g.depth * 1.0 / g.length)
            SL += g.length; SC += g.cover; SD += g.depth; SG += g.gc
    out[suffix] = txt + footer(SL, SC, SD, SG)
    return out
