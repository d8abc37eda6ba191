"""tools/synth.py — seeded synthetic workloads: genomes, coordinate-sorted read
sets with the C2 CIGAR mix, their run (interval) streams, and a small BAM writer.

Used by bench.py (torch generator: the 3 Gb / 10^9-record stream is produced directly in HBM) and
by tests (numpy generator + BAM writer).  Nothing here is product code.

CIGAR mix (fractions of records): 85 % 150M · 5 % aM dD bM (d 1-20) · 5 % aM iI bM (i 1-10) ·
4 % sS (150-s)M (s 1-40) · 1 % aM nN bM (n 100-5000).  Read starts are the order statistics of a
piecewise-constant density over 100 kb blocks: 99 % of blocks at 1x, 1 % at a multiplier drawn
from {0, 0, 0.25, 2, 4} (zero-coverage holes and pile-ups), so starts come out already sorted.

Every record's FIRST run starts at the record position, hence the first-run stream is sorted by
(tid, beg) — the PD_PUSH_SORTED stream; second runs form the small unsorted stream.
"""
import struct
import zlib

import numpy as np

READ_LEN = 150
BLOCK = 100000
MAX_SPAN = READ_LEN + 5000 + 64

# Capsicum annuum chromosome lengths as printed in the reference README (lines 63-73) for
# Chr01-Chr09; Chr10-12 and 500 scaffolds are filled in to reach 3.0 Gb (README:128 "3 Gb").
CAPSICUM_CHR = [332615375, 177319215, 289790774, 248932513, 254874144, 253233553, 266382521,
                174326481, 278410012, 215000000, 200000000, 182000000]


def genome_c2(seed=42, scale=1.0):
    """12 chromosomes + 500 scaffolds of 10-500 kb, ~3.0e9 bp at scale 1."""
    rng = np.random.default_rng(seed)
    chrs = [max(20000, int(l * scale)) for l in CAPSICUM_CHR]
    n_scaf = 500 if scale >= 0.05 else max(4, int(500 * scale * 10))
    scaf = rng.integers(10000, 500001, n_scaf)
    scaf = np.maximum(10000, (scaf * min(1.0, scale * 4 if scale < 0.25 else 1.0)).astype(np.int64))
    lens = np.array(chrs + list(scaf), dtype=np.int64)
    names = ["Chr%02d" % (i + 1) for i in range(12)] + ["scaf%04d" % i for i in range(len(scaf))]
    return names, lens


def _block_cdf(rng, length):
    nb = max(1, -(-int(length) // BLOCK))
    w = np.ones(nb, dtype=np.float64)
    odd = rng.random(nb) < 0.01
    w[odd] = rng.choice([0.0, 0.0, 0.25, 2.0, 4.0], size=int(odd.sum()))
    if w.sum() == 0:
        w[:] = 1.0
    cdf = np.concatenate([[0.0], np.cumsum(w)])
    return cdf / cdf[-1]


def records_per_contig(lens, n_records):
    lens = np.asarray(lens, dtype=np.float64)
    n = np.floor(lens / lens.sum() * n_records).astype(np.int64)
    n[0] += n_records - n.sum()
    return n


# ------------------------------------------------------------------------------------------------
# numpy generator (CPU; tests and BAM fixtures)
# ------------------------------------------------------------------------------------------------
def gen_records_numpy(lens, n_records, seed=42):
    """Returns dict of arrays: tid, pos (0-based, sorted within tid), kind (0..4), a, x
    (kind 0: 150M; 1: aM xD (150-a)M; 2: aM xI (150-a-x)M; 3: xS (150-x)M; 4: aM xN (150-a)M)."""
    rng = np.random.default_rng(seed)
    per = records_per_contig(lens, n_records)
    out = {k: [] for k in ("tid", "pos", "kind", "a", "x")}
    for t, (ln, n) in enumerate(zip(lens, per)):
        if n == 0:
            continue
        cdf = _block_cdf(rng, ln)
        e = rng.exponential(size=n + 1)
        c = np.cumsum(e)
        u = c[:-1] / c[-1]
        b = np.searchsorted(cdf, u, side="right") - 1
        b = np.clip(b, 0, len(cdf) - 2)
        frac = (u - cdf[b]) / np.maximum(cdf[b + 1] - cdf[b], 1e-300)
        span_room = max(1, int(ln) - MAX_SPAN)
        pos = np.minimum(((b + frac) * BLOCK).astype(np.int64), span_room - 1)
        pos = np.maximum.accumulate(np.clip(pos, 0, None))
        r = rng.random(n)
        kind = np.select([r < 0.85, r < 0.90, r < 0.95, r < 0.99], [0, 1, 2, 3], 4).astype(np.int8)
        a = rng.integers(10, 131, n).astype(np.int32)
        x = np.where(kind == 1, rng.integers(1, 21, n),
            np.where(kind == 2, rng.integers(1, 11, n),
            np.where(kind == 3, rng.integers(1, 41, n),
            np.where(kind == 4, rng.integers(100, 5001, n), 0)))).astype(np.int32)
        out["tid"].append(np.full(n, t, dtype=np.int32)); out["pos"].append(pos.astype(np.int32))
        out["kind"].append(kind); out["a"].append(a); out["x"].append(x)
    return {k: np.concatenate(v) for k, v in out.items()}


def records_to_runs(rec):
    """-> (sorted_runs (n,3) int32 first run of each record, other_runs (m,3) int32)."""
    tid, pos, kind, a, x = rec["tid"], rec["pos"], rec["kind"], rec["a"], rec["x"]
    first_len = np.where(kind == 0, READ_LEN, np.where(kind == 3, READ_LEN - x, a)).astype(np.int32)
    first = np.stack([tid, pos, pos + first_len], axis=1).astype(np.int32)
    two = (kind == 1) | (kind == 2) | (kind == 4)
    gap = np.where((kind == 1) | (kind == 4), x, 0)
    second_len = np.where(kind == 2, READ_LEN - a - x, READ_LEN - a)
    b2 = pos + a + gap
    other = np.stack([tid[two], b2[two], (b2 + second_len)[two]], axis=1).astype(np.int32)
    return first, other


# ------------------------------------------------------------------------------------------------
# torch generator (bench: the whole stream is produced in HBM)
# ------------------------------------------------------------------------------------------------
NEAR_SPAN = 1024      # pd_bamwalk.h: later runs of a read that begin within this many bases of its start


def gen_runs_torch(lens, n_records, device, seed=42, split=False):
    """Same workload definition as gen_records_numpy + records_to_runs, generated on `device`.
    Returns (sorted_runs int32 (n,3), other_runs int32 (m,3)) as torch tensors; with split=True the other runs come
    as the two streams the product's decoder emits: (near, far) = second runs that begin within NEAR_SPAN bases of
    their read's start / after a longer gap (N operations)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    per = records_per_contig(lens, n_records)
    first = torch.empty((int(n_records), 3), dtype=torch.int32, device=device)
    others, fars = [], []
    o = 0
    for t, (ln, n) in enumerate(zip(lens, per)):
        n = int(n)
        if n == 0:
            continue
        cdf = torch.from_numpy(_block_cdf(rng, ln)).to(device)
        e = torch.empty(n + 1, dtype=torch.float64, device=device).exponential_(generator=g)
        c = torch.cumsum(e, 0)
        u = c[:-1] / c[-1]
        del e, c
        b = torch.searchsorted(cdf, u, right=True) - 1
        b.clamp_(0, cdf.numel() - 2)
        frac = (u - cdf[b]) / (cdf[b + 1] - cdf[b]).clamp_min(1e-300)
        span_room = max(1, int(ln) - MAX_SPAN)
        pos = ((b.to(torch.float64) + frac) * BLOCK).to(torch.int64).clamp_(0, span_room - 1)
        del u, b, frac
        pos = torch.cummax(pos, 0).values.to(torch.int32)
        r = torch.rand(n, device=device, generator=g)
        kind = (r >= 0.85).to(torch.int32) + (r >= 0.90).to(torch.int32) + (r >= 0.95).to(torch.int32) + \
               (r >= 0.99).to(torch.int32)
        a = torch.randint(10, 131, (n,), device=device, generator=g, dtype=torch.int32)
        xr = torch.rand(n, device=device, generator=g)
        x = torch.zeros(n, dtype=torch.int32, device=device)
        x = torch.where(kind == 1, (1 + xr * 20).to(torch.int32), x)
        x = torch.where(kind == 2, (1 + xr * 10).to(torch.int32), x)
        x = torch.where(kind == 3, (1 + xr * 40).to(torch.int32), x)
        x = torch.where(kind == 4, (100 + xr * 4901).to(torch.int32), x)
        first_len = torch.where(kind == 0, torch.full_like(a, READ_LEN),
                                torch.where(kind == 3, READ_LEN - x, a))
        sl = first[o:o + n]
        sl[:, 0] = t
        sl[:, 1] = pos
        sl[:, 2] = pos + first_len
        two = (kind == 1) | (kind == 2) | (kind == 4)
        gap = torch.where((kind == 1) | (kind == 4), x, torch.zeros_like(x))
        second_len = torch.where(kind == 2, READ_LEN - a - x, READ_LEN - a)
        b2 = (pos + a + gap)[two]
        oth = torch.stack([torch.full_like(b2, t), b2, b2 + second_len[two]], dim=1)
        if split:
            isfar = ((a + gap) > NEAR_SPAN)[two]
            others.append(oth[~isfar]); fars.append(oth[isfar])
        else:
            others.append(oth)
        o += n
        del r, kind, a, xr, x, first_len, two, gap, second_len, b2, pos
    other = torch.cat(others, 0).contiguous() if others else torch.empty((0, 3), dtype=torch.int32, device=device)
    if split:
        far = torch.cat(fars, 0).contiguous() if fars else torch.empty((0, 3), dtype=torch.int32, device=device)
        return first[:o].contiguous(), other, far
    return first[:o].contiguous(), other


# ------------------------------------------------------------------------------------------------
# BAM writer (BGZF + records with '*' SEQ/QUAL), for generated fixtures
# ------------------------------------------------------------------------------------------------
def _reg2bin(beg, end):
    end = end - 1
    out = np.zeros(beg.shape, dtype=np.int64)
    done = np.zeros(beg.shape, dtype=bool)
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        m = (~done) & ((beg >> shift) == (end >> shift))
        out[m] = base + (beg[m] >> shift)
        done |= m
    return out.astype(np.uint16)


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


_BODY = None      # set before the fork so that pool workers see the byte stream without pickling it


def _bam_header_bytes(names, lens, sorted_header):
    text = "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if sorted_header else "unsorted")
    text += "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, ln) for nm, ln in zip(names, lens))
    tb = text.encode()
    hdr = b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(names))
    for nm, ln in zip(names, lens):
        nb = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", int(ln))
    return hdr


_PAY = None


def _payload_chunk(args):
    """records [lo, hi) with SEQ/QUAL payload -> BGZF bytes (whole records per chunk)."""
    lo, hi, level, seed = args
    tid, pos, kind, a, x, flags, mapq = _PAY
    n = hi - lo
    sl = slice(lo, hi)
    k, aa, xx = kind[sl], a[sl], x[sl]
    M, I, D, N, S = 0, 1, 2, 3, 4
    ncig = np.where(k == 0, 1, np.where(k == 3, 2, 3)).astype(np.int64)
    c0 = np.where(k == 0, (READ_LEN << 4) | M, np.where(k == 3, (xx << 4) | S, (aa << 4) | M)).astype(np.uint32)
    op1 = np.select([k == 1, k == 2, k == 4], [D, I, N], 0)
    c1 = np.where(k == 3, ((READ_LEN - xx) << 4) | M, (xx << 4) | op1).astype(np.uint32)
    c2 = np.where(k == 2, ((READ_LEN - aa - xx) << 4) | M, ((READ_LEN - aa) << 4) | M).astype(np.uint32)
    span = np.where(k == 0, READ_LEN, np.where(k == 3, READ_LEN - xx, np.where(k == 2, READ_LEN - xx, READ_LEN + xx))).astype(np.int64)
    p64 = pos[sl].astype(np.int64)
    SEQB, QB = (READ_LEN + 1) // 2, READ_LEN
    fixed = 36 + 2                      # block_size + 32 + "r\0"
    width = fixed + 12 + SEQB + QB
    mat = np.zeros((n, width), dtype=np.uint8)
    head = np.zeros(n, dtype=np.dtype([("bs", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("lrn", "u1"), ("mq", "u1"),
                                       ("bin", "<u2"), ("nc", "<u2"), ("flag", "<u2"), ("lseq", "<i4"),
                                       ("mtid", "<i4"), ("mpos", "<i4"), ("tlen", "<i4"), ("name", "S2")]))
    head["bs"] = (34 + 4 * ncig + SEQB + QB).astype(np.int32)
    head["tid"], head["pos"], head["lrn"], head["mq"] = tid[sl], pos[sl], 2, mapq[sl]
    head["bin"] = _reg2bin(p64, p64 + span)
    head["nc"], head["flag"], head["lseq"], head["mtid"], head["mpos"] = ncig, flags[sl], READ_LEN, -1, -1
    head["name"] = b"r"
    mat[:, :fixed] = head.view(np.uint8).reshape(n, fixed)
    rng = np.random.default_rng(seed)
    seq = rng.integers(0, 4, (n, SEQB * 2), dtype=np.uint8)
    nib = np.array([1, 2, 4, 8], dtype=np.uint8)[seq]
    seqb = (nib[:, 0::2] << 4) | nib[:, 1::2]
    qual = np.array([37, 37, 37, 37, 37, 37, 25, 25, 11, 2], dtype=np.uint8)[rng.integers(0, 10, (n, QB))]
    # variable part: cigar (4*ncig) then seq then qual, left-aligned after the fixed head
    cig = np.stack([c0, c1, c2], axis=1).view(np.uint8).reshape(n, 12)
    body = np.concatenate([cig, seqb, qual], axis=1)                  # (n, 12+SEQB+QB), cigar padded to 3 ops
    # drop the unused cigar slots row-wise with a mask
    col = np.arange(12 + SEQB + QB)[None, :]
    keep = (col < (4 * ncig)[:, None]) | (col >= 12)
    mat[:, fixed:] = body
    full_keep = np.concatenate([np.ones((n, fixed), dtype=bool), keep], axis=1)
    data = mat[full_keep]
    blocks = []
    for o in range(0, data.size, 65280):
        blocks.append(_bgzf_block(data[o:o + 65280].tobytes(), level))
    return b"".join(blocks)


def _write_bam_payload(path, names, lens, rec, flags, mapq, sorted_header, level, procs):
    global _PAY
    tid, pos, kind, a, x = (np.asarray(rec[k]) for k in ("tid", "pos", "kind", "a", "x"))
    n = tid.size
    flags = np.zeros(n, dtype=np.uint16) if flags is None else np.asarray(flags, dtype=np.uint16)
    mapq = np.full(n, 60, dtype=np.uint8) if mapq is None else np.asarray(mapq, dtype=np.uint8)
    _PAY = (tid, pos, kind, a, x, flags, mapq)
    hdr = _bam_header_bytes(names, lens, sorted_header)
    step = 200000
    tasks = [(o, min(o + step, n), level, 1000 + o // step) for o in range(0, n, step)]
    with open(path, "wb") as f:
        for o in range(0, len(hdr), 65280):
            f.write(_bgzf_block(hdr[o:o + 65280], level))
        if procs > 1 and len(tasks) > 1:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(procs) as pool:
                for chunk in pool.imap(_payload_chunk, tasks):
                    f.write(chunk)
        else:
            for t in tasks:
                f.write(_payload_chunk(t))
        f.write(_BGZF_EOF)
    _PAY = None
    return n


def _compress_range(args):
    lo, hi, level = args
    buf = _BODY
    out = []
    for o in range(lo, hi, 65280):
        out.append(_bgzf_block(bytes(buf[o:min(o + 65280, hi)]), level))
    return b"".join(out)


def write_bam(path, names, lens, rec, flags=None, mapq=None, sorted_header=True, level=1, procs=8, payload=False):
    """Writes records (dict from gen_records_numpy) as a BAM.  Returns the number of records.
    flags/mapq: optional per-record arrays (default 0 / 60).  payload=False: '*' SEQ/QUAL (tiny
    files); payload=True: 150 random bases + 4-level binned qualities per record, so that the
    compressed size per record (~100 B) and the inflate cost resemble a real short-read BAM."""
    if payload:
        return _write_bam_payload(path, names, lens, rec, flags, mapq, sorted_header, level, procs)
    tid, pos, kind, a, x = (np.asarray(rec[k]) for k in ("tid", "pos", "kind", "a", "x"))
    n = tid.size
    flags = np.zeros(n, dtype=np.uint16) if flags is None else np.asarray(flags, dtype=np.uint16)
    mapq = np.full(n, 60, dtype=np.uint8) if mapq is None else np.asarray(mapq, dtype=np.uint8)
    M, I, D, N, S = 0, 1, 2, 3, 4
    ncig = np.where(kind == 0, 1, np.where(kind == 3, 2, 3)).astype(np.uint16)
    c0 = np.where(kind == 0, (READ_LEN << 4) | M, np.where(kind == 3, (x << 4) | S, (a << 4) | M)).astype(np.uint32)
    op1 = np.select([kind == 1, kind == 2, kind == 4], [D, I, N], 0)
    c1 = np.where(kind == 3, ((READ_LEN - x) << 4) | M, (x << 4) | op1).astype(np.uint32)
    c2 = np.where(kind == 2, ((READ_LEN - a - x) << 4) | M, ((READ_LEN - a) << 4) | M).astype(np.uint32)
    span = np.where(kind == 0, READ_LEN, np.where(kind == 3, READ_LEN - x,
            np.where(kind == 2, READ_LEN - x, READ_LEN + x))).astype(np.int64)
    dt = np.dtype([("bs", "<i4"), ("tid", "<i4"), ("pos", "<i4"), ("lrn", "u1"), ("mq", "u1"), ("bin", "<u2"),
                   ("nc", "<u2"), ("flag", "<u2"), ("lseq", "<i4"), ("mtid", "<i4"), ("mpos", "<i4"),
                   ("tlen", "<i4"), ("name", "S2"), ("c0", "<u4"), ("c1", "<u4"), ("c2", "<u4")])
    arr = np.zeros(n, dtype=dt)
    arr["bs"] = 34 + 4 * ncig.astype(np.int32)
    arr["tid"], arr["pos"], arr["lrn"], arr["mq"] = tid, pos, 2, mapq
    arr["bin"] = _reg2bin(pos.astype(np.int64), pos.astype(np.int64) + span)
    arr["nc"], arr["flag"], arr["lseq"], arr["mtid"], arr["mpos"] = ncig, flags, 0, -1, -1
    arr["name"] = b"r"
    arr["c0"], arr["c1"], arr["c2"] = c0, c1, c2
    mat = arr.view(np.uint8).reshape(n, dt.itemsize)
    size = (38 + 4 * ncig.astype(np.int64))
    mask = np.arange(dt.itemsize)[None, :] < size[:, None]
    body = mat[mask]                                   # row-major: records back to back
    text = "@HD\tVN:1.6\tSO:%s\n" % ("coordinate" if sorted_header else "unsorted")
    text += "".join("@SQ\tSN:%s\tLN:%d\n" % (nm, ln) for nm, ln in zip(names, lens))
    tb = text.encode()
    hdr = b"BAM\x01" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(names))
    for nm, ln in zip(names, lens):
        nb = nm.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", int(ln))
    with open(path, "wb") as f:
        for o in range(0, len(hdr), 65280):
            f.write(_bgzf_block(hdr[o:o + 65280], level))
        total = body.size
        if total:
            step = max(65280, (total // max(1, procs * 4) // 65280 + 1) * 65280)
            global _BODY
            _BODY = body
            tasks = [(o, min(o + step, total), level) for o in range(0, total, step)]
            if procs > 1 and len(tasks) > 1:
                import multiprocessing as mp
                with mp.get_context("fork").Pool(procs) as pool:
                    for chunk in pool.imap(_compress_range, tasks):
                        f.write(chunk)
            else:
                for t in tasks:
                    f.write(_compress_range(t))
            _BODY = None
        f.write(_BGZF_EOF)
    return n
