"""Host pipeline on GENERATED inputs (tools/synth.py, a few hundred thousand records): BAM written
by the generator, index built by this repo's bai_build, targets from a synthetic annotation.  The
expected files come from the compiled reference (oracle/_ref/pandepth_ref, reading the SAME BAM and
the SAME .bai through htslib) when it is present, else from the Python oracle.  Exercises what the
small golden fixtures cannot: multi-block BGZF, records spanning blocks, index-driven target fetch
across many chunks, parallel range readers, the threaded sequential reader."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("gen")
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a", "pandepth_index"], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness")], check=True, stdout=subprocess.DEVNULL)
    names, lens = synth.genome_c2(scale=0.002)
    rec = synth.gen_records_numpy(lens, 300000, seed=5)
    rng = np.random.default_rng(9)
    n = rec["tid"].size
    flags = np.where(rng.random(n) < 0.03, 1024, 0).astype(np.uint16) | np.where(rng.random(n) < 0.5, 16, 0).astype(np.uint16)
    mapq = rng.choice([0, 20, 60, 60], n).astype(np.uint8)
    bam = str(d / "g.bam")
    synth.write_bam(bam, names, lens, rec, flags=flags, mapq=mapq, procs=2, payload=True)
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], check=True)
    # a file whose header says SO:coordinate although two blocks of records have changed places (two sorted files
    # concatenated look like this): no index can exist for it; the reference reads it with its no-index cursor, whose
    # result depends on the order the records come in — the device decoder must notice and leave the file to the host reader
    idx = np.arange(n)
    idx[40000:43000], idx[200000:203000] = idx[200000:203000].copy(), idx[40000:43000].copy()
    synth.write_bam(str(d / "l.bam"), names, lens, {k: v[idx] for k, v in rec.items()}, flags=flags[idx], mapq=mapq[idx], procs=1, payload=True)
    # annotation: 300 transcripts of 1-5 CDS, plus a BED of 400 regions
    g, b = ["##gff-version 3"], []
    for t in range(300):
        ci = int(rng.integers(0, 12))
        s = int(rng.integers(1, lens[ci] - 30000))
        for e in range(int(rng.integers(1, 6))):
            el = int(rng.integers(60, 900))
            g.append("%s\tsyn\tCDS\t%d\t%d\t.\t+\t0\tID=c%d.%d;Parent=t%d" % (names[ci], s, s + el - 1, t, e, t))
            s += el + int(rng.integers(50, 4000))
    for k in range(400):
        ci = int(rng.integers(0, len(names)))
        s = int(rng.integers(1, max(2, lens[ci] - 5000)))
        b.append("%s\t%d\t%d" % (names[ci], s, min(int(lens[ci]), s + int(rng.integers(1, 5000)))))
    (d / "g.gff").write_text("\n".join(g) + "\n")
    (d / "g.bed").write_text("\n".join(b) + "\n")
    # targets that are whole chromosomes (index chunks larger than a device batch are cut at record starts the index
    # names) next to neighbouring small ones (chunks that share BGZF members become one unit)
    wb = ["%s\t0\t%d\tall%d" % (names[ci], lens[ci], ci) for ci in (0, 1, 2, 5)]
    wb += ["%s\t%d\t%d\tnear%d" % (names[7], 20000 + 700 * k, 20000 + 700 * k + 300, k) for k in range(40)]
    (d / "w.bed").write_text("\n".join(wb) + "\n")
    return d


CASES = [
    ("chr", ["-i", "g.bam"], "chr.stat.gz"),
    ("chr_q30_x0", ["-i", "g.bam", "-q", "30", "-x", "0"], "chr.stat.gz"),
    ("w100", ["-i", "g.bam", "-w", "100"], "win.stat.gz"),
    ("w5000", ["-i", "g.bam", "-w", "5000"], "win.stat.gz"),
    ("gff", ["-i", "g.bam", "-g", "g.gff"], "gene.stat.gz"),
    ("gff_a", ["-i", "g.bam", "-g", "g.gff", "-a"], "gene.stat.gz"),
    ("bed", ["-i", "g.bam", "-b", "g.bed", "-d", "3"], "bed.stat.gz"),
    ("bed_whole", ["-i", "g.bam", "-b", "w.bed"], "bed.stat.gz"),
    ("bed_whole_a", ["-i", "g.bam", "-b", "w.bed", "-a", "-q", "20"], "bed.stat.gz"),
    ("noindex", ["-i", "g.bam", "-s"], "chr.stat.gz"),
    ("noindex_gff", ["-i", "g.bam", "-s", "-g", "g.gff"], "gene.stat.gz"),
    ("header_lies", ["-i", "l.bam"], "chr.stat.gz"),
    ("header_lies_w", ["-i", "l.bam", "-w", "5000", "-q", "20"], "win.stat.gz"),
]


BINARIES = [
    pytest.param(os.path.join(HERE, "harness", "pandepth_oracle_cli"), id="host+oracle-engine"),
    pytest.param(os.path.join(HERE, "harness", "pandepth_oracle_cli") + ":dd", id="host+oracle-engine-small-batches"),
    pytest.param(os.path.join(HERE, "harness", "pandepth_oracle_cli") + ":host", id="host+oracle-engine-host-decode"),
    pytest.param(os.path.join(ROOT, "pandepth_amd", "pandepth"), id="pandepth-mi355x", marks=pytest.mark.gpu),
    pytest.param(os.path.join(ROOT, "pandepth_amd", "pandepth") + ":dd", id="pandepth-mi355x-small-batches", marks=pytest.mark.gpu),
    pytest.param(os.path.join(ROOT, "pandepth_amd", "pandepth") + ":host", id="pandepth-mi355x-host-decode", marks=pytest.mark.gpu),
]


@pytest.mark.parametrize("cli", BINARIES)
@pytest.mark.parametrize("name,args,suffix", CASES, ids=[c[0] for c in CASES])
def test_generated_inputs_match_reference(data, cli, name, args, suffix):
    env = dict(os.environ)
    # the pandepth binary decodes BAM on the GPU by default (inflate, record boundaries, filter, CIGAR walk)
    if cli.endswith(":dd"):            # ... in small batches: several per file, several feeder threads
        cli = cli[:-3]
        env.update(PANDEPTH_TUNE="dd_batch_mb=2")
    elif cli.endswith(":host"):        # ... or not at all: the host readers (libdeflate) feed pd_push_intervals
        cli = cli[:-5]
        env.update(PANDEPTH_TUNE="device_decode=0")
    for t in ("1", "5"):
        p = subprocess.run([cli] + args + ["-o", "mine_" + name, "-t", t], cwd=data, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-400:]
        mine = {s: (data / ("mine_%s.%s" % (name, s))).read_bytes() for s in [suffix] + (["SiteDepth.gz"] if "-a" in args else [])}
        if os.access(REF, os.X_OK):
            if t == "1":
                subprocess.run([REF] + args + ["-o", "ref_" + name], cwd=data, check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=900)
            for s, got in mine.items():
                assert got == (data / ("ref_%s.%s" % (name, s))).read_bytes(), "%s differs (-t %s)" % (s, t)
        else:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pd_oracle as O
            exp = O.run(args, cwd=str(data))
            for s, got in mine.items():
                assert gzip.decompress(got).decode() == exp[s], "%s differs (-t %s)" % (s, t)
            break


def _members_and_record_starts(bam_bytes):
    """BGZF member offsets of a BAM, and for every record (tid, pos, index of the member it starts in)"""
    import struct
    import zlib
    offs, chunks, p = [], [], 0
    while p < len(bam_bytes):
        xlen = struct.unpack_from("<H", bam_bytes, p + 10)[0]
        bsize = struct.unpack_from("<H", bam_bytes, p + 16)[0] + 1          # generator and fixtures: BC is the only extra field
        offs.append(p)
        chunks.append(zlib.decompress(bam_bytes[p + 12 + xlen:p + bsize - 8], -15))
        p += bsize
    starts = np.cumsum([0] + [len(c) for c in chunks])
    data = b"".join(chunks)
    l_text = struct.unpack_from("<i", data, 4)[0]
    q = 8 + l_text
    n_ref = struct.unpack_from("<i", data, q)[0]; q += 4
    for _ in range(n_ref):
        q += 4 + struct.unpack_from("<i", data, q)[0] + 4
    recs = []
    while q + 4 <= len(data):
        bs = struct.unpack_from("<i", data, q)[0]
        tid, pos = struct.unpack_from("<ii", data, q + 4)
        recs.append((tid, pos, int(np.searchsorted(starts, q, side="right") - 1)))
        q += 4 + bs
    return offs, recs


def _bai_chunks(bai_bytes, tid, beg0, end):
    """chunks of a .bai that can hold reads overlapping [beg0, end) of reference tid (SAM spec 5.3), pruned with the linear index"""
    import struct
    p = 8
    for r in range(tid + 1):
        bins = {}
        n_bin = struct.unpack_from("<i", bai_bytes, p)[0]; p += 4
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai_bytes, p); p += 8
            bins[b] = [struct.unpack_from("<QQ", bai_bytes, p + 16 * k) for k in range(n_chunk)]
            p += 16 * n_chunk
        n_intv = struct.unpack_from("<i", bai_bytes, p)[0]; p += 4
        lin = struct.unpack_from("<%dQ" % n_intv, bai_bytes, p); p += 8 * n_intv
    want = [0]
    e = end - 1
    for sh, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        want += range(off + (beg0 >> sh), off + (e >> sh) + 1)
    min_off = lin[beg0 >> 14] if (beg0 >> 14) < len(lin) else 0
    return [c for b in want for c in bins.get(b, []) if c[1] > min_off]


def test_damage_between_target_chunks_is_nobodys_business(data):
    """Target chunks a member or two apart: the device path reads them as ONE unit, gaps included (fewer bytes than a tail
    behind every chunk); the host reader and the reference's iterator visit the chunks only.  A damaged member in a gap must not
    change the outcome — the unit goes back to the host reader chunk by chunk — and damage inside a chunk is an error on both."""
    import random
    names, lens = synth.genome_c2(scale=0.0004)
    rec = synth.gen_records_numpy(lens, 40000, seed=3)
    synth.write_bam(str(data / "thin.bam"), names, lens, rec, procs=1, payload=True)
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), str(data / "thin.bam")], check=True)
    bam = (data / "thin.bam").read_bytes()
    offs, _ = _members_and_record_starts(bam)
    rng = random.Random(1)
    targets = []
    for t in range(40):
        ci = rng.randrange(12)
        s0 = rng.randrange(1, int(lens[ci]) - 3000)
        targets.append((ci, s0, s0 + rng.randrange(50, 900)))
    (data / "gap.bed").write_text("".join("%s\t%d\t%d\tt%d\n" % (names[ci], b - 1, e, k) for k, (ci, b, e) in enumerate(targets)))
    bai = (data / "thin.bam.bai").read_bytes()
    chunks = sorted(c for ci, b, e in targets for c in _bai_chunks(bai, ci, max(0, b - 2), e + 1))
    used = np.zeros(len(offs), bool)                                 # members some chunk reads
    for cb, ce in chunks:
        lo = int(np.searchsorted(offs, cb >> 16, side="right") - 1)
        hi = int(np.searchsorted(offs, ce >> 16, side="right") - 1)
        used[lo:hi + (1 if ce & 0xffff else 0)] = True
    gaps = [m for m in range(1, len(offs) - 1) if not used[m] and used[:m].any() and used[m + 1:].any() and
            offs[int(np.flatnonzero(used[m:])[0]) + m] - offs[m] <= 40000]
    assert gaps, "this input should have unread members between chunks"
    inside = [int(m) for m in np.flatnonzero(used)[::max(1, int(used.sum()) // 3)][:3]]
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    work = data / "gapcase"
    work.mkdir()
    os.symlink(data / "gap.bed", work / "gap.bed")
    (work / "m.bam.bai").write_bytes(bai)
    outcomes = {}
    for m in gaps[:4] + inside:
        b = bytearray(bam)
        b[(offs[m] + 18 + offs[m + 1] - 8) // 2] ^= 0x10             # inside the member's deflate stream
        (work / "m.bam").write_bytes(bytes(b))
        got = []
        for env in ({"PANDEPTH_TUNE": "dd_batch_mb=2", "PANDEPTH_TIMING": "1"}, {"PANDEPTH_TUNE": "device_decode=0"}):
            p = subprocess.run([cli, "-i", "m.bam", "-b", "gap.bed", "-o", "o", "-t", "3"], cwd=work, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
            f = work / "o.bed.stat.gz"
            got.append((p.returncode, f.read_bytes() if p.returncode == 0 and f.exists() else None))
            if f.exists():
                f.unlink()
            if "PANDEPTH_TIMING" in env:
                line = [x for x in p.stderr.decode().splitlines() if "region fetch:" in x][0].split()
                assert int(line[-2]) < int(line[-6]), "chunks should have been joined: " + " ".join(line)
        assert got[0] == got[1], "member %d: device path rc %d, host path rc %d" % (m, got[0][0], got[1][0])
        outcomes[m] = got[1][0]
    assert any(outcomes[m] == 0 for m in gaps[:4]) and any(outcomes[m] != 0 for m in inside), outcomes
    if os.access(REF, os.X_OK):                                      # the undamaged answer, for the cases that went through
        subprocess.run([REF, "-i", "thin.bam", "-b", "gap.bed", "-o", "gap_ref"], cwd=data, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        (work / "m.bam").write_bytes(bam)
        subprocess.run([cli, "-i", "m.bam", "-b", "gap.bed", "-o", "o"], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert (work / "o.bed.stat.gz").read_bytes() == (data / "gap_ref.bed.stat.gz").read_bytes()


def test_declined_pass_has_counted_nothing(data):
    """A region-fetch pass that is DECLINED after some of its units were handed back: a damaged deflate stream in an early
    gap member sends its unit back to the host reader (status 2), a member header that is not one in a late gap stops that
    batch's member scan short, the device pass is abandoned and the host reader goes through the file chunk by chunk.  The
    handed-back unit must not have been counted by the abandoned pass (it used to be: its depth came out doubled)."""
    import random
    names, lens = synth.genome_c2(scale=0.0004)
    rec = synth.gen_records_numpy(lens, 40000, seed=3)
    bam_path = data / "thin2.bam"
    synth.write_bam(str(bam_path), names, lens, rec, procs=1, payload=True)
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), str(bam_path)], check=True)
    bam = bam_path.read_bytes()
    offs, _ = _members_and_record_starts(bam)
    rng = random.Random(1)
    targets = []
    for t in range(40):
        ci = rng.randrange(12)
        s0 = rng.randrange(1, int(lens[ci]) - 3000)
        targets.append((ci, s0, s0 + rng.randrange(50, 900)))
    (data / "gap2.bed").write_text("".join("%s\t%d\t%d\tt%d\n" % (names[ci], b - 1, e, k) for k, (ci, b, e) in enumerate(targets)))
    bai = (data / "thin2.bam.bai").read_bytes()
    chunks = sorted(c for ci, b, e in targets for c in _bai_chunks(bai, ci, max(0, b - 2), e + 1))
    used = np.zeros(len(offs), bool)
    for cb, ce in chunks:
        lo = int(np.searchsorted(offs, cb >> 16, side="right") - 1)
        hi = int(np.searchsorted(offs, ce >> 16, side="right") - 1)
        used[lo:hi + (1 if ce & 0xffff else 0)] = True
    gaps = [m for m in range(1, len(offs) - 1) if not used[m] and used[:m].any() and used[m + 1:].any() and
            offs[int(np.flatnonzero(used[m:])[0]) + m] - offs[m] <= 40000]
    assert len(gaps) >= 2
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    work = data / "declcase"
    work.mkdir()
    os.symlink(data / "gap2.bed", work / "gap.bed")
    (work / "m.bam.bai").write_bytes(bai)
    (work / "m.bam").write_bytes(bam)
    subprocess.run([cli, "-i", "m.bam", "-b", "gap.bed", "-o", "clean"], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, PANDEPTH_TUNE="device_decode=0"))
    want = (work / "clean.bed.stat.gz").read_bytes()
    hit = 0
    for ga, gb in [(gaps[0], gaps[-1]), (gaps[1], gaps[-2]), (gaps[0], gaps[len(gaps) // 2])]:
        if ga >= gb:
            continue
        b = bytearray(bam)
        b[(offs[ga] + 18 + offs[ga + 1] - 8) // 2] ^= 0x10             # early gap member: its deflate stream (the unit goes back)
        b[offs[gb]] ^= 0xff                                            # late gap member: its magic (the member scan stops short)
        (work / "m.bam").write_bytes(bytes(b))
        for env in ({"PANDEPTH_TUNE": "dd_batch_mb=1", "PANDEPTH_TIMING": "1"}, {"PANDEPTH_TUNE": "device_decode=0"}):
            p = subprocess.run([cli, "-i", "m.bam", "-b", "gap.bed", "-o", "o", "-t", "1"], cwd=work, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env))
            assert p.returncode == 0, p.stderr.decode()[-400:]
            assert (work / "o.bed.stat.gz").read_bytes() == want, "gap members %d / %d, %s" % (ga, gb, sorted(env))
            (work / "o.bed.stat.gz").unlink()
            if "PANDEPTH_TIMING" in env:
                line = [x for x in p.stderr.decode().splitlines() if "device decode:" in x][0]
                if "DECLINED" in line and " 0 units handed back" not in line:
                    hit += 1
    assert hit, "no case had a unit handed back before the pass was declined: the test does not exercise what it is for"


# ---- round 5: the record chain confirmed on the device, batches queued and collected ----------------------------------------------
def _expected_chr(data):
    """the whole-chromosome table of g.bam: the reference's when it is here, else the oracle's"""
    if os.access(REF, os.X_OK):
        subprocess.run([REF, "-i", "g.bam", "-o", "ref_chain"], cwd=data, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        return gzip.decompress((data / "ref_chain.chr.stat.gz").read_bytes()).decode()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pd_oracle as O
    return O.run(["-i", "g.bam"], cwd=str(data))["chr.stat.gz"]


QUEUE_CASES = [
    # (id, PANDEPTH_TUNE, extra environment of the stand-in engine, what its [chain] line must say)
    ("depth1", "dd_batch_mb=1,dd_depth=1,dd_threads=4", {}, "dev"),
    ("depth2", "dd_batch_mb=1,dd_depth=2,dd_threads=6", {}, "dev"),
    ("depth3", "dd_batch_mb=1,dd_depth=3,dd_threads=4", {}, "dev"),
    ("depth_clamped", "dd_batch_mb=1,dd_depth=5,dd_threads=6", {}, "dev"),          # twelve slots: six readers hold two each
    ("no_queue_entry", "dd_batch_mb=1", {"PANDEPTH_TEST_NO_QUEUE": "1"}, "dev"),      # an engine without decode_queue: submit, as before
    ("spoil_every", "dd_batch_mb=1", {"PANDEPTH_TEST_SPOIL_GUESS": "1"}, "redo"),
    ("spoil_third", "dd_batch_mb=2", {"PANDEPTH_TEST_SPOIL_GUESS": "3"}, "redo"),
    ("spoil_left_to_host", "dd_batch_mb=1", {"PANDEPTH_TEST_SPOIL_GUESS": "2", "PANDEPTH_TEST_MAX_REDO": "1"}, "host"),
    ("host_chain", "dd_batch_mb=1", {"PANDEPTH_TEST_HOST_CHAIN": "1", "PANDEPTH_TEST_SPOIL_GUESS": "2"}, "none"),
]


@pytest.mark.parametrize("name,tune,extra,expect", QUEUE_CASES, ids=[c[0] for c in QUEUE_CASES])
def test_queued_batches_and_device_chain(data, name, tune, extra, expect):
    """pd_decode_queue / pd_decode_collect through the product's feeder (host/pipeline.cpp) on the stand-in engine, whose batches run
    the product's cores — pdb2::chain_device included — with their lanes in a loop: whatever the readers hold in flight, and however many
    guessed record starts are spoilt behind the first pass, the table is the reference's; chain_device repairs what it is allowed to
    and leaves the rest to check_chain."""
    import re
    env = dict(os.environ, PANDEPTH_TUNE=tune, PANDEPTH_TEST_CHAIN_STATS="1", **extra)
    p = subprocess.run([os.path.join(HERE, "harness", "pandepth_oracle_cli"), "-i", "g.bam", "-o", "q_" + name, "-t", "6"], cwd=data,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert gzip.decompress((data / ("q_%s.chr.stat.gz" % name)).read_bytes()).decode() == _expected_chr(data)
    m = re.search(r"confirmed by chain_device: (\d+), left to check_chain: (\d+), segments chain_device walked again: (\d+)", p.stderr.decode())
    assert m, p.stderr.decode()[-400:]
    dev, host, redo = (int(x) for x in m.groups())
    if expect == "dev":
        assert dev > 3 and host == 0 and redo == 0
    elif expect == "redo":
        assert dev > 3 and host == 0 and redo > 10
    elif expect == "host":
        assert host > 3
    else:
        assert dev == 0 and host == 0


def _expected(data, args, suffix, tag):
    """a table of g.bam in the given mode: the reference's when it is here, else the oracle's"""
    if os.access(REF, os.X_OK):
        subprocess.run([REF] + args + ["-o", "ref_" + tag], cwd=data, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        return gzip.decompress((data / ("ref_%s.%s" % (tag, suffix))).read_bytes()).decode()
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pd_oracle as O
    return O.run(args, cwd=str(data))[suffix]


# the modes whose sessions keep 12-byte runs (everything that needs the arrays): the device confirms their chains too
GPU_CHAIN_MODES = [
    ("w100", ["-i", "g.bam", "-w", "100"], "win.stat.gz"),
    ("gff", ["-i", "g.bam", "-g", "g.gff"], "gene.stat.gz"),
    ("bed_whole_a", ["-i", "g.bam", "-b", "w.bed", "-a", "-q", "20"], "bed.stat.gz"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("tune", ["dd_batch_mb=1", "dd_batch_mb=1,decode_spoil=2", "dd_batch_mb=1,decode_spoil=1,decode_max_redo=2", "dd_batch_mb=2,decode_fast=0"],
                         ids=["fast", "spoil", "spoil_left_to_host", "host_chain"])
@pytest.mark.parametrize("mode,args,suffix", GPU_CHAIN_MODES, ids=[m[0] for m in GPU_CHAIN_MODES])
def test_device_chain_in_12_byte_sessions_gpu(data, mode, args, suffix, tune):
    """`-w 100`, `-g` (region fetch: units are index chunks) and `-b … -a` on the MI355X: batches queued and collected, the chain confirmed —
    and with decode_spoil repaired — on the device, the order of the 12-byte first runs reported by the emission itself; the reference's table"""
    import re
    env = dict(os.environ, PANDEPTH_TUNE=tune, PANDEPTH_TIMING="1")
    tag = "c12_%s_%s" % (mode, re.sub(r"\W", "_", tune))
    p = subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth")] + args + ["-o", tag, "-t", "6"], cwd=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert gzip.decompress((data / ("%s.%s" % (tag, suffix))).read_bytes()).decode() == _expected(data, args, suffix, "c12_" + mode)
    m = re.search(r"chain confirmed on the device for (\d+) batches, by the host for (\d+); segments the device walked again: (\d+)", p.stderr.decode())
    assert m, p.stderr.decode()[-600:]
    dev, host, redo = (int(x) for x in m.groups())
    if "decode_fast=0" in tune:
        assert dev == 0 and host > 0
    elif "decode_max_redo" in tune:
        assert host > 0
    else:
        assert dev > 0 and host == 0 and (redo > 0) == ("decode_spoil" in tune)


GPU_CHAIN_CASES = [
    ("fast_depth2", "dd_batch_mb=1"),
    ("fast_depth3", "dd_batch_mb=1,dd_depth=3,dd_threads=4"),
    ("fast_depth1", "dd_batch_mb=2,dd_depth=1"),
    ("host_chain", "dd_batch_mb=1,decode_fast=0"),
    ("spoil_every", "dd_batch_mb=1,decode_spoil=1"),
    ("spoil_third_depth1", "dd_batch_mb=2,decode_spoil=3,dd_depth=1"),
    ("spoil_left_to_host", "dd_batch_mb=1,decode_spoil=2,decode_max_redo=1"),
    ("spoil_host_chain", "dd_batch_mb=1,decode_spoil=2,decode_fast=0"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,tune", GPU_CHAIN_CASES, ids=[c[0] for c in GPU_CHAIN_CASES])
def test_queued_batches_and_device_chain_gpu(data, name, tune):
    """the same on the MI355X: k_chain_segments confirms (and, with decode_spoil, repairs) every batch's chain between the two record
    passes, the readers queue and collect; byte-identical with the reference whatever is in flight and whoever repairs."""
    import re
    env = dict(os.environ, PANDEPTH_TUNE=tune, PANDEPTH_TIMING="1")
    p = subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth"), "-i", "g.bam", "-o", "gq_" + name, "-t", "6"], cwd=data,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert gzip.decompress((data / ("gq_%s.chr.stat.gz" % name)).read_bytes()).decode() == _expected_chr(data)
    m = re.search(r"chain confirmed on the device for (\d+) batches, by the host for (\d+); segments the device walked again: (\d+)", p.stderr.decode())
    assert m, p.stderr.decode()[-600:]
    dev, host, redo = (int(x) for x in m.groups())
    if "decode_fast=0" in tune:
        assert dev == 0
    elif "decode_max_redo=1" in tune:
        assert host > 3
    else:
        assert dev > 3 and host == 0 and (redo > 10) == ("decode_spoil" in tune)


# Round 6: how a batch's bytes reach the device and how it is collected are choices with alternatives kept behind -X keys; the table must not know which ran.
GPU_TRANSPORT_CASES = [
    ("default", "dd_batch_mb=1"),
    ("copies_on_the_batch_streams", "dd_batch_mb=1,h2d_fifo=0"),
    ("collect_on_the_stream", "dd_batch_mb=1,sync_event=0"),
    ("both_as_in_round_5", "dd_batch_mb=1,h2d_fifo=0,sync_event=0,dd_depth=2,dd_threads=4"),
    ("two_copy_lanes", "dd_batch_mb=1,h2d_lanes=2"),
    ("copy_kernel", "dd_batch_mb=1,h2d_kernel=2"),
    ("helper_thread_for_the_slots", "dd_batch_mb=1,decode_warm=1"),
    ("one_buffer_pinned_ahead", "dd_batch_mb=1,dd_pin_ahead=1,dd_threads=5"),
    ("more_readers_than_batches_on_the_device", "dd_batch_mb=1,dd_threads=8,dd_inflight=3"),
    ("tables_do_not_fit_behind_the_members", "dd_batch_mb=1,dd_depth=2,dd_threads=3,dd_trace=1"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,tune", GPU_TRANSPORT_CASES, ids=[c[0] for c in GPU_TRANSPORT_CASES])
def test_decode_transport_alternatives_gpu(data, name, tune):
    """first come, first served copies on the main stream (one per batch) / per-batch streams / two lanes / a copy kernel; collect on the batch's
    event / on its stream; slots made ready by a helper thread; readers capped: the same bytes as the reference in every case, all batches on the device."""
    import re
    env = dict(os.environ, PANDEPTH_TUNE=tune, PANDEPTH_TIMING="1")
    p = subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth"), "-i", "g.bam", "-o", "gt_" + name, "-t", "6"], cwd=data,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert gzip.decompress((data / ("gt_%s.chr.stat.gz" % name)).read_bytes()).decode() == _expected_chr(data)
    m = re.search(r"device decode: (\d+) batches.*?(\d+) records on the device, (\d+) units handed back", p.stderr.decode())
    assert m and int(m.group(1)) > 3 and int(m.group(3)) == 0, p.stderr.decode()[-600:]
    if "dd_trace" in tune:
        assert len(re.findall(r"\[trace\] batch \d+ thread \d+", p.stderr.decode())) == int(m.group(1))
