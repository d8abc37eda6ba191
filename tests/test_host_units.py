"""Unit-level checks of the host modules through small generated files (no GPU): BGZF reader
(plain / threaded read-ahead / seek), BAM vs SAM decoding of the same records, the BAI builder and
the range splitter, and region-model quirks that the golden cases do not isolate."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from tools import synth  # noqa: E402
import pd_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def cli():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a", "pandepth_index"], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness")], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "harness", "pandepth_oracle_cli")


def run(cli, args, cwd):
    p = subprocess.run([cli] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    return p


def stat(path):
    return gzip.open(path).read().decode()


def test_bai_builder_index_is_usable_by_split_and_matches_sequential(cli, tmp_path):
    names, lens = synth.genome_c2(scale=0.0015)
    rec = synth.gen_records_numpy(lens, 120000, seed=3)
    bam = str(tmp_path / "x.bam")
    synth.write_bam(bam, names, lens, rec, procs=1, payload=True)
    run(cli, ["-i", "x.bam", "-o", "seq", "-t", "1"], tmp_path)           # no index: sequential stream
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], check=True)
    assert os.path.getsize(bam + ".bai") > 100
    for t in ("1", "3", "16"):
        run(cli, ["-i", "x.bam", "-o", "idx" + t, "-t", t], tmp_path)     # indexed: range-partitioned readers
        assert stat(tmp_path / ("idx%s.chr.stat.gz" % t)) == stat(tmp_path / "seq.chr.stat.gz")
    first, other = synth.records_to_runs(rec)
    d, off = O.depth_from_intervals(list(lens), np.concatenate([first, other]))
    rows = stat(tmp_path / "seq.chr.stat.gz").strip().split("\n")[1:-1]
    for t, row in enumerate(rows):
        f = row.split("\t")
        x = d[off[t]:off[t] + lens[t]]
        assert f[0] == names[t] and int(f[2]) == int((x > 0).sum()) and int(f[3]) == int(x.sum(dtype=np.uint64))


def test_index_build_rejects_unsorted(tmp_path):
    names, lens = synth.genome_c2(scale=0.0015)
    rec = synth.gen_records_numpy(lens, 2000, seed=4)
    rec = {k: v[::-1].copy() for k, v in rec.items()}
    bam = str(tmp_path / "u.bam")
    synth.write_bam(bam, names, lens, rec, procs=1, sorted_header=False)
    p = subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"not coordinate sorted" in p.stderr


def test_sam_gz_and_bam_give_the_same_tables(cli, tmp_path, golden_dir):
    sam = open(os.path.join(golden_dir, "f1", "f1.sam"), "rb").read()
    (tmp_path / "a.sam").write_bytes(sam)
    with gzip.open(tmp_path / "b.sam.gz", "wb") as f:
        f.write(sam)
    run(cli, ["-i", "a.sam", "-o", "a"], tmp_path)
    run(cli, ["-i", "b.sam.gz", "-o", "b"], tmp_path)
    assert (tmp_path / "a.chr.stat.gz").read_bytes() == (tmp_path / "b.chr.stat.gz").read_bytes()
    exp = open(os.path.join(golden_dir, "f1", "expected", "chr_sam.chr.stat.gz"), "rb").read()
    assert (tmp_path / "a.chr.stat.gz").read_bytes() == exp


def test_region_quirks(cli, tmp_path):
    # contigs: len 1 (no bin at all), len 201 with -w 200 (last 1-base bin dropped), len 400
    sam = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:one\tLN:1\n@SQ\tSN:q\tLN:201\n@SQ\tSN:r\tLN:400\n"
    sam += "a\t0\tq\t150\t60\t60M\t*\t0\t0\t*\t*\n"        # overhangs the contig end by 8 bases
    sam += "b\t0\tr\t1\t60\t400M\t*\t0\t0\t*\t*\n"
    (tmp_path / "q.sam").write_text(sam)
    run(cli, ["-i", "q.sam", "-o", "w", "-w", "200"], tmp_path)
    rows = stat(tmp_path / "w.win.stat.gz").split("\n")
    assert rows[1].split("\t")[:3] == ["q", "1", "200"] and rows[2].split("\t")[:3] == ["r", "1", "200"]
    assert rows[-2].startswith("##RegionLength: 600\t")             # 200 (q, base 201 dropped) + 400; "one" absent
    run(cli, ["-i", "q.sam", "-o", "m6", "-w", "100"], tmp_path)       # mode 6: `j < len` also drops q's base 201
    rows = stat(tmp_path / "m6.win.stat.gz").split("\n")
    assert [r.split("\t")[1] for r in rows[1:3]] == ["1", "101"] and rows[3].startswith("r\t1\t100")
    # BED is 1-based inclusive, ids built from the raw strings, duplicates double count
    (tmp_path / "r.bed").write_text("r\t010\t20\nr\t010\t20\nr\t30\t29\nzz\t1\t2\n")
    p = run(cli, ["-i", "q.sam", "-o", "b", "-b", "r.bed"], tmp_path)
    rows = stat(tmp_path / "b.bed.stat.gz").split("\n")
    assert rows[1] == "r\t10\t20\tr_010_20\t22\t22\t22\t100.00\t1.00"
    assert p.stderr.decode().count("Warning: This region may be incorrect.") == 2


def test_parallel_site_writer_same_text_as_single_stream(cli, tmp_path, golden_dir):
    """The per-site file is the reference's single zlib stream, byte for byte, also when it is produced on several
    threads (pgz::Stream).  The fallback for text pgz declines — concatenated gzip members above a size threshold — has
    different .gz bytes, identical decompressed bytes (and is still one valid gzip file)."""
    d = os.path.join(golden_dir, "f3")
    for name, env in (("serial", {}), ("par", {"PANDEPTH_TUNE": "site_identical=0,site_parallel_min=1"})):
        p = subprocess.run([cli, "-i", "tiny.bam", "-w", "100", "-a", "-t", "4", "-o", str(tmp_path / name)], cwd=d,
                           env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-300:]
    a = (tmp_path / "serial.SiteDepth.gz").read_bytes()
    b = (tmp_path / "par.SiteDepth.gz").read_bytes()
    assert a != b and gzip.decompress(a) == gzip.decompress(b)
    assert (tmp_path / "serial.win.stat.gz").read_bytes() == (tmp_path / "par.win.stat.gz").read_bytes()
    import hashlib, json
    m = [e for e in json.load(open(os.path.join(golden_dir, "manifest.json"))) if e["fixture"] == "f3" and e["name"] == "w100_a"][0]
    assert hashlib.sha256(gzip.decompress(b)).hexdigest() == m["outputs"]["SiteDepth.gz"]["text_sha256"]
    assert hashlib.sha256(a).hexdigest() == m["outputs"]["SiteDepth.gz"]["gz_sha256"]


def test_parallel_site_writer_backpressure(cli, tmp_path):
    """More chunks than the writer keeps in flight (regression: the submitter must keep draining)."""
    sam = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:30000000\nr\t0\tc\t1000\t60\t100M\t*\t0\t0\t*\t*\n"
    (tmp_path / "big.sam").write_text(sam)
    p = subprocess.run([cli, "-i", "big.sam", "-a", "-t", "2", "-o", "o"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, PANDEPTH_TUNE="site_identical=0"))
    assert p.returncode == 0, p.stderr.decode()[-300:]
    import zlib
    d = zlib.decompressobj(31)
    n_lines, members, tail = 0, 1, b""
    raw = (tmp_path / "o.SiteDepth.gz").read_bytes()
    while raw:
        out = d.decompress(raw)
        n_lines += out.count(b"\n")
        if out:
            tail = out[-40:]
        raw = d.unused_data
        if raw:
            d = zlib.decompressobj(31); members += 1
    assert n_lines == 30000000 and members >= 7 and tail.endswith(b"c\t29999999\t0\n")


def test_site_writer_identical_stream_at_size(cli, tmp_path):
    """30 M per-site lines (~400 MB of text, several pgz rounds with PANDEPTH-independent defaults): the file written on 4
    threads equals, byte for byte, the single zlib stream of the same text"""
    import zlib
    sam = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c\tLN:30000000\n" + "".join(
        "r%d\t0\tc\t%d\t60\t100M\t*\t0\t0\t*\t*\n" % (i, 1000 + 37 * i) for i in range(20000))
    (tmp_path / "big.sam").write_text(sam)
    p = subprocess.run([cli, "-i", "big.sam", "-a", "-t", "4", "-o", "o"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    raw = (tmp_path / "o.SiteDepth.gz").read_bytes()
    text = gzip.decompress(raw)
    assert text.count(b"\n") == 30000000 and text.endswith(b"c\t29999999\t0\n")
    co = zlib.compressobj(-1, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY)
    ref = co.compress(text) + co.flush()
    # zlib.compressobj writes OS = 3 and mtime 0 like gzopen: the whole file must match
    assert raw == ref


def test_csi_index_is_used_like_bai(cli, tmp_path, golden_dir):
    """Only a .csi next to the BAM (built by htslib, committed as a fixture): indexed semantics (uint32
    cells, region fetch) and range-partitioned readers must give the golden tables of the .bai runs."""
    import hashlib, json, shutil
    shutil.copy(os.path.join(golden_dir, "f3", "tiny.bam"), tmp_path / "tiny.bam")
    shutil.copy(os.path.join(golden_dir, "f3_csi", "tiny.bam.csi"), tmp_path / "tiny.bam.csi")
    shutil.copy(os.path.join(golden_dir, "f3", "tiny.gff"), tmp_path / "tiny.gff")
    man = {e["name"]: e for e in json.load(open(os.path.join(golden_dir, "manifest.json"))) if e["fixture"] == "f3"}
    for name, args, suffix in (("chr", [], "chr.stat.gz"), ("gff", ["-g", "tiny.gff"], "gene.stat.gz"), ("w1000", ["-w", "1000"], "win.stat.gz")):
        p = run(cli, ["-i", "tiny.bam", "-o", name, "-t", "3"] + args, tmp_path)
        assert p.stdout.decode() == man[name]["stdout"]          # no "No Index mode" warning
        assert hashlib.sha256((tmp_path / (name + "." + suffix)).read_bytes()).hexdigest() == man[name]["outputs"][suffix]["gz_sha256"]


def test_gc_column_of_small_windows_is_the_real_count(cli, tmp_path, golden_dir):
    """-c with -w < 150: the reference frees its sequences before the sweep reads them (PD:4097 / PD:4327) and prints
    heap contents; here the column holds the window's G/C count.  Every other column equals the run without -c, and the
    per-window counts add up to the whole-chromosome table's (which IS pinned against the reference, fixture f5)."""
    f5 = os.path.join(golden_dir, "f5")
    for fn in ("c.bam", "c.bam.bai", "c.fa"):
        os.symlink(os.path.join(f5, fn), tmp_path / fn)
    run(cli, ["-i", "c.bam", "-o", "g", "-w", "100", "-c", "-r", "c.fa"], tmp_path)
    run(cli, ["-i", "c.bam", "-o", "n", "-w", "100"], tmp_path)
    g = [l.split("\t") for l in stat(tmp_path / "g.win.stat.gz").strip().split("\n")]
    n = [l.split("\t") for l in stat(tmp_path / "n.win.stat.gz").strip().split("\n")]
    assert g[0][6] == "GC(%)" and [x[:6] + x[7:] for x in g[:-1]] == n[:-1]
    seqs, name = {}, None
    for line in open(os.path.join(f5, "c.fa"), "rb").read().decode().split("\n"):
        line = line.rstrip("\r")
        if line.startswith(">"):
            name = line[1:].split()[0]; seqs[name] = ""
        elif name:
            seqs[name] += line
    total = 0
    for row in g[1:-1]:
        s = seqs[row[0]][int(row[1]) - 1:int(row[2])]
        cnt = sum(s.count(c) for c in "CcGg")
        total += cnt
        assert row[6] == "%.2f" % (cnt * 100.0 / int(row[3]))
    L = sum(int(r[3]) for r in g[1:-1])
    assert g[-1][2] == "GC(%%): %.2f" % (total * 100.0 / L)


def test_gc_needs_a_readable_reference(cli, tmp_path, golden_dir):
    f5 = os.path.join(golden_dir, "f5")
    os.symlink(os.path.join(f5, "c.bam"), tmp_path / "c.bam")
    p = subprocess.run([cli, "-i", "c.bam", "-o", "x", "-c", "-r", "missing.fa"], cwd=tmp_path, capture_output=True, timeout=60)
    assert p.returncode == 1 and b"Cannot open the reference sequence file" in p.stderr     # (the reference spins forever here)
    p = subprocess.run([cli, "-i", "c.bam", "-o", "x", "-c"], cwd=tmp_path, capture_output=True, timeout=60)
    assert p.returncode == 0 and b"lack reference sequence (-r) for GC parse" in p.stderr    # PD:3530-3533
    assert not os.path.exists(tmp_path / "x.chr.stat.gz")


def test_paf_lines_the_reference_cannot_answer(cli, tmp_path):
    """tstart = 0 without a cg:Z: tag (the reference starts one cell BEFORE its array), a line with fewer than 12 columns
    (it indexes past its vector) and a cg:Z: value without a number (std::stoi throws): the first is clipped to the
    target, the other two lines are skipped; everything else on the file is counted."""
    good = ["q1\t100\t0\t50\t+\ttg\t300\t0\t50\t50\t50\t60\ttp:A:P",                     # cells [0, 50) after clipping -1
            "q2\t100\t0\t50\t+\ttg\t300\t100\t160\t60\t60\t60\ttp:A:P\tcg:Z:30M10D20M",   # [100,130) + [140,160)
            "q3\t100\t0\t50\t+\ttg\t300\t10\t20",                                            # short line
            "q4\t100\t0\t50\t+\ttg\t300\t200\t260\t60\t60\t60\ttp:A:P\tcg:Z:M30"]          # malformed cg
    (tmp_path / "u.paf").write_text("\n".join(good) + "\n")
    p = run(cli, ["-i", "u.paf", "-o", "u", "-a"], tmp_path)
    assert p.stdout.decode() == "INFO: Run paf Format data \nINFO: Input data read done\n"
    d = [int(l.split("\t")[2]) for l in stat(tmp_path / "u.SiteDepth.gz").strip().split("\n")]
    want = [0] * 300
    for a, b in ((0, 50), (100, 130), (140, 160)):
        for k in range(a, b):
            want[k] += 1
    assert d == want
    assert stat(tmp_path / "u.chr.stat.gz").split("\n")[1] == "tg\t300\t100\t100\t33.33\t0.33"


def test_damaged_index_is_not_trusted(cli, tmp_path, golden_dir):
    """a .bai whose reference count has a flipped high byte (it asked for 64 GB once): the file is read without it, the
    reads the indexed path selects are the same, so is the table"""
    f1 = os.path.join(golden_dir, "f1")
    for fn in ("f1.bam", "f1.gff"):
        os.symlink(os.path.join(f1, fn), tmp_path / fn)
    bai = bytearray(open(os.path.join(f1, "f1.bam.bai"), "rb").read())
    (tmp_path / "good").mkdir()
    os.symlink(os.path.join(f1, "f1.bam"), tmp_path / "good" / "f1.bam")
    os.symlink(os.path.join(f1, "f1.gff"), tmp_path / "good" / "f1.gff")
    (tmp_path / "good" / "f1.bam.bai").write_bytes(bytes(bai))
    bai[7] = 160
    (tmp_path / "f1.bam.bai").write_bytes(bytes(bai))
    for extra in ([], ["-g", "f1.gff"]):
        run(cli, ["-i", "f1.bam", "-o", "o", "-t", "3"] + extra, tmp_path)
        run(cli, ["-i", "f1.bam", "-o", "o", "-t", "3"] + extra, tmp_path / "good")
        name = "o.gene.stat.gz" if extra else "o.chr.stat.gz"
        assert (tmp_path / name).read_bytes() == (tmp_path / "good" / name).read_bytes()
