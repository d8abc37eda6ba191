"""CRAM 2.1 / 3.0 / 3.1 input (pandepth_amd/host/cram.cpp): the reader against the SAM text of the committed fixture files (any box),
against freshly generated files written by the reference's htslib (dev container only), and its error paths."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
S2B = os.path.join(ROOT, "oracle", "_ref", "sam2bam")
F7 = os.path.join(HERE, "golden", "f7")


@pytest.fixture(scope="module")
def chk():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "cram_check", "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "harness", "cram_check")


def records(chk, path):
    p = subprocess.run([chk, path], capture_output=True, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    out = []
    for line in p.stdout.decode().split("\n"):
        f = line.split("\t")
        if len(f) == 5 and int(f[2]) & 4:
            f[3] = "-"                                   # CRAM stores no mapping quality for reads flagged unmapped
        out.append("\t".join(f))
    return out


@pytest.mark.parametrize("name", ["m.cram", "m_noidx.cram", "m_ref.cram", "m_v31.cram", "m_v21.cram"])
def test_reader_returns_the_records_of_the_sam_text(chk, name):
    """reference-free and reference-based files, bases / qualities / tags / mate fields present, multi-reference slices"""
    want = records(chk, os.path.join(F7, "m.sam"))
    assert len(want) > 1400
    assert records(chk, os.path.join(F7, name)) == want


def test_unsupported_and_damaged_files_are_errors_not_crashes(chk, tmp_path):
    good = open(os.path.join(F7, "m.cram"), "rb").read()
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    cases = {"v20.cram": good[:4] + b"\x02\x00" + good[6:], "v32.cram": good[:4] + b"\x03\x02" + good[6:],
             "cut_header.cram": good[:40], "cut_body.cram": good[:len(good) // 2],
             "flipped.cram": good[:3000] + bytes(b ^ 0x5a for b in good[3000:3400]) + good[3400:]}
    for fn, data in cases.items():
        (tmp_path / fn).write_bytes(data)
        p = subprocess.run([chk, str(tmp_path / fn)], capture_output=True, timeout=60)
        assert p.returncode in (0, 1), (fn, p.returncode)          # never a signal
        if fn.startswith("v"):
            assert p.returncode == 1 and b"is not supported (2.1, 3.0 and 3.1 only)" in p.stderr
        if fn.startswith("cut"):
            assert p.returncode == 1
        q = subprocess.run([cli, "-i", fn, "-o", "o"], cwd=tmp_path, capture_output=True, timeout=60)
        assert q.returncode >= 0, (fn, q.returncode)


@pytest.mark.skipif(not os.access(S2B, os.X_OK), reason="needs the reference's htslib (dev container only)")
def test_generated_crams_match_their_sam_text(chk):
    r = subprocess.run([sys.executable, os.path.join(HERE, "cram_vs_sam.py"), "11", "11"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-1500:]


@pytest.mark.skipif(not os.access(S2B, os.X_OK), reason="needs the reference's htslib and binary (dev container only)")
def test_target_regions_step_over_containers_and_still_match_the_reference(chk, tmp_path):
    """a CRAM of many small containers + .crai, BED targets on a few places: containers whose stretch meets no (widened)
    target are never read, and the table is the reference's byte for byte"""
    import random
    rng = random.Random(5)
    contigs = [("u1", 200000), ("u2", 60000)]
    recs = []
    for i in range(6000):
        ci = rng.randrange(2)
        L = contigs[ci][1]
        cig = rng.choice(["100M", "40M200N60M", "10S90M", "50M5D50M", "30M3I67M"])
        span = sum(int(n) for n, o in __import__("re").findall(r"(\d+)([MDN])", cig))
        pos = rng.randrange(1, L - span)
        recs.append((ci, pos, "r%d\t%d\t%s\t%d\t%d\t%s\t*\t0\t0\t*\t*" % (i, rng.choice([0, 16, 1024]), contigs[ci][0], pos, rng.choice([0, 30, 60]), cig)))
    recs.sort()
    sam = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "\n".join(r[2] for r in recs) + "\n"
    (tmp_path / "t.sam").write_text(sam)
    subprocess.run([S2B, "t.sam", "t.cram", "sps=100"], cwd=tmp_path, check=True, capture_output=True)
    (tmp_path / "t.bed").write_text("u1\t5000\t5600\tA\nu1\t150000\t150100\tB\nu2\t1\t300\tC\nu2\t59000\t60000\tD\n")
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    for extra in ([], ["-a"], ["-q", "20"]):
        m = subprocess.run([cli, "-i", "t.cram", "-b", "t.bed", "-o", "m", "-t", "3"] + extra, cwd=tmp_path, capture_output=True,
                           env=dict(os.environ, PANDEPTH_TIMING="1"), timeout=120)
        r = subprocess.run([ref, "-i", "t.cram", "-b", "t.bed", "-o", "r", "-t", "3"] + extra, cwd=tmp_path, capture_output=True, timeout=120)
        assert m.returncode == 0 and r.returncode == 0 and m.stdout == r.stdout
        line = [l for l in m.stderr.decode().split("\n") if "containers decoded" in l][0]
        decoded, skipped = int(line.split(" containers decoded")[0].split()[-1]), int(line.split(" stepped over")[0].split()[-1])
        assert skipped > 40 and decoded < 15, line
        for suffix in ("bed.stat.gz",) + (("SiteDepth.gz",) if extra == ["-a"] else ()):
            assert (tmp_path / ("m." + suffix)).read_bytes() == (tmp_path / ("r." + suffix)).read_bytes(), suffix


@pytest.mark.skipif(not os.access(S2B, os.X_OK), reason="needs the reference's libhts.a (dev container only)")
def test_rans_nx16_decoder_against_the_htscodecs_encoder():
    """CRAM 3.1's block codecs (rANS Nx16, adaptive arithmetic coder): every transform combination round-trips through the
    encoders bundled in the reference's htslib"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "nx16_check"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HERE, "harness", "nx16_check")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 failures" in r.stdout, r.stdout[-1500:]


def test_core_bit_stream_codecs_and_varints(chk):
    """BETA / GAMMA / SUBEXP / canonical HUFFMAN on hand-made bit strings, ITF8 / LTF8 at their length boundaries"""
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "cram_bits_check"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HERE, "harness", "cram_bits_check")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "0 failures", r.stdout


@pytest.mark.skipif(not os.access(S2B, os.X_OK), reason="needs the reference's htslib and binary (dev container only)")
def test_region_reads_of_a_cram_are_the_reads_of_the_same_data_as_bam(chk, tmp_path):
    """Declared deviation.  With a .crai and GFF / BED targets, htslib's slice lookup (cram_index_query) does not return a
    read that sits in an earlier slice and spans into the target across slices that end before it (long N gaps; easy to
    provoke with tiny slices), so the reference reports less depth for the CRAM than for the same records written as
    BAM.  This reader selects reads per record, so its CRAM answer is the reference's BAM answer."""
    import random
    rng = random.Random(9)
    L = 6000
    recs = []
    for i in range(400):
        cig = rng.choice(["50M", "25M%dN25M" % rng.randrange(200, 2500), "40M3D10M", "10S40M"])
        span = sum(int(n) for n, o in __import__("re").findall(r"(\d+)([MDN])", cig))
        pos = rng.randrange(1, L - span)
        recs.append((pos, "r%d\t0\tz1\t%d\t60\t%s\t*\t0\t0\t*\t*" % (i, pos, cig)))
    recs.sort()
    (tmp_path / "t.sam").write_text("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:z1\tLN:%d\n" % L + "\n".join(r[1] for r in recs) + "\n")
    subprocess.run([S2B, "t.sam", "t.cram", "sps=3"], cwd=tmp_path, check=True, capture_output=True)
    subprocess.run([S2B, "t.sam", "t.bam"], cwd=tmp_path, check=True, capture_output=True)
    (tmp_path / "t.bed").write_text("z1\t3000\t3100\tA\nz1\t5200\t5300\tB\n")
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    subprocess.run([cli, "-i", "t.cram", "-b", "t.bed", "-a", "-o", "mine"], cwd=tmp_path, check=True, capture_output=True)
    subprocess.run([ref, "-i", "t.bam", "-b", "t.bed", "-a", "-o", "refbam"], cwd=tmp_path, check=True, capture_output=True)
    subprocess.run([ref, "-i", "t.cram", "-b", "t.bed", "-a", "-o", "refcram"], cwd=tmp_path, check=True, capture_output=True)
    for suffix in ("bed.stat.gz", "SiteDepth.gz"):
        assert (tmp_path / ("mine." + suffix)).read_bytes() == (tmp_path / ("refbam." + suffix)).read_bytes(), suffix
    import gzip
    depth = lambda fn: sum(int(l.split("\t")[2]) for l in gzip.open(tmp_path / fn).read().decode().strip().split("\n"))
    assert depth("refcram.SiteDepth.gz") <= depth("refbam.SiteDepth.gz")          # what the reference loses on the CRAM, if anything
