"""pandepth_amd/host/pgzip.cpp: the byte stream of zlib's gzopen/gzwrite/gzclose, produced with the LZ77
parse spread over threads.  Every case compares with the system zlib's own stream (tests/harness/pgzip_check
writes the same text through gzopen the way GzWriter does): identical bytes, or an explicit "declined"
(the caller then uses zlib's serial stream) — never different bytes."""
import glob
import gzip
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MANIFEST_DIR = os.path.join(HERE, "golden")


@pytest.fixture(scope="session")
def check():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "pgzip_check", "pandepth_oracle_cli"], check=True,
                   stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "harness", "pgzip_check")


def verdict(check, path, *cfg):
    r = subprocess.run([check, str(path)] + [str(c) for c in cfg], capture_output=True, text=True, timeout=600)
    word = r.stdout.split()[0] if r.stdout.split() else "?"
    assert word in ("identical", "declined"), r.stdout + r.stderr          # "DIFFERENT" must never happen
    assert (r.returncode == 0) == (word == "identical")
    return word


def test_every_golden_table_is_reproduced(check, tmp_path):
    """all reference outputs committed under tests/golden (tables and per-site files)"""
    files = sorted(glob.glob(os.path.join(MANIFEST_DIR, "*", "expected", "*.gz")))
    assert len(files) >= 50
    for f in files:
        p = tmp_path / "t.txt"
        p.write_bytes(gzip.decompress(open(f, "rb").read()))
        assert verdict(check, p, 2) == "identical", f
        assert verdict(check, p, 3, 65536, 2048) == "identical", f


def window_table(rows, seed):
    rng = random.Random(seed)
    out, chrom, pos = [], 1, 1
    for i in range(rows):
        if rng.random() < 0.0005:
            chrom += 1; pos = 1
        ln = 1000
        cov = min(ln, max(0, int(rng.gauss(900, 150))))
        dep = int(cov * max(0.0, rng.gauss(48, 9)))
        out.append("Chr%02d\t%d\t%d\t%d\t%d\t%d\t%.2f\t%.2f\n" % (chrom, pos, pos + ln - 1, ln, cov, dep, cov * 100.0 / ln, dep / ln))
        pos += ln
    return "".join(out).encode()


def site_table(lines, seed):
    rng = random.Random(seed)
    return "".join("Chr%02d\t%d\t%d\n" % (1 + i // 700000, i % 700000, max(0, int(rng.gauss(50, 12)))) for i in range(lines)).encode()


@pytest.mark.parametrize("cfg", [(1,), (4,), (8, 65536, 2048), (3, 100000, 3000), (5, 262144, 8192)], ids=str)
def test_generated_tables_stitched_across_chunks(check, tmp_path, cfg):
    for name, data in (("win", window_table(150000, 11)), ("site", site_table(600000, 12))):
        p = tmp_path / (name + ".txt")
        p.write_bytes(data)
        assert verdict(check, p, *cfg) == "identical", name


def test_streaming_interface_in_irregular_pieces(check, tmp_path):
    """pgz::Stream fed with writes of 1 byte .. 3 MB and rounds of 2 MB: same bytes as one zlib stream"""
    p = tmp_path / "s.txt"
    p.write_bytes(site_table(900000, 21) + window_table(60000, 22))
    for cfg in ((4, 65536, 4096, 2000000), (3, 262144, 16384, 1), (8,)):
        r = subprocess.run([check, str(p)] + [str(c) for c in cfg], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, PGZ_PIECES="1"))
        assert r.stdout.split()[0] == "identical", r.stdout


def test_edge_cases(check, tmp_path):
    cases = {
        "empty": (b"", "identical"),
        "one": (b"x", "identical"),
        "short": (b"#Chr\tLength\n", "identical"),
        "below_one_chunk": (window_table(500, 3), "identical"),
        # zlib stores incompressible blocks; that branch is not re-stated -> declined
        "random": (os.urandom(400000), "declined"),
        # the parses of neighbouring chunks never meet on a pure run (258-byte matches at different phases): every successor is parsed
        # again from its predecessor's last match end (emit_round), and the stream is zlib's
        "zeros": (b"\0" * 700000, "identical"),
    }
    for name, (data, want) in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        assert verdict(check, p, 4, 65536, 2048) == want, name
        if name == "zeros":       # rounds of 100 KB: the pairs at the rounds' boundaries are mended from the text the stream keeps of the round before
            assert verdict(check, p, 4, 65536, 2048, 100000) == want, name
    # mixtures may go either way, but never to different bytes
    rng = random.Random(9)
    for k in range(6):
        parts = []
        for _ in range(rng.randrange(2, 6)):
            kind = rng.randrange(4)
            n = rng.randrange(1000, 300000)
            parts.append([window_table(n // 50 + 1, rng.random()), site_table(n // 15 + 1, rng.random()), os.urandom(n // 8),
                          bytes([rng.randrange(97, 101) for _ in range(n)])][kind])
        p = tmp_path / ("mix%d" % k)
        p.write_bytes(b"".join(parts))
        verdict(check, p, 4, 65536, 2048)
        verdict(check, p, 2, 131072, 4096)


def test_remote_text_source_rounds_release_and_mends(check, tmp_path):
    """pgz::Stream over a Remote source (what the engine's resident text is to it): the harness answers parse / fetch from the file's bytes and REFUSES
    anything below the release mark, as the device's ring does.  Small rounds, so that hand-overs at the rounds' boundaries occur; a pure run and a pure
    period, whose pairs of chunk parses never meet (every successor is parsed again from text the stream must still be able to fetch), and per-site rows."""
    rng = random.Random(4)
    per = bytes(rng.randint(0, 255) for _ in range(37))
    cases = {"zeros": b"\0" * 700000, "period": per * 50000 + bytes(rng.randint(0, 255) for _ in range(900)) + per * 20000, "site": site_table(400000, 5)}
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        for cfg in ((4, 16384, 4096, 200000), (3, 16384, 2048, 100000), (4, 32768, 4096, 1000000)):
            r = subprocess.run([check, str(p)] + [str(c) for c in cfg], capture_output=True, text=True, timeout=600, env=dict(os.environ, PGZ_REMOTE="1", PGZ_DEBUG="1"))
            assert r.stdout.split()[0] == "identical", (name, cfg, r.stdout, r.stderr[-300:])
            if name != "site":
                # the pairs were mended from fetched text — since round 5 a round's pairs from ONE fetch (a Remote fetch is a device synchronise)
                import re
                assert int(r.stderr.split("remote source: ")[1].split()[0]) >= 1, r.stderr
                assert sum(int(m) for m in re.findall(r"(\d+) parsed again where a pair did not meet", r.stderr)) > 10, r.stderr[-600:]


def test_four_provider_calls_of_a_round_in_flight(check, tmp_path):
    """pgz::Params::calls (round 6): a round's chunks split over four provider calls running side by side (the engine's four parse slots) instead of two —
    the stream's bytes must not depend on how a round is shared out.  Native parse (host emulation of the device's), 8 KiB chunks, rounds of 12 MiB = 1 536
    chunks: four groups of 384; with two and with one call the same bytes."""
    p = tmp_path / "site"
    p.write_bytes(site_table(1500000, 9))
    outs = []
    for calls in ("4", "2", "1"):
        r = subprocess.run([check, str(p), "4", "8192", "2048", str(12 << 20)], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, PGZ_NATIVE="1", PGZ_CALLS=calls))
        assert r.stdout.split()[0] == "identical", (calls, r.stdout, r.stderr[-300:])
        outs.append(r.stdout.split()[0])
    assert len(set(outs)) == 1


def test_cli_tables_through_the_parallel_writer(check, tmp_path):
    """the host pipeline with the writer forced onto pgz for every output (pgz_min=0 in PANDEPTH_TUNE): the golden gz bytes"""
    import json
    import hashlib
    cli = os.path.join(HERE, "harness", "pandepth_oracle_cli")
    manifest = json.load(open(os.path.join(MANIFEST_DIR, "manifest.json")))
    picked = [e for e in manifest if e["name"] in ("chr", "w100", "w200_a", "gff_a", "bed3", "list3", "w100_a", "gff")][:10]
    assert len(picked) >= 5
    for case in picked:
        d = os.path.join(MANIFEST_DIR, case["fixture"])
        out = tmp_path / (case["fixture"] + "_" + case["name"])
        args = [cli] + case["args"] + ["-o", str(out)]
        if "-t" not in case["args"]:
            args += ["-t", "4"]
        p = subprocess.run(args, cwd=d, capture_output=True, timeout=600, env=dict(os.environ, PANDEPTH_TUNE="pgz_min=0"))
        assert p.returncode == case["returncode"]
        for suffix, meta in case["outputs"].items():
            gz = open(str(out) + "." + suffix, "rb").read()
            assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], (case["name"], suffix)


def test_fuzz_skewed_alphabets_and_texts(check, tmp_path):
    """length-limited Huffman repair (Fibonacci frequencies push codes past 15 bits), sparse alphabets, repeated
    records with point mutations, digit streams: never a different byte"""
    rng = random.Random(2024)

    def fib_text(nsym, total):
        f = [1, 1]
        while len(f) < nsym:
            f.append(f[-1] + f[-2])
        scale = total / sum(f)
        syms = []
        for a, c in zip(range(33, 33 + nsym), f):
            syms += [a] * max(1, int(c * scale))
        rng.shuffle(syms)
        return bytes(syms)

    def gen():
        k = rng.randrange(6)
        if k == 0:
            return fib_text(rng.randrange(18, 40), rng.randrange(5000, 60000))
        if k == 1:
            return bytes(rng.choices(range(256), weights=[1 / (i + 1) ** rng.uniform(0.5, 3) for i in range(256)], k=rng.randrange(100, 80000)))
        if k == 2:
            words = [bytes(rng.choices(range(97, 123), k=rng.randrange(2, 12))) for _ in range(rng.randrange(5, 400))]
            return b" ".join(rng.choices(words, k=rng.randrange(50, 30000)))
        if k == 3:
            return b"".join(b"%d\t%d\t%.2f\n" % (rng.randrange(10 ** rng.randrange(1, 9)), i, rng.random() * 100) for i in range(rng.randrange(10, 20000)))
        if k == 4:
            return fib_text(rng.randrange(25, 45), rng.randrange(20000, 200000)) + bytes(rng.choices(range(65, 91), k=rng.randrange(0, 5000)))
        base = bytes(rng.choices(range(97, 110), k=rng.randrange(20, 300)))
        out = bytearray()
        for _ in range(rng.randrange(10, 3000)):
            b = bytearray(base)
            for _ in range(rng.randrange(0, 4)):
                b[rng.randrange(len(b))] = rng.randrange(97, 123)
            out += b + b"\n"
        return bytes(out)

    seen = {"identical": 0, "declined": 0}
    for it in range(80):
        p = tmp_path / "f.bin"
        p.write_bytes(gen())
        seen[verdict(check, p, *rng.choice([(2,), (3, 65536, 2048), (4, 70000, 2500)]))] += 1
    assert seen["identical"] >= 70


def test_exact_two_decimal_formatting_without_printf():
    """pdh::fmt2_to (the tables' "%.2f") against snprintf on 1.8e7 values: table ratios, exact ties (odd multiples of
    1/8) and their neighbours, x.xx5 decimals, powers of two, denormals, the edges of the fast path, random bit patterns"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "fmt_check"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(HERE, "harness", "fmt_check")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-600:]
