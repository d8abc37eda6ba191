"""zlib's level-6 LZ77 parse restated for one wave per position (pandepth_amd/csrc/pd_lz77.h; the reference writes its tables and
per-site files through one zlib stream — gzstream, PD:4264-4284 — so the .gz bytes are a function of that parse): the host
emulation (64 lanes in a loop) and, on the GPU, pd_deflate_parse through the C-ABI, both against the symbols ZLIB ITSELF produces
for the same chunks (deflate primed with the chunk's dictionary, symbols read back out of its stream)."""
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
H = os.path.join(HERE, "harness")


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("lz")
    rnd = random.Random(5)
    # a per-site file (PD:4278: "<chr>\t<index>\t<depth>\n") with depths that wander
    dep, rows = 50, []
    for j in range(400000):
        if rnd.random() < 0.3:
            dep = max(0, dep + rnd.choice((-2, -1, -1, 1, 1, 2)))
        rows.append("Chr01\t%d\t%d\n" % (j, dep))
    (d / "site.txt").write_text("".join(rows))
    # a window table (PD:4381-4388)
    rows = ["#Chr\tStart\tEnd\tLength\tCoveredSite\tTotalDepth\tCoverage(%)\tMeanDepth\n"]
    for k in range(80000):
        c, t = rnd.randint(0, 100), rnd.randint(0, 9000)
        rows.append("Chr%02d\t%d\t%d\t100\t%d\t%d\t%.2f\t%.2f\n" % (1 + k // 9000, 1 + 100 * k, 100 * k + 100, c, t, c * 1.0, t / 100.0))
    (d / "win.txt").write_text("".join(rows))
    A = b"ACDEFGHIKLMNPQRS"
    rb = lambda n: bytes(rnd.choice(A) for _ in range(n))
    # repeats at distances around MAX_DIST (32506), 3-byte repeats around TOO_FAR (4096), matches of nice length and of 258
    out = bytearray()
    for _ in range(40):
        blk = rb(rnd.choice((3, 4, 5, 6, 20, 130, 258, 300, 600)))
        dist = rnd.choice((32500, 32504, 32505, 32506, 32507, 32508, 32600, 4090, 4095, 4096, 4097, 4100, 100, 9, 1))
        out += blk + rb(max(0, dist - len(blk))) + blk + rb(rnd.randint(10, 3000))
    (d / "far.txt").write_bytes(bytes(out))
    # one trigram thousands of times (chains beyond the 128 / 32 budgets), runs of one byte, a short period
    s = bytearray()
    for k in range(60000):
        s += b"ABC" + bytes([rnd.choice(A)]) * rnd.randint(0, 3)
        if k % 97 == 0:
            s += b"ABCDEFGHIJKLMNOPQRSTUVWXYZ" * rnd.randint(1, 12)
    (d / "chains.txt").write_bytes(bytes(s))
    (d / "runs.bin").write_bytes(b"".join(bytes([rnd.randint(0, 255)]) * rnd.randint(1, 600) for _ in range(6000)))
    per = bytes(rnd.randint(0, 255) for _ in range(37))
    (d / "per.bin").write_bytes(per * 20000 + rb(1000) + per * 9000)
    (d / "dna.txt").write_text("".join(rnd.choice("ACGT") for _ in range(1500000)))
    subprocess.run(["make", "-C", H, "lz77_check"], check=True, stdout=subprocess.DEVNULL)
    return d


FILES = ["site.txt", "win.txt", "far.txt", "chains.txt", "runs.bin", "per.bin", "dna.txt"]
GEOMETRIES = [("262144", "16384"), ("65536", "2048"), ("100000", "5000")]


@pytest.mark.parametrize("geo", GEOMETRIES, ids=lambda g: "chunk%s_tail%s" % g)
@pytest.mark.parametrize("name", FILES)
def test_wave_parse_equals_zlib_on_the_host(corpus, name, geo):
    p = subprocess.run([os.path.join(H, "lz77_check"), str(corpus / name), geo[0], geo[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and " 0 chunks differ" in p.stdout, p.stdout + p.stderr[-1500:]
    assert int(p.stdout.split(" chunks, ")[1].split(" symbols")[0]) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("geo", GEOMETRIES[:2], ids=lambda g: "chunk%s_tail%s" % g)
@pytest.mark.parametrize("name", FILES)
def test_device_parse_equals_zlib(corpus, name, geo):
    """pd_deflate_parse (sort by hash on the device, one wave per chunk) = zlib's symbols, chunk by chunk"""
    subprocess.run(["make", "-C", H, "lz77_gpu_check"], check=True, stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(H, "lz77_gpu_check"), str(corpus / name), geo[0], geo[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0 and " 0 chunks differ" in p.stdout, p.stdout + p.stderr[-1500:]


@pytest.mark.gpu
@pytest.mark.parametrize("geo", [("16384", "4096"), ("65536", "2048")], ids=lambda g: "chunk%s_tail%s" % g)
@pytest.mark.parametrize("name", FILES)
def test_device_parse_with_the_text_in_lds_equals_zlib(corpus, name, geo):
    """the same with "lz_group" = 16: consecutive chunks parsed by one workgroup out of the LDS copy of their text (k_lz_parse_lds; nine
    chunks of 16 + 4 KiB a workgroup, one of 64 + 2 KiB) — opt-in in the product, zlib's symbols all the same"""
    subprocess.run(["make", "-C", H, "lz77_gpu_check"], check=True, stdout=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(H, "lz77_gpu_check"), str(corpus / name), geo[0], geo[1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env=dict(os.environ, LZ_GROUP="16", PD_LZ_DEBUG="1"))
    assert p.returncode == 0 and " 0 chunks differ" in p.stdout, p.stdout + p.stderr[-1500:]
    groups = [int(l.split()[1]) for l in p.stderr.splitlines() if l.startswith("[lz] ") and " groups of up to " in l]
    assert groups and max(groups) > 0, p.stderr[-1500:]


@pytest.mark.gpu
def test_cli_per_site_and_window_files_through_the_device_parse(tmp_path):
    """`pandepth -w 100 -a`: both gzip streams with stage 1 on the GPU (pd_deflate_parse) — the same bytes as with zlib parsing on the
    host threads (device_deflate=0 in PANDEPTH_TUNE) and as the reference binary's files where it is present."""
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    names, lens = synth.genome_c2(scale=0.002)
    rec = synth.gen_records_numpy(lens, 100000, seed=5)
    bam = str(tmp_path / "g.bam")
    synth.write_bam(bam, names, lens, rec, procs=2, payload=False)
    subprocess.run([os.path.join(ROOT, "pandepth_amd", "pandepth_index"), bam], check=True)
    cli = os.path.join(ROOT, "pandepth_amd", "pandepth")
    outs = {}
    for tag, env in (("dev", {"PANDEPTH_TIMING": "1", "PANDEPTH_TUNE": "table_resident_min=1"}),
                     ("hosttext", {"PANDEPTH_TIMING": "1", "PANDEPTH_TUNE": "site_resident=0,table_resident=0"}), ("host", {"PANDEPTH_TUNE": "device_deflate=0"})):
        p = subprocess.run([cli, "-i", "g.bam", "-w", "100", "-a", "-o", tag, "-t", "8"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=900, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-600:]
        outs[tag] = {s: (tmp_path / ("%s.%s" % (tag, s))).read_bytes() for s in ("win.stat.gz", "SiteDepth.gz")}
        err = p.stderr.decode()
        if tag == "dev":          # the per-site text and the table's rows stay on the device (pd_text_*)
            assert "per-site writer (text resident on the device)" in err and " 0 parse calls" not in err and "FAILED" not in err, err[-1500:]
            assert "window table (text resident on the device)" in err and "rows, written" in err, err[-1500:]
        if tag == "hosttext":     # both streams through pd_deflate_parse on host text
            assert "text resident on the device" not in err and err.count("pd_deflate_parse:") >= 2 and "FAILED" not in err, err[-1500:]
    assert outs["dev"] == outs["hosttext"]
    assert outs["dev"] == outs["host"]
    ref = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
    if os.access(ref, os.X_OK):
        subprocess.run([ref, "-i", "g.bam", "-w", "100", "-a", "-o", "ref", "-t", "4"], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for s in ("win.stat.gz", "SiteDepth.gz"):
            assert outs["dev"][s] == (tmp_path / ("ref." + s)).read_bytes(), s


def test_resident_text_stream_through_the_host_logic(tmp_path):
    """host/pgzip.cpp with a Remote text source (the per-site file whose text stays with the engine: pd_text_*), driven on the CPU
    by the oracle engine's stand-in (tests/harness/oracle_engine.cpp: a byte vector, the product's parse core in host emulation):
    several rounds, a ring that runs full and is waited for, a pair of neighbouring parses that does not meet inside its overlap
    (mended in place: the successor is parsed again by zlib from the hand-over point) — the same bytes as the host-text path and as zlib-only."""
    import sys
    sys.path.insert(0, ROOT)
    from tools import synth
    subprocess.run(["make", "-C", H, "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    names, lens = synth.genome_c2(scale=0.0004)
    rec = synth.gen_records_numpy(lens, 20000, seed=6)
    bam = str(tmp_path / "g.bam")
    synth.write_bam(bam, names, lens, rec, procs=2, payload=False)
    cli = os.path.join(H, "pandepth_oracle_cli")
    outs = {}
    for tag, env in (("resident", {"PANDEPTH_TIMING": "1", "PGZ_DEV_BATCH_MB": "1", "PGZ_DEV_CHUNK_KB": "32", "PANDEPTH_TUNE": "table_resident_min=1"}),
                     ("full", {"PANDEPTH_TIMING": "1", "PGZ_DEV_BATCH_MB": "1", "PANDEPTH_TEST_TEXT_CAP": "4000000"}),
                     ("flaky", {"PANDEPTH_TIMING": "1", "PGZ_DEV_BATCH_MB": "1", "PANDEPTH_TEST_TEXT_PARSE_FAIL": "3", "PANDEPTH_TUNE": "table_resident_min=1"}),
                     ("mended", {"PANDEPTH_TIMING": "1", "PGZ_DEBUG": "1", "PGZ_DEV_BATCH_MB": "1", "PGZ_DEV_TAIL_KB": "2", "PANDEPTH_TUNE": "table_resident_min=1"}),
                     ("hosttext", {"PANDEPTH_TUNE": "site_resident=0", "PGZ_DEV_BATCH_MB": "1"}), ("zlib", {"PANDEPTH_TEST_NO_PARSE": "1"})):
        p = subprocess.run([cli, "-i", "g.bam", "-w", "100", "-a", "-o", tag, "-t", "4"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=1500, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-600:]
        outs[tag] = {s: (tmp_path / ("%s.%s" % (tag, s))).read_bytes() for s in ("win.stat.gz", "SiteDepth.gz")}
        if tag in ("resident", "full"):
            err = p.stderr.decode()
            assert "per-site writer (text resident on the device)" in err and " 0 parse calls" not in err, err[-1500:]
        if tag == "resident":
            assert "window table (text resident on the device)" in err and "rows, written" in err, err[-1500:]
        if tag == "mended":       # 2 KiB of overlap: a pair of neighbouring parses does not meet in it — the successor is parsed again from the hand-over point
            import re
            err = p.stderr.decode()
            assert "per-site writer (text resident on the device)" in err and "per-site writer: producer" not in err, err[-1500:]
            assert sum(int(m) for m in re.findall(r"(\d+) parsed again", err)) >= 1, err[-1500:]
        if tag == "flaky":        # every third parse call of the engine fails: those chunks are fetched and parsed by zlib on the host
            err = p.stderr.decode()
            assert "pd_text_parse failed" in err and "per-site writer (text resident on the device)" in err, err[-1500:]
    assert outs["resident"] == outs["zlib"] and outs["full"] == outs["zlib"] and outs["hosttext"] == outs["zlib"] and outs["flaky"] == outs["zlib"]
    assert outs["mended"] == outs["zlib"]


def _pgz(check, path, env, *args):
    r = subprocess.run([check, str(path)] + [str(a) for a in args], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    return r.stdout.split()[0] if r.stdout.split() else r.stderr[-300:]


def test_gzip_streams_with_the_engines_parse_in_the_devices_geometry(tmp_path):
    """host/pgzip.cpp with stage 1 by the engine's parse (host emulation) in the geometry a device provider gets — 16 KiB chunks
    with 4 KiB of overlap, rounds of a few MiB running behind the writer, two provider calls per round — whole buffers and the
    streaming interface fed in irregular pieces: the bytes of zlib's single stream; texts whose parses cannot meet inside the
    overlap (a pure period) decline, never differ."""
    import random
    subprocess.run(["make", "-C", H, "pgzip_check"], check=True, stdout=subprocess.DEVNULL)
    check = os.path.join(H, "pgzip_check")
    rnd = random.Random(9)
    site = "".join("Chr%02d\t%d\t%d\n" % (1 + i // 400000, i % 400000, max(0, int(rnd.gauss(40, 15)))) for i in range(700000)).encode()
    win = "".join("Chr%02d\t%d\t%d\t100\t%d\t%d\t%.2f\t%.2f\n" % (1 + k // 9000, 1 + 100 * k, 100 * k + 100, c, t, c * 1.0, t / 100.0)
                  for k, c, t in ((k, rnd.randint(0, 100), rnd.randint(0, 9000)) for k in range(120000))).encode()
    zero = "".join("scaffold_%d\t%d\t0\n" % (i // 50000, i % 50000) for i in range(400000)).encode()
    for name, data in (("site", site), ("win", win), ("zero", zero)):
        f = tmp_path / (name + ".txt")
        f.write_bytes(data)
        for env in ({"PGZ_NATIVE": "1", "PGZ_DEV_BATCH_MB": "2"}, {"PGZ_NATIVE": "1", "PGZ_DEV_BATCH_MB": "1", "PGZ_PIECES": "1"},
                    {"PGZ_NATIVE": "1", "PGZ_DEV_CHUNK_KB": "32", "PGZ_DEV_TAIL_KB": "8", "PGZ_DEV_BATCH_MB": "3"}):
            assert _pgz(check, f, env, 4) == "identical", (name, env)
    per = bytes(rnd.randint(0, 255) for _ in range(37))
    f = tmp_path / "per.bin"
    f.write_bytes(per * 60000 + bytes(rnd.randint(0, 255) for _ in range(900)) + per * 20000)
    assert _pgz(check, f, {"PGZ_NATIVE": "1", "PGZ_DEV_BATCH_MB": "1"}, 4) in ("identical", "declined")
    # mutated records, digit streams, skewed alphabets: identical or declined
    seen = {"identical": 0, "declined": 0}
    for it in range(24):
        k = rnd.randrange(3)
        if k == 0:
            base = bytes(rnd.choices(range(97, 110), k=rnd.randrange(20, 300)))
            out = bytearray()
            for _ in range(rnd.randrange(2000, 12000)):
                b = bytearray(base)
                for _ in range(rnd.randrange(0, 4)):
                    b[rnd.randrange(len(b))] = rnd.randrange(97, 123)
                out += b + b"\n"
            data = bytes(out)
        elif k == 1:
            data = b"".join(b"%d\t%d\t%.2f\n" % (rnd.randrange(10 ** rnd.randrange(1, 9)), i, rnd.random() * 100) for i in range(rnd.randrange(20000, 90000)))
        else:
            data = bytes(rnd.choices(range(256), weights=[1 / (i + 1) ** rnd.uniform(0.5, 3) for i in range(256)], k=rnd.randrange(200000, 900000)))
        f = tmp_path / "fz.bin"
        f.write_bytes(data)
        seen[_pgz(check, f, {"PGZ_NATIVE": "1", "PGZ_DEV_BATCH_MB": "1"}, 3)] += 1
    assert seen["identical"] >= 18, seen
