"""Long reads (the reference "supports … HIFI/CLR/ONT", README.md:149-150; its record loop takes any n_cigar, PD:440-460) through every
decode path: tools/bamgen --long writes 10-20 kb reads with an insertion or deletion every 8-15 bases (~2 600 CIGAR operations and
~1 300 covered runs per read), soft clips, N gaps of 50-180 kb (reference spans up to 200 kb) and 1 % of reads with MORE THAN 65 535
operations, stored as htslib stores them (<l_seq>S<ref_len>N in the record, the real CIGAR in the CG:B,I tag); records are 20-450 KB
and span BGZF members.  The expected files come from the compiled reference (oracle/_ref/pandepth_ref, htslib moves the CG CIGAR
back into place) on the same BAM + BAI; skipped where that binary does not exist.  Whole-chromosome, window, annotation and per-site
modes, with and without the index, device decode (default and small batches) and host decode."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")
GEN = os.path.join(ROOT, "tools", "bamgen")

pytestmark = pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (oracle/_ref/pandepth_ref)")


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("long")
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    if not os.access(GEN, os.X_OK) or os.path.getmtime(GEN) < os.path.getmtime(GEN + ".cpp"):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", GEN + ".cpp", "-lz", "-ldl", "-o", GEN], check=True)
    # 17.6 Mb genome (12 chromosomes of 1.0-2.0 Mb + scaffolds that get no reads), 1 200 long reads = 1x
    subprocess.run([GEN, "-o", str(d / "l.bam"), "-n", "1200", "-s", "0.006", "-t", "4", "--long", "-S", "11"], check=True, stderr=subprocess.DEVNULL)
    import random
    rng = random.Random(5)
    names = ["Chr%02d" % (i + 1) for i in range(12)]
    g = ["##gff-version 3"]
    for t in range(200):
        c = rng.choice(names)
        s = rng.randint(1, 900000)
        for e in range(rng.randint(1, 5)):
            el = rng.randint(60, 900)
            g.append("%s\tsyn\tCDS\t%d\t%d\t.\t+\t0\tID=c%d.%d;Parent=t%d" % (c, s, s + el - 1, t, e, t))
            s += el + rng.randint(50, 4000)
    (d / "l.gff").write_text("\n".join(g) + "\n")
    # regions on ONE chromosome: the per-site file of `-b ... -a` then holds that chromosome's 1.0e6 cells only
    (d / "l.bed").write_text("".join("Chr08\t%d\t%d\tr%d\n" % (20000 + 30000 * k, 20000 + 30000 * k + 4000, k) for k in range(30)))
    return d


CASES = [
    ("chr", ["-i", "l.bam"], ["chr.stat.gz"]),
    ("chr_q30_x0", ["-i", "l.bam", "-q", "30", "-x", "0"], ["chr.stat.gz"]),
    ("w100", ["-i", "l.bam", "-w", "100"], ["win.stat.gz"]),
    ("w5000", ["-i", "l.bam", "-w", "5000"], ["win.stat.gz"]),
    ("bed_a", ["-i", "l.bam", "-b", "l.bed", "-a"], ["bed.stat.gz", "SiteDepth.gz"]),
    ("gff", ["-i", "l.bam", "-g", "l.gff"], ["gene.stat.gz"]),
    ("noindex", ["-i", "l.bam", "-s"], ["chr.stat.gz"]),
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
@pytest.mark.parametrize("name,args,suffixes", CASES, ids=[c[0] for c in CASES])
def test_long_reads_match_reference(data, cli, name, args, suffixes):
    env = dict(os.environ)
    if cli.endswith(":dd"):
        cli = cli[:-3]
        env.update(PANDEPTH_TUNE="dd_batch_mb=2")
    elif cli.endswith(":host"):
        cli = cli[:-5]
        env.update(PANDEPTH_TUNE="device_decode=0")
    if not os.path.exists(data / ("ref_%s.%s" % (name, suffixes[0]))):
        subprocess.run([REF] + args + ["-o", "ref_" + name, "-t", "2"], cwd=data, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    p = subprocess.run([cli] + args + ["-o", "mine_" + name, "-t", "4"], cwd=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    for s in suffixes:
        assert (data / ("mine_%s.%s" % (name, s))).read_bytes() == (data / ("ref_%s.%s" % (name, s))).read_bytes(), s + " differs"
