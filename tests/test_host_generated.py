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
        env.update(PANDEPTH_DD_BATCH_MB="2")
    elif cli.endswith(":host"):        # ... or not at all: the host readers (libdeflate) feed pd_push_intervals
        cli = cli[:-5]
        env.update(PANDEPTH_DEVICE_DECODE="0")
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
