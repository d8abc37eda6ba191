"""The speculative BAM record walk shared by the host build and the gfx950 kernels (pandepth_amd/csrc/pd_bamwalk.h)
with its 64 lanes emulated on the host: the same reads and runs as the sequential host reader on the fixture and
generated BAMs (several filters, segment sizes, the near/far split), and the host half of the speculation
(pdb2::check_chain) against a model walker with wrong guesses planted."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

H = os.path.join(HERE, "harness")
FIX = [os.path.join(HERE, "golden", d, f) for d, f in (("f1", "f1.bam"), ("f1", "f1_unsorted.bam"), ("f2", "f2.bam"), ("f3", "tiny.bam"))]


@pytest.fixture(scope="module")
def walker():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(H, "bamwalk_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(H, "bamwalk_check.cpp"),
                    os.path.join(ROOT, "pandepth_amd", "libpandepth_host.a"), "-lz", "-ldl", "-o", exe], check=True)
    return exe


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    d = tmp_path_factory.mktemp("walk")
    names, lens = synth.genome_c2(scale=0.001)
    rec = synth.gen_records_numpy(lens, 60000, seed=21)
    out = []
    for payload in (False, True):
        p = str(d / ("w%d.bam" % payload))
        synth.write_bam(p, names, lens, rec, procs=1, payload=payload, level=1)
        out.append(p)
    return out


@pytest.mark.parametrize("opts", [[], ["-x", "0"], ["-q", "30"], ["-seg", "4"], ["-seg", "1", "-x", "256"], ["-near", "1024"]], ids=lambda o: "_".join(o) or "default")
def test_walk_core_equals_host_reader(walker, generated, opts):
    p = subprocess.run([walker] + opts + FIX + generated, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, (p.stdout.decode()[-1500:], p.stderr.decode()[-400:])
    assert b"DIFFERENT" not in p.stdout and p.stdout.count(b"identical") == 2 * len(FIX + generated)


def test_chain_check_against_model_walker():
    exe = os.path.join(H, "chain_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(H, "chain_check.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b" 0 failures" in p.stdout, (p.stdout.decode()[-400:], p.stderr.decode()[-1500:])
