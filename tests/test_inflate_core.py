"""The DEFLATE decoder shared by the host build and the gfx950 kernel (pandepth_amd/csrc/
pd_inflate_core.h): on the CPU every BGZF block of the fixture and generated BAMs must inflate to
exactly what zlib produces, and corrupted streams must end with an error, never a hang; on the GPU
(one lane per block) the same."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from tools import synth  # noqa: E402

FIX = [os.path.join(HERE, "golden", d, f) for d, f in (("f1", "f1.bam"), ("f1", "f1_unsorted.bam"), ("f2", "f2.bam"), ("f3", "tiny.bam"))]


@pytest.fixture(scope="module")
def generated(tmp_path_factory):
    d = tmp_path_factory.mktemp("inf")
    names, lens = synth.genome_c2(scale=0.001)
    rec = synth.gen_records_numpy(lens, 60000, seed=8)
    out = []
    for level in (1, 6, 9):
        p = str(d / ("p%d.bam" % level))
        synth.write_bam(p, names, lens, rec, procs=1, payload=True, level=level)
        out.append(p)
    return out


def test_decoder_equals_zlib_on_host(generated):
    exe = os.path.join(HERE, "harness", "inflate_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "harness", "inflate_check.cpp"), "-lz", "-o", exe], check=True)
    p = subprocess.run([exe] + FIX + generated, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert b" 0 mismatches" in p.stdout


def test_wave_decoder_equals_zlib_on_host(generated):
    """pd_inflate_wave.h (one wave per BGZF member: speculative parallel Huffman decode, token list, batched match
    copies) with its 64 lanes emulated on the host: every block type / strategy / level on generated streams, every
    block of the fixture and payload BAMs, and mutated streams (an error or zlib's bytes, never past the output)."""
    exe = os.path.join(HERE, "harness", "inflate_wave_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(HERE, "harness", "inflate_wave_check.cpp"), "-lz", "-o", exe], check=True)
    p = subprocess.run([exe, "-g", "-f", "2"] + FIX + generated, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, (p.stdout.decode()[-400:], p.stderr.decode()[-400:])
    assert b" 0 failures" in p.stdout and b" 0 mismatches" in p.stdout


@pytest.mark.gpu
def test_decoder_equals_zlib_on_gpu(generated):
    import gzip
    from pandepth_amd import capi
    for path in FIX + generated:
        data = open(path, "rb").read()
        ref = gzip.decompress(data)
        for variant in (0, 1, 2):            # 2 = the wave-cooperative decoder (pd_inflate_wave.h)
            out, ms, nb, n = capi.bgzf_inflate(data, variant=variant, reps=1)
            assert n == len(ref) and out == ref, (path, variant)
