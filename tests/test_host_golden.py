"""The product's HOST code (BGZF/BAM/SAM/BAI readers, read selection, region model, option
parser, table writer — pandepth_amd/host/) driven end to end on a CPU engine built from the
oracle's loops (tests/harness/oracle_engine.cpp) and compared BYTE FOR BYTE — gz bytes, text,
stdout, exit code — with what the reference binary produced for the same command line."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MANIFEST = json.load(open(os.path.join(HERE, "golden", "manifest.json")))


@pytest.fixture(scope="session")
def harness():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True,
                   stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness")], check=True, stdout=subprocess.DEVNULL)
    return os.path.join(HERE, "harness", "pandepth_oracle_cli")


def run_case(cli, case, tmp_path, threads, env=None):
    d = os.path.join(HERE, "golden", case["fixture"])
    args = [cli] + case["args"] + ["-o", str(tmp_path / "o")]
    if "-t" not in case["args"]:
        args += ["-t", str(threads)]
    p = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gzip.decompress(gz)).hexdigest() == meta["text_sha256"], suffix
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix + " (gz bytes)"


@pytest.mark.parametrize("case", MANIFEST, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_host_pipeline_byte_identical(harness, case, tmp_path):
    run_case(harness, case, tmp_path, 1)


@pytest.mark.parametrize("case", [e for e in MANIFEST if e["fixture"] in ("f3", "f6", "f7") or e["name"] in ("gff_a", "list3", "w100")],
                         ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_host_pipeline_parallel_readers(harness, case, tmp_path):
    run_case(harness, case, tmp_path, 4)


@pytest.mark.parametrize("gpus", ["2", "3"])
@pytest.mark.parametrize("case", [e for e in MANIFEST if ".list" in e["args"][1]], ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_list_mode_sharded_over_several_contexts(harness, case, gpus, tmp_path):
    """`#.list` with one engine context per (stand-in) GPU: files round-robin over the contexts,
    contexts summed before the statistics; output and stdout exactly as with one context."""
    run_case(harness, case, tmp_path, 4, env={"PANDEPTH_FAKE_GPUS": gpus})


def test_cli_messages(harness, tmp_path):
    p = subprocess.run([harness], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout.decode().startswith("Usage: pandepth -i in.bam")
    p = subprocess.run([harness, "-i", "x.bam"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0 and "Error: lack argument -i or -o" in p.stderr.decode()
    p = subprocess.run([harness, "-i", "x.bam", "-o"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert "Error: Lack argument for [ -o ]" in p.stderr.decode()
    p = subprocess.run([harness, "-z"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert "Error UnKnow argument -z" in p.stderr.decode()
    p = subprocess.run([harness, "-i", str(tmp_path / "missing.bam"), "-o", str(tmp_path / "o")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and "Error: Failed to open the BAM/CRAM file" in p.stderr.decode()
