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


BAM_CASES = [e for e in MANIFEST if e["args"][1].endswith(".bam")]


@pytest.mark.parametrize("case", BAM_CASES, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_device_decode_is_the_default(case, tmp_path):
    """BAM input is decoded on the GPU (BGZF inflate, record boundaries, filter, CIGAR walk) in every mode that has
    a device form: whole-contig modes with or without an index, GFF / BED targets with an index.  The timing line says so."""
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, PANDEPTH_TIMING="1")
    args = [CLI] + case["args"] + ["-o", str(tmp_path / "o")] + ([] if "-t" in case["args"] else ["-t", "3"])
    p = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    no_index = "-s" in case["args"] or "noidx" in case["args"][1] or "unsorted" in case["args"][1]
    targets = "-g" in case["args"] or "-b" in case["args"]
    if case["outputs"] and not (no_index and targets):
        assert b"device decode:" in p.stderr, p.stderr.decode()[-800:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix


@pytest.mark.parametrize("case", BAM_CASES[::3], ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_host_decode_byte_identical(case, tmp_path):
    """-X device_decode=0 (PANDEPTH_TUNE): the host readers (libdeflate + pd_push_intervals) still give the same bytes."""
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, PANDEPTH_TUNE="device_decode=0", PANDEPTH_TIMING="1")
    args = [CLI] + case["args"] + ["-o", str(tmp_path / "o")] + ([] if "-t" in case["args"] else ["-t", "3"])
    p = subprocess.run(args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert b"device decode:" not in p.stderr
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix


@pytest.mark.parametrize("case", [e for e in MANIFEST if ".list" in e["args"][1]], ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_pandepth_cli_list_over_two_contexts(case, tmp_path):
    """gpus=2 (PANDEPTH_TUNE) on a one-GPU box: two contexts (sharing the GPU), pd_accumulate_from at the end."""
    d = os.path.join(HERE, "golden", case["fixture"])
    p = subprocess.run([CLI] + case["args"] + ["-o", str(tmp_path / "o"), "-t", "4"], cwd=d, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, PANDEPTH_TUNE="gpus=2"))
    assert p.returncode == case["returncode"], p.stderr.decode()[-500:]
    assert p.stdout.decode() == case["stdout"]
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix


def _run_cli(args, env_extra, cwd):
    p = subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, PANDEPTH_TIMING="1", **env_extra))
    assert p.returncode == 0, p.stderr.decode()[-800:]
    return p.stderr.decode()


@pytest.mark.parametrize("kind", ["payload", "dense"])
def test_compact_session_over_many_batches(kind, tmp_path):
    """Whole-chromosome mode on a generated sorted BAM cut into dozens of small batches (six feeders): every batch's pass 2 writes its
    runs straight to their final places in the compact sample, the places being handed out in batch order (pd_decode_cfg::n_batches).
    "dense" is a file with ~7 bytes of BGZF per record (no SEQ / QUAL): far more runs than the arena was sized for from the compressed
    bytes, so the arena grows while batches are in flight.  Same bytes as the host readers' run, and the timing line names the session."""
    import sys
    sys.path.insert(0, ROOT)
    bam = str(tmp_path / "s.bam")
    if kind == "payload":
        gen = os.path.join(ROOT, "tools", "bamgen")
        assert os.access(gen, os.X_OK), "tools/bamgen not built (__graft_entry__.build)"
        subprocess.run([gen, "-o", bam, "-n", "3000000", "-t", "16"], check=True, stderr=subprocess.PIPE, timeout=600)
    else:
        from tools import synth
        names, lens = synth.genome_c2(scale=0.004)
        rec = synth.gen_records_numpy(lens, 4000000, seed=5)
        synth.write_bam(bam, names, lens, rec, procs=8, payload=False)
    err_d = _run_cli(["-i", bam, "-o", str(tmp_path / "dev"), "-t", "8"], {"PANDEPTH_TUNE": "dd_batch_mb=" + ("1" if kind == "dense" else "4")}, str(tmp_path))
    assert "runs (compact session)" in err_d, err_d[-1500:]
    _run_cli(["-i", bam, "-o", str(tmp_path / "host"), "-t", "8"], {"PANDEPTH_TUNE": "device_decode=0"}, str(tmp_path))
    assert (tmp_path / "dev.chr.stat.gz").read_bytes() == (tmp_path / "host.chr.stat.gz").read_bytes()
    # the same file through a mode that needs the cells (no compact session): still the same table as the host readers give
    _run_cli(["-i", bam, "-w", "1000", "-o", str(tmp_path / "devw"), "-t", "8"], {"PANDEPTH_TUNE": "dd_batch_mb=4"}, str(tmp_path))
    _run_cli(["-i", bam, "-w", "1000", "-o", str(tmp_path / "hostw"), "-t", "8"], {"PANDEPTH_TUNE": "device_decode=0"}, str(tmp_path))
    assert (tmp_path / "devw.win.stat.gz").read_bytes() == (tmp_path / "hostw.win.stat.gz").read_bytes()
