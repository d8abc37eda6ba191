"""INTEGRATION.md's reference-side patch is code: tools/integration_patch.py cuts the marked blocks out of the document, applies them to
a scratch copy of the reference's translation unit (by line range — no reference text lives in this repository) and compiles and links
it against libpandepth_amd.so with the document's link line.  Dev container only: needs /root/reference (skipped on the GPU box)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

needs_ref = pytest.mark.skipif(not (os.path.exists("/root/reference/src/PanDepth.cpp") and os.path.exists("/root/reference/lib/libhts.a")),
                               reason="the reference checkout is not mounted here")


@needs_ref
def test_integration_patch_compiles_and_links(tmp_path):
    from tools import integration_patch as ip
    assert os.path.exists(os.path.join(ROOT, "pandepth_amd", "libpandepth_amd.so")), "build the library first (__graft_entry__.build)"
    rc, out, exe = ip.build(str(tmp_path))
    assert rc == 0, out[-3000:]
    # the patched program binds the engine's entry points, and only through the C-ABI
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], stdout=subprocess.PIPE, check=True).stdout.decode()
    for sym in ("pd_create", "pd_push_intervals", "pd_scan", "pd_reduce_intervals", "pd_strerror"):
        assert (" U " + sym) in und, sym
    # ... and still is the reference's program: its usage text comes out (no device is touched before an input is opened)
    p = subprocess.run([exe, "-h"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    assert b"-i" in p.stdout and b"-o" in p.stdout


@needs_ref
def test_patch_blocks_are_the_documents():
    """every marked block is applied exactly once, at a line that still looks like the one it was written for"""
    from tools import integration_patch as ip
    bl = ip.blocks(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    assert sorted(b[0] for b in bl) == sorted(ip.ANCHOR)
    src = ip.patched_source()
    assert src.count('#include "pandepth_amd.h"') == 1 and src.count("pd_reduce_intervals(ctx") == 1 and "pd_push_intervals(ctx, runs.data()" in src


# ---- the patched reference RUNS: built by __graft_entry__.build() in the dev container (oracle/_ref/pandepth_patched, untracked like pandepth_ref,
# travels to the GPU box), executed here on the MI355X on the golden fixtures' per-site cases — the reference's own main, option parser, region model,
# htslib record loop, table and per-site writers, with the depth arrays, the increments and the statistics behind the C-ABI (PD:434-461, 329-348, 4278-4281)
import hashlib
import json

PATCHED = os.path.join(ROOT, "oracle", "_ref", "pandepth_patched")
MANIFEST = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
# the indexed per-site path the six blocks cover (PD:4127-4284): indexed BAM + -a, not the mode-6 sweep (-w < 150), not -s / lists / SAM
PATCHED_CASES = [c for c in MANIFEST if (c["fixture"], c["name"]) in {("f1", "chr_a"), ("f1", "w200_a"), ("f1", "gff_a"), ("f1", "bed3_a"), ("f2", "chr_a"),
                                                                         ("f2", "bed4_a"), ("f4", "bed4_a")}]


@pytest.mark.gpu
@pytest.mark.parametrize("case", PATCHED_CASES, ids=lambda c: "%s-%s" % (c["fixture"], c["name"]))
def test_patched_reference_runs_on_the_engine(case, tmp_path):
    if not os.access(PATCHED, os.X_OK):
        if os.access(os.path.join(ROOT, "oracle", "_ref", "pandepth_ref"), os.X_OK):
            pytest.fail("oracle/_ref/pandepth_ref travelled here but pandepth_patched did not: __graft_entry__.build() makes both")
        pytest.skip("oracle/_ref/ was not built where this tree comes from (no reference checkout there)")
    d = os.path.join(HERE, "golden", case["fixture"])
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "pandepth_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([PATCHED] + case["args"] + ["-o", str(tmp_path / "o"), "-t", "4"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert p.returncode == case["returncode"], p.stderr.decode()[-800:]
    assert p.stdout.decode() == case["stdout"]
    assert len(case["outputs"]) == 2
    for suffix, meta in case["outputs"].items():
        gz = (tmp_path / ("o." + suffix)).read_bytes()
        assert hashlib.sha256(gz).hexdigest() == meta["gz_sha256"], suffix + " (gz bytes)"
