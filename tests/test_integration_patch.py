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
