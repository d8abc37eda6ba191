#!/usr/bin/env python3
"""Applies INTEGRATION.md's marked code blocks to a scratch copy of the reference's translation unit and compiles + links the result
against libpandepth_amd.so — the proof that the boundary document is code, not prose.  Nothing of the reference is kept in this
repository: the script names LINE RANGES of /root/reference/src/PanDepth.cpp (with a one-token anchor per range, so that a changed
reference fails loudly instead of being patched in the wrong place); the replacement text comes out of INTEGRATION.md.
Dev container only (needs the reference checkout, its headers and its libhts.a).  Usage: tools/integration_patch.py [--keep DIR]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
# marker kind -> a token that must occur on the first line of the range (replace) or on the line itself (insert after)
ANCHOR = {"include": "unordered_map", "create": "}", "increment": "sam_itr_next", "no_stat_in_worker": "StatChrDepthLowMEM", "statistics": "}", "site_rows": "int32_t j"}


def blocks(md):
    out = []
    for m in re.finditer(r"<!-- patch: (\w+); (insert after|replace) PD:(\d+)(?:-(\d+))? -->\n```cpp\n(.*?)```", md, re.S):
        kind, how, a, b, code = m.group(1), m.group(2), int(m.group(3)), m.group(4), m.group(5)
        out.append((kind, how, a, int(b) if b else a, code))
    return out


def patched_source():
    src = open(os.path.join(REF, "src", "PanDepth.cpp"), encoding="utf-8", errors="replace").read().split("\n")
    bl = blocks(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    assert {b[0] for b in bl} == set(ANCHOR), "INTEGRATION.md: patch blocks %s, expected %s" % (sorted(b[0] for b in bl), sorted(ANCHOR))
    for kind, how, a, b, code in sorted(bl, key=lambda x: -x[2]):          # bottom up: line numbers above stay valid
        assert ANCHOR[kind] in src[a - 1], "PD:%d does not look like the line block '%s' was written for: %r" % (a, kind, src[a - 1][:80])
        if how == "replace":
            src[a - 1:b] = code.rstrip("\n").split("\n")
        else:
            src[a:a] = code.rstrip("\n").split("\n")
    return "\n".join(src)


def build(workdir):
    path = os.path.join(workdir, "PanDepth_mi355x.cpp")
    open(path, "w").write(patched_source())
    exe = os.path.join(workdir, "pandepth_patched")
    cmd = ["g++", "--std=c++11", "-g", "-O1", path, "-I" + os.path.join(REF, "include"), "-I" + os.path.join(ROOT, "include"),
           "-L" + os.path.join(REF, "lib"), "-lhts", "-ldeflate", "-lz", "-pthread",
           "-L" + os.path.join(ROOT, "pandepth_amd"), "-lpandepth_amd", "-Wl,-rpath=" + os.path.join(ROOT, "pandepth_amd"),
           "-Wl,-rpath-link=/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return p.returncode, p.stdout.decode(errors="replace"), exe


if __name__ == "__main__":
    keep = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--keep" else None
    d = keep or tempfile.mkdtemp(prefix="pdpatch")
    os.makedirs(d, exist_ok=True)
    rc, out, exe = build(d)
    print(out[-3000:])
    print("patched translation unit: %s  ->  %s (rc %d)" % (os.path.join(d, "PanDepth_mi355x.cpp"), exe, rc))
    sys.exit(rc)
