"""Pins the CPU oracle (oracle/pd_oracle.c + pd_oracle.py) against the compiled reference:
every case in tests/golden/manifest.json is replayed through the restatement and the text of
every output file must hash to what oracle/_ref/pandepth_ref wrote (see tests/golden/make_golden.py)."""
import hashlib
import json
import os

import pytest

import pd_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "manifest.json")))


@pytest.mark.parametrize("case", MANIFEST, ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))
def test_oracle_reproduces_reference_output(case):
    d = os.path.join(HERE, "golden", case["fixture"])
    got = O.run(case["args"], cwd=d)
    assert set(got) == set(case["outputs"])
    for suffix, meta in case["outputs"].items():
        assert hashlib.sha256(got[suffix].encode()).hexdigest() == meta["text_sha256"], suffix


def test_interval_form_equals_record_walk():
    """pdo_add_intervals (the unit the C-ABI carries) == pdo_walk_records (the reference's loop)."""
    import numpy as np
    r = O.read_alignments(os.path.join(HERE, "golden", "f3", "tiny.bam"))
    sel = [i for i in range(len(r.tid)) if r.tid[i] >= 0]
    tid, pos, flag, mapq, coff, cig = r.arrays(sel)
    off = O.contig_offsets(r.lens)
    d1 = np.zeros(int(off[-1]), dtype=np.uint32)
    O.lib().pdo_walk_records(len(tid), O._p(tid), O._p(pos), O._p(flag), O._p(mapq), O._p(coff), O._p(cig),
                             1796, -1, O._p(d1), O._p(off))
    runs = []
    for k in range(len(tid)):
        if flag[k] & 1796:
            continue
        cur = int(pos[k])
        for c in cig[coff[k]:coff[k + 1]]:
            op, ln = int(c) & 0xf, int(c) >> 4
            if op in (0, 7, 8):
                runs.append((tid[k], cur, cur + ln)); cur += ln
            elif op in (2, 3):
                cur += ln
    d2, _ = O.depth_from_intervals(r.lens, np.array(runs, dtype=np.int32))
    assert np.array_equal(d1, d2)
