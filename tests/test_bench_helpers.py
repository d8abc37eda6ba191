"""bench.py's host-side helpers that the driver's run depends on (no GPU): the BAM header reader behind the e2e annotation leg."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tools import synth  # noqa: E402


def test_bam_contigs_reads_a_header_that_spans_members(tmp_path):
    # 4 000 contigs with long names: a header of about 140 KB, three BGZF members
    names = ["scaffold_with_a_long_name_%06d" % i for i in range(4000)]
    lens = np.arange(1000, 5000, dtype=np.int64)
    rec = synth.gen_records_numpy(lens[:50], 200, seed=1)
    bam = str(tmp_path / "h.bam")
    synth.write_bam(bam, names, lens, rec, procs=1, payload=False)
    got_names, got_lens = bench._bam_contigs(bam)
    assert got_names == names
    assert got_lens == [int(x) for x in lens]
