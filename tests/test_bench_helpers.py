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


def test_parse_timing_reads_the_executables_diagnostics():
    """bench.py's e2e.phases_s / e2e.roofline come from the PANDEPTH_TIMING lines of the timed run (host/pipeline.cpp)."""
    import bench
    err = ("[timing] options + header                0.002 s   (total 0.002 s)\n"
           "[timing] engine create                   0.093 s   (total 0.101 s)\n"
           "[timing] device decode: 498 batches (index cuts), 6 feeders, 300000000 records on the device, 0 units handed back (0 records on the host); "
           "feeder thread-seconds: read+scan 1.20, submit 3.10; device ms summed over batches: H2D 400.5, inflate 3000.2, walk 90.1, emit 40.0; "
           "bytes: compressed 15900000000, inflated 99000000000\n"
           "[timing] decode + scatter                0.745 s   (total 0.846 s)\n")
    ph, dec = bench.parse_timing(err)
    assert ph["engine create"] == 0.093 and ph["decode + scatter"] == 0.745 and ph["total_in_main"] == 0.846
    assert dec["batches"] == 498 and dec["feeders"] == 6 and dec["records_on_device"] == 300000000
    assert dec["device_ms_summed"]["inflate"] == 3000.2 and dec["inflated_bytes"] == 99000000000 and dec["compressed_bytes"] == 15900000000


def test_spread_and_median_run():
    sp = bench._spread([3.0, 1.0, 2.0, 10.0])
    assert sp == {"median": 2.5, "min": 1.0, "max": 10.0, "mean": 4.0, "runs": 4}
    assert bench._spread([5.0])["median"] == 5.0
    walls = [2.9, 2.7, 3.4, 2.8, 3.0]
    assert walls[bench._median_run(walls)] == 2.9 == bench._spread(walls)["median"]
