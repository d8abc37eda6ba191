"""Differential fuzzing of the host pipeline (CPU oracle engine) against the compiled reference, when it is present
(dev container: oracle/_ref/pandepth_ref built from /root/reference): tests/fuzz_vs_ref.py generates small SAM / BAM /
BAM+BAI / CRAM 3.0 and 3.1 (+CRAI) / PAF / #.list inputs, GFF / GTF / BED3 / BED4 files with the quirks real files have (comments, blank lines, unknown
contigs, start > end, duplicate ids, odd attribute orders, leading zeros, spaces for tabs) and random option mixes, and
compares exit code, stdout and every output file byte for byte.  This is how the last-base target rule of the reference's
indexed path was found (tests/golden/f4).  Skipped where the reference binary does not exist (GPU box, CI)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref", "pandepth_ref")


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
@pytest.mark.parametrize("seed", [101, 102])
def test_random_inputs_match_the_reference(seed):
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), str(seed), "150"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
def test_random_command_lines_match_the_reference():
    """option parser and early error paths: flags in any order, missing values, unknown flags, unreadable target files
    (which the reference treats as empty lists), -c without -r"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "7", "200", "args"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
def test_large_contigs_around_the_reference_window_steps():
    """contigs of 10-32 Mb with reads and targets clustered around the 10 Mb steps of the reference's indexed window walk"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "41", "3", "big"], capture_output=True, text=True,
                       timeout=1800)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
def test_messy_target_files_match_the_reference():
    """Windows line endings, extra columns, runs of spaces, truncated BED lines (the reference re-uses the previous line's values)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "51", "150", "messy"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
def test_oracle_restatement_matches_the_reference_on_random_inputs():
    """oracle/pd_oracle.py itself (the checker of the GPU tests) against the reference binary: table and per-site texts"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "61", "80", "oracle"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.access(REF, os.X_OK), reason="needs the compiled reference (dev container only)")
def test_paf_inputs_match_the_reference():
    """PAF front end (paf_main): targets from the first file or from -r, cg:Z: walks, reversed coordinates, tp:A:S / -x,
    -q on column 12, lists, gzip, GFF / BED targets, GC columns"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "pandepth_amd"), "libpandepth_host.a"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(HERE, "harness"), "pandepth_oracle_cli"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "71", "150", "paf"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_vs_ref.py"), "72", "60", "oracle-paf"], capture_output=True, text=True,
                       timeout=1200)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
