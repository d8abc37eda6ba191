"""GPU end-to-end parity of the GC(%) column cases (fixture f5: `-c -r ref.fa`) and of PAF input (fixture f6) and CRAM input (fixture f7): the same
checks as tests/test_cli_gpu.py — plain (BAM input is decoded on the GPU), host decode, `#.list` over two contexts.
Kept in a file that sorts last: these cases were added after the last GPU run of their round."""
import os

import pytest

import test_cli_gpu as T

pytestmark = pytest.mark.gpu

F5 = [e for e in T.ALL_CASES if e["fixture"] in ("f5", "f6", "f7")]
ID = dict(ids=lambda e: "%s-%s" % (e["fixture"], e["name"]))


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("case", F5, **ID)
def test_gc_cases_byte_identical(case, threads, tmp_path):
    T.test_pandepth_cli_byte_identical(case, threads, tmp_path)


BAM = [e for e in F5 if e["args"][1].endswith(".bam")]


@pytest.mark.parametrize("case", BAM, **ID)
def test_gc_cases_device_decode_is_the_default(case, tmp_path):
    T.test_pandepth_cli_device_decode_is_the_default(case, tmp_path)


@pytest.mark.parametrize("case", BAM[::2], **ID)
def test_gc_cases_host_decode_byte_identical(case, tmp_path):
    T.test_pandepth_cli_host_decode_byte_identical(case, tmp_path)


@pytest.mark.parametrize("case", [e for e in F5 if ".list" in e["args"][1]], **ID)
def test_gc_cases_list_over_two_contexts(case, tmp_path):
    T.test_pandepth_cli_list_over_two_contexts(case, tmp_path)
