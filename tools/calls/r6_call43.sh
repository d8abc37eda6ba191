#!/bin/bash
# round 6, call 43: the new GPU test of the decode's transport alternatives (tests/test_host_generated.py::test_decode_transport_alternatives_gpu) and its neighbours
O=$GRAFT_REPO_ROOT/gpurun_out/r6c43; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_host_generated.py tests/test_pgzip.py -m gpu -q --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
