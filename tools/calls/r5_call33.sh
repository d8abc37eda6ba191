#!/bin/bash
# round 5, call 33 (the last seconds of the budget): the BGZF / inflate GPU tests and smoke() on the round's FINAL build
O=$GRAFT_REPO_ROOT/gpurun_out/r5c33; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 30 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
