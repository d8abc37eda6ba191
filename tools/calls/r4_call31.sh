#!/bin/bash
# round 4, call 31: k_direct_c8 with the sorted stream's remainder and the later runs as one sequence of chunks (JOIN) against the two-stream loop, on the bench sample
# (tables of every variant compared over six window / threshold / wrap cases), then the compact-sample tests
O=$GRAFT_REPO_ROOT/gpurun_out/r4c31; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
VARIANTS=c803,c0,c1703,c1704,c1802,c803,c0 EXPORT=0 timeout 240 python tools/ubench/direct_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
( timeout 200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 150 -k "compact or runs" ) 2>&1 | tail -2 | tee $O/pytest.txt
