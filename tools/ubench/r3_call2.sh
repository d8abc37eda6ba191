# round 3, call 2: compact sample tests + A/B of the C8 variants + loopback overflow test
O=$GRAFT_REPO_ROOT/gpurun_out/r3c2; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "compact or direct" 2>&1 | tail -25 ) > $O/tests_compact.log 2>&1
( timeout 600 python -m pytest tests/test_comm_loopback_gpu.py -x -q -m gpu -k "overflow" 2>&1 | tail -8 ) > $O/tests_overflow.log 2>&1
export VARIANTS=0,c0,c502,c504,c508,c602,c604,c702,c704,c802,c804,c801,c0@32768,c0@131072,c802@131072
( timeout 900 python tools/ubench/direct_ab.py > $O/ab.log 2>&1 )
