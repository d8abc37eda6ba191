# round 3, call 8: the API change (pd_keep_deferred, direct calls read the sample) on the device
O=$GRAFT_REPO_ROOT/gpurun_out/r3c8; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 ) > $O/tests_gpu.log 2>&1
( PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_1rank.json 2> $O/bench_1rank.err )
