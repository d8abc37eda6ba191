#!/bin/bash
# round 4, call 3: the compact sample built by the decoder (two streams, runs placed by batch order): GPU suite with guards, the direct
# kernel's variants on the bench sample, the bench line (3e8-record e2e), and the kernel trace of the bench's device part
O=$GRAFT_REPO_ROOT/gpurun_out/r4c3; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log
VARIANTS=c803,c703,c704,c802,c604 EXPORT=0 timeout 300 python tools/ubench/direct_ab.py > $O/direct_ab.log 2>&1
( time timeout 900 python bench.py --e2e-records 3e8 --e2e-multi-records 5e7 ) > $O/bench.json 2> $O/bench.err
cd /tmp && PD_BENCH_CONFIG_LEGS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/bench.py --e2e-records 0 --e2e-multi-records 0 > $O/kt.log 2>&1
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/bench_kernel_stats.csv; rm -rf $O/kt
tail -3 $O/pytest_gpu.log; cat $O/direct_ab.log | tail -8; tail -c 1500 $O/bench.err
