#!/bin/bash
# round 4, call 33: kernel trace and PMC passes of the bench with the joined-tail direct kernel as the default
O=$GRAFT_REPO_ROOT/gpurun_out/r4c33; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/bench.py --e2e-records 0 --e2e-multi-records 0 > $O/kt_bench.json 2> $O/kt.log ); ls $O/kt | head -3
timeout 110 bash tools/pmc_collect.sh > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-200
