#!/bin/bash
# round 5, call 7: the round's evidence — kernel trace of the bench's device legs, PMC traffic passes, kernel traces of the executable (whole-chromosome run on 1e8 records
# and -w 100 -a on 2e7), the GPU suite with the final code
O=$GRAFT_REPO_ROOT/gpurun_out/r5c7; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/bench.py --e2e-records 0 --e2e-multi-records 0 > $O/kt_bench.json 2> $O/kt.log ); ls $O/kt/* | head -5
timeout 600 bash tools/pmc_collect.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
mkdir -p $O/pmc && cp gpurun_out/pmc/*_pdk.csv $O/pmc/ 2>/dev/null
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 100000000 -t 32 2> $O/gen.txt; $GEN -o /tmp/w.bam -n 20000000 -t 32 2>> $O/gen.txt
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli -o run -- $CLI -i /tmp/s.bam -o /tmp/o_s -t 16 > $O/cli.log 2>&1 )
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli4 -o run -- $CLI -i /tmp/w.bam -w 100 -a -o /tmp/o_w -t 16 > $O/cli4.log 2>&1 )
( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_s2 -t 16 > $O/cli_plain.log 2>&1 )
rm -f /tmp/s.bam* /tmp/w.bam* /tmp/o_*
find $O -name "*kernel_stats.csv" | head
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
du -sh $O
