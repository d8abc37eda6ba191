# round 3, call 5: whole GPU suite, kernel trace of the bench command, PMC traffic, CLI trace at 3e8 records, the N > 1 step with a 1-rank group
O=$GRAFT_REPO_ROOT/gpurun_out/r3c5; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -25 ) > $O/tests_gpu.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --e2e-records 0 > $O/kt_bench.json 2> $O/kt_bench.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && (head -1 $f; grep -E "pdk::|anonymous" $f) > $O/bench_kernel_stats.csv; rm -rf $O/kt
cd $GRAFT_REPO_ROOT
( bash tools/pmc_collect.sh 1e9 > $O/pmc.log 2>&1 ); cp gpurun_out/pmc/FETCH_SIZE_pdk.csv $O/ 2>/dev/null; cp gpurun_out/pmc/WRITE_SIZE_pdk.csv $O/ 2>/dev/null
( PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_1rank.json 2> $O/bench_1rank.err )
mkdir -p /tmp/e2e && cd /tmp/e2e
( timeout 900 $GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 300000000 -t 32 ) > $O/gen.log 2>&1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
PANDEPTH_TIMING=1 $P -i b.bam -o m -t 16 > /dev/null 2> $O/cli_timing.log
cd /tmp && PANDEPTH_ORDERLY_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o cli -- $P -i /tmp/e2e/b.bam -o /tmp/e2e/dd -t 16 > $O/prof.log 2>&1
f=$(find $O/kt2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cli_kernel_stats.csv; rm -rf $O/kt2
rm -rf /tmp/e2e
