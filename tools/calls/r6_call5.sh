#!/bin/bash
# round 6, call 5: the whole gpu suite with the round's code so far; long reads after the compact sample's arrays grow by projection (one growth instead of
# eight); kernel trace of the one-rank sliced sum (k_sweep_i4_fast, export) and the PMC passes with the receiving sweep on real tile sums
O=$GRAFT_REPO_ROOT/gpurun_out/r6c5; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
run() { # name tune input
  local t0=$(date +%s%N)
  ( cd /tmp && PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i $3 -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s%N)
  echo "$1 [$2] rc $rc wall $(( (t1 - t0) / 1000000 )) ms | $(grep -E 'decode \+ scatter' $O/cli_$1.log | tr -s ' ') | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1) | $(grep -o 'runs to their arrays [0-9.]*' $O/cli_$1.log | head -1) | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | tail -1 | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$GEN -o /tmp/l.bam -n 600000 --long -t 32 2>> $O/gen.txt
$CLI -i /tmp/l.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do run long_$rep "x=1" /tmp/l.bam; done
run long_host "device_decode=0" /tmp/l.bam
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli_long -o run -- $CLI -i /tmp/l.bam -o /tmp/o_lp -t 16 > $O/cli_long_prof.log 2>&1 )
( cd /tmp && t0=$(date +%s%N); $GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i /tmp/l.bam -o /tmp/ref_l -t 16 > /dev/null 2>&1; t1=$(date +%s%N); echo "long: reference -t 16 wall $(( (t1 - t0) / 1000000 )) ms" >> $O/summary.txt; cmp /tmp/ref_l.chr.stat.gz /tmp/o_long_1.chr.stat.gz && echo "long: same as the reference" >> $O/summary.txt )
rm -f /tmp/o_* /tmp/warm* /tmp/l.bam* /tmp/ref_l*
( cd /tmp && PD_BENCH_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dist -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --e2e-records 0 --e2e-multi-records 0 > $O/kt_dist_bench.json 2> $O/kt_dist.log ); find $O/kt_dist -name "*kernel_stats.csv" | head -2
timeout 600 bash tools/pmc_collect.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
mkdir -p $O/pmc && cp gpurun_out/pmc/*_pdk.csv $O/pmc/ 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*agent_info.csv" -delete
cat $O/summary.txt | cut -c1-400; du -sh $O
