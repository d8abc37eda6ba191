#!/bin/bash
# round 6, call 39: the product with two literals per trip (phases 1 and 2) and the opaque lane: decode GPU tests, ticks, the executable on the 3e8-record file and on 1e8 records
# with 40-level qualities
O=$GRAFT_REPO_ROOT/gpurun_out/r6c39; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py tests/test_long_reads.py tests/test_host_generated.py tests/test_cli_gpu.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
cd /tmp; $GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
CHECK=102037 timeout 300 $GRAFT_REPO_ROOT/tools/ubench/wd_p2_ticks w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" > $O/ticks.txt; cat $O/ticks.txt; rm -f w.bam; cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5 6; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/cli_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "3e8 run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter' $O/cli_$rep.log | tr -s ' ') | $(grep -E 'summed over' $O/cli_$rep.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 2
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
tools/bamgen -o /tmp/q.bam -n 100000000 -Q 40 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/q.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/q.bam -o /tmp/o_t -t 16 > $O/cliq_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "q40 1e8 run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter' $O/cliq_$rep.log | tr -s ' ') | $(grep -E 'summed over' $O/cliq_$rep.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/')" >> $O/summary.txt
  sleep 2
done
rm -f /tmp/q.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
