#!/bin/bash
# round 6, call 35: two literals per trip of phases 1 and 2 (PD_LIT2) and the lane number opaque in every each() (no spills, no scratch again): wave_debug A/B with ticks,
# the inflate GPU tests, the executable on the 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c35; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py tests/test_long_reads.py -m gpu -q --timeout 600 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
for B in prev lit0 lit2 lit2no prev lit0 lit2 lit2_ticks; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
$GRAFT_REPO_ROOT/tools/bamgen -o q.bam -n 5000000 -Q 40 -t 32 2>> $O/gen.log
for B in prev lit2 prev lit2; do
  echo "== $B, 40-level qualities" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B q.bam 5120 1000000 60 2>&1 | grep -v " % " >> $O/ab.log
done
cat $O/ab.log; rm -f w.bam q.bam; cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5 6; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/cli_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter' $O/cli_$rep.log | tr -s ' ') | $(grep -E 'summed over' $O/cli_$rep.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 2
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
