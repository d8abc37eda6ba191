#!/bin/bash
# round 6, call 40: what PANDEPTH_TIMING's per-batch device events cost the run (the bench's timed runs set it for the phase lines): wall with and without, 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c40; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5 6 7 8; do
  t0=$(date +%s.%N); ( cd /tmp && PANDEPTH_TIMING=1 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > /dev/null 2>&1 ); t1=$(date +%s.%N); sleep 2
  t2=$(date +%s.%N); ( cd /tmp && $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > /dev/null 2>&1 ); t3=$(date +%s.%N); sleep 2
  echo "run $rep: with PANDEPTH_TIMING $(awk "BEGIN{print $t1-$t0}")  without $(awk "BEGIN{print $t3-$t2}")" >> $O/summary.txt
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
