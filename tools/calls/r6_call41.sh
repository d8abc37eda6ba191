#!/bin/bash
# round 6, call 41: with the faster inflate kernel the copy engine is the busier of the two: two copy lanes (-X h2d_lanes=2), seven / eight readers again, 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c41; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5; do for tn in x=1 h2d_lanes=2 dd_threads=8 h2d_lanes=2,dd_threads=8; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=$tn,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/cli.log 2>&1 )
  t1=$(date +%s.%N)
  echo "$tn run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter' $O/cli.log | tr -s ' ') | $(grep -E 'summed over' $O/cli.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(python tools/feeder_trace.py $O/cli.log | sed -n 1p | cut -c40-90)" >> $O/summary.txt
  sleep 2
done; done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm* $O/cli.log
cat $O/summary.txt
