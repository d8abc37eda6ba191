#!/bin/bash
# round 6, call 28: the readers' buffers page-locked up front (six, 35-49 ms before the first read) against one up front and the rest by their readers, one at a time,
# beside the first batches (-X dd_pin_ahead=1), 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c28; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5; do for pa in 6 1 0; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=dd_trace=1,dd_pin_ahead=$pa PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_p${pa}_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "pin ahead $pa run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter|work list' $O/trace_p${pa}_$rep.log | tr -s ' ' | tr '\n' ';') first batches collected at $(grep '\[trace\] batch [0-5] ' $O/trace_p${pa}_$rep.log | awk '{printf "%.0f ", $15/1000}') ms; $(python tools/feeder_trace.py $O/trace_p${pa}_$rep.log | head -1 | cut -c1-90)" >> $O/summary.txt
  sleep 2
done; done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt | cut -c1-400
