#!/bin/bash
# round 6, call 11: the feeders' trace (-X dd_trace=1: a line per batch, tools/feeder_trace.py) at readers x buffers 6x1, 6x2, 8x1, 12x1 on the 3e8-record file:
# where a batch's time goes between "read" and "collected", how many are in flight, and why more in flight does not mean more per second
O=$GRAFT_REPO_ROOT/gpurun_out/r6c11; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do
for cfg in "6 1 8" "6 2 8" "8 1 8" "12 1 16" "3 2 8"; do
  set -- $cfg
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_t$1_d$2_$rep.log 2>&1 )
  echo "==== readers $1 x buffers $2 (hw queues $3), run $rep: $(grep 'decode + scatter' $O/trace_t$1_d$2_$rep.log | tr -s ' ')" >> $O/summary.txt
  python tools/feeder_trace.py $O/trace_t$1_d$2_$rep.log >> $O/summary.txt 2>&1
  sleep 1
done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
cut -c1-900 $O/summary.txt
