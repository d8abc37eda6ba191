#!/bin/bash
# round 6, call 13: the batches' host-to-device copies first come, first served on one copy stream (-X h2d_fifo=1, the new default) against each on its batch's
# stream (=0), readers x buffers 6x1, 6x2, 8x1, 4x2, 12x1 on the 3e8-record file, with the feeders' trace
O=$GRAFT_REPO_ROOT/gpurun_out/r6c13; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
for cfg in "6 1 8" "6 2 8" "8 1 8" "4 2 8" "12 1 16" "8 2 16"; do
  set -- $cfg
  for fifo in 0 1; do
  t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1,h2d_fifo=$fifo PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_f${fifo}_t$1_d$2_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "==== fifo $fifo readers $1 x buffers $2 (hw queues $3), run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep 'decode + scatter' $O/trace_f${fifo}_t$1_d$2_$rep.log | tr -s ' ') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  python tools/feeder_trace.py $O/trace_f${fifo}_t$1_d$2_$rep.log | head -6 >> $O/summary.txt 2>&1
  sleep 1
  done
done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
grep -E "====" $O/summary.txt | cut -c1-200
