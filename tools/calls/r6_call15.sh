#!/bin/bash
# round 6, call 15: pd_decode_collect waits for the batch's last EVENT instead of its stream (-X sync_event=1, the new default; 0 = as before) x 8 / 16 hardware
# queues x readers, copies first come first served, on the 3e8-record file; devtrace of the default
O=$GRAFT_REPO_ROOT/gpurun_out/r6c15; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
for cfg in "6 1 8" "6 1 16" "8 1 16" "6 2 16" "10 1 16" "4 2 8"; do
  set -- $cfg
  for se in 0 1; do
  t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1,sync_event=$se PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_e${se}_t$1_d$2_q$3_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "==== sync_event $se readers $1 x buffers $2 (hw queues $3), run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep 'decode + scatter' $O/trace_e${se}_t$1_d$2_q$3_$rep.log | tr -s ' ') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  python tools/feeder_trace.py $O/trace_e${se}_t$1_d$2_q$3_$rep.log | head -6 >> $O/summary.txt 2>&1
  sleep 1
  done
done
done
( cd /tmp && PANDEPTH_DEVTRACE=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/devtrace.log 2>&1 ); grep devtrace $O/devtrace.log | sed -n 100,112p | cut -c1-300
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
grep -E "====" $O/summary.txt | cut -c1-200
