#!/bin/bash
# round 6, call 19: the configs[1] file (1e9 records, 52.8 GB) through the executable: the round's pipeline (copies first come first served on the main stream, one
# copy per batch, collect on the batch's event, 16 hardware queues, the inflate kernel without scratch and with phase 2 balanced) against the build of the round's
# start (tools/ubench/var_r5), batch sizes, readers
O=$GRAFT_REPO_ROOT/gpurun_out/r6c19; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
df -h /tmp | tail -1
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log; ls -la /tmp/b.bam*
P=$GRAFT_REPO_ROOT/pandepth_amd; V=$GRAFT_REPO_ROOT/tools/ubench/var_r5
run() { # name dir tune
  local t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=$3 PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i /tmp/b.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  echo "$1 [$3] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4 5; do
  run r5_$rep $V x=1
  run new_$rep $P x=1
done
for rep in 1 2 3; do
  run b16_$rep $P dd_batch_mb=16
  run b48_$rep $P dd_batch_mb=48
  run b64_$rep $P dd_batch_mb=64
  run t8_$rep $P dd_threads=8
  run t5_$rep $P dd_threads=5
  run t4d2_$rep $P dd_threads=4,dd_depth=2
  run b64t4d2_$rep $P dd_threads=4,dd_depth=2,dd_batch_mb=64
done
( cd /tmp && PANDEPTH_TUNE=dd_trace=1 PANDEPTH_TIMING=1 $P/pandepth -i /tmp/b.bam -o /tmp/o_tr -t 16 > $O/trace_1e9.log 2>&1 ); python tools/feeder_trace.py $O/trace_1e9.log | head -6 > $O/trace_1e9_summary.txt; grep -v "\[trace\]" $O/trace_1e9.log > $O/trace_1e9_timing.log; rm -f $O/trace_1e9.log
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
cat $O/summary.txt | cut -c1-330; cat $O/trace_1e9_summary.txt | cut -c1-300
