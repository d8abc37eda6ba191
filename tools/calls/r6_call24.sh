#!/bin/bash
# round 6, call 24: more readers than batches on the device (-X dd_inflight=n caps the batches queued at a time; the other readers read meanwhile) on the configs[1] file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c24; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
run() { # name tune queues
  local t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=$2,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/b.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  echo "$1 [$2 q$3] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  python tools/feeder_trace.py $O/cli_$1.log | head -5 | cut -c1-330 >> $O/summary.txt
  grep -v "\[trace\]" $O/cli_$1.log > $O/cli_$1.txt; rm -f $O/cli_$1.log
  sleep 1
}
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
  run base_$rep x=1 8
  run t9i6_$rep dd_threads=9,dd_inflight=6 16
  run t8i5_$rep dd_threads=8,dd_inflight=5 16
  run t10i7_$rep dd_threads=10,dd_inflight=7 16
  run t9i6q8_$rep dd_threads=9,dd_inflight=6 8
  run t12i8_$rep dd_threads=12,dd_inflight=8 16
done
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
grep -E "rc 0|per batch$|ms per batch|read done|in flight" $O/summary.txt | cut -c1-300
