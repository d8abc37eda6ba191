#!/bin/bash
# round 6, call 20: the configs[1] file: two copy lanes (-X h2d_lanes=2), inflate workgroups per launch (-X inflate_waves), the whole GPU suite with the round's pipeline
O=$GRAFT_REPO_ROOT/gpurun_out/r6c20; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
run() { # name tune
  local t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/b.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  echo "$1 [$2] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4; do
  run base_$rep x=1
  run lanes2_$rep h2d_lanes=2
  run lanes2t8_$rep h2d_lanes=2,dd_threads=8
  run w12_$rep inflate_waves=12
  run w10t8_$rep inflate_waves=10,dd_threads=8
  run w16_$rep inflate_waves=16
done
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
cat $O/summary.txt | cut -c1-330
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
