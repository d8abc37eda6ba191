#!/bin/bash
# round 6, call 9: is it the token scratch?  More batches in flight make the decode SLOWER (call 8: six readers with two buffers each 0.93-1.08 s against 0.66-0.83 s
# with one) — every batch slot carries 5 120 token areas of 175 KB of which a member uses 24 KB: 900 MB per slot touched at a 175 KB stride.  `tok4k`: areas of
# 4 096 tokens (33 KB; enough for this file's members), everything else the same.  wave_debug isolated, then readers x buffers with both builds.
O=$GRAFT_REPO_ROOT/gpurun_out/r6c9; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
for B in r5 cur tok4k r5 cur tok4k; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
cat $O/ab.log
rm -f w.bam
cd $GRAFT_REPO_ROOT
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
run() { # name dir-of-the-executable tune [hw queues]
  local t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=${4:-8} PANDEPTH_TUNE=$3 PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  local en=$(grep 'main entered' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/'); local lv=$(grep 'main leaving' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/')
  echo "$1 [$3 q${4:-8}] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s: exec -> main $(awk "BEGIN{print $en-$t0}"), main $(awk "BEGIN{print $lv-$en}"), main's end -> reaped $(awk "BEGIN{print $t1-$lv}") | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(grep -o '[0-9]* units handed back' $O/cli_$1.log) | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
P=$GRAFT_REPO_ROOT/pandepth_amd; V=$GRAFT_REPO_ROOT/tools/ubench/var_tok4k
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
  for cfg in "6 1 8" "6 2 8" "8 1 8" "12 1 16" "4 3 8"; do
    set -- $cfg
    run cur_t$1_d$2_q$3_$rep $P dd_threads=$1,dd_depth=$2 $3
    run tok4k_t$1_d$2_q$3_$rep $V dd_threads=$1,dd_depth=$2 $3
  done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
cat $O/summary.txt | cut -c1-400
