#!/bin/bash
# round 6, call 25: the configs[1] file with the .bai mapped and its bins left alone (whole-genome modes only use the linear index): where the time outside the
# feeders' window goes (index, work list, pd_decode_begin), 6 runs
O=$GRAFT_REPO_ROOT/gpurun_out/r6c25; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd; V=$GRAFT_REPO_ROOT/tools/ubench/var_r5
run() { # name dir tune
  local t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=$3 PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i /tmp/b.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  local en=$(grep 'main entered' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/'); local lv=$(grep 'main leaving' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/')
  echo "$1 [$3] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s (exec -> main $(awk "BEGIN{print $en-$t0}"), main's end -> reaped $(awk "BEGIN{print $t1-$lv}")) | $(grep -E 'decode \+ scatter|engine create|index of|work list' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4 5 6; do
  run new_$rep $P x=1
  [ $rep -le 3 ] && run r5_$rep $V x=1
done
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
cat $O/summary.txt | cut -c1-520
