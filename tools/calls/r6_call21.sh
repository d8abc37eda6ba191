#!/bin/bash
# round 6, call 21: `-w 100 -a` on 2e8 records (11.9 GB of per-site text) with the round's pipeline: 8 against 16 hardware queues, the writer's timing lines
O=$GRAFT_REPO_ROOT/gpurun_out/r6c21; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/m.bam -n 200000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/m.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do for q in 8 16; do
  t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$q PANDEPTH_TIMING=1 timeout 600 $P/pandepth -i /tmp/m.bam -w 100 -a -o /tmp/o_s -t 16 > $O/site_q${q}_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "q$q run $rep: wall $(awk "BEGIN{print $t1-$t0}") | $(grep -E 'decode \+ scatter|per-site file|per-site writer' $O/site_q${q}_$rep.log | tr -s ' ' | tr '\n' ' ' | cut -c1-600) | $(md5sum < /tmp/o_s.SiteDepth.gz | cut -c1-8)" >> $O/summary.txt
  sleep 1
done; done
rm -f /tmp/o_* /tmp/warm* /tmp/m.bam*
cat $O/summary.txt; grep "timing" $O/site_q16_1.log | cut -c1-300 | tail -25
