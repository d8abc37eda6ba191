#!/bin/bash
# round 6, call 33: the per-site writer by the size of its rounds (PGZ_DEV_BATCH_MB: a parse call of 48 MB is 375 workgroups on 256 CUs — a wave and a half), `-w 100 -a`, 2e8 records
O=$GRAFT_REPO_ROOT/gpurun_out/r6c33; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/m.bam -n 200000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/m.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3; do for mb in 96 64 128 192 66 132; do
  t0=$(date +%s.%N)
  ( cd /tmp && PGZ_DEV_BATCH_MB=$mb PANDEPTH_TIMING=1 timeout 600 $P/pandepth -i /tmp/m.bam -w 100 -a -o /tmp/o_s -t 16 > $O/site_b${mb}_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "batch $mb MB run $rep: wall $(awk "BEGIN{print $t1-$t0}") | $(grep -E 'per-site file|per-site writer' $O/site_b${mb}_$rep.log | tr -s ' ' | tr '\n' ' ' | cut -c1-420) | $(md5sum < /tmp/o_s.SiteDepth.gz | cut -c1-8)" >> $O/summary.txt
  sleep 2
done; done
rm -f /tmp/o_* /tmp/warm* /tmp/m.bam*
cat $O/summary.txt | cut -c1-330
