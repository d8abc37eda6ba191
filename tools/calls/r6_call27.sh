#!/bin/bash
# round 6, call 27: where pd_decode_begin's occasional 1.2 s go (its parts against the clock), 10 runs on the configs[1] file with 1 s and 3 s between them
O=$GRAFT_REPO_ROOT/gpurun_out/r6c27; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for pause in 1 1 1 1 1 1 3 3 3 3 3 3 0.3 0.3 0.3 0.3; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/b.bam -o /tmp/o_t -t 16 > $O/cli.log 2>&1 )
  t1=$(date +%s.%N)
  echo "pause $pause before the next: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter|engine create|pd_decode_begin|pd_create: cell|pd_create: other' $O/cli.log | tr -s ' ' | tr '\n' ';') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep $pause
done
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
cat $O/summary.txt | cut -c1-640
