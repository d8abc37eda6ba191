#!/bin/bash
# round 6, call 44: two against four literals per trip of phase 2 IN THE PIPELINE (the payload file is copy-bound now, 40-level qualities inflate-bound): 3e8 records and
# 1e8 records with 40-level qualities, runs interleaved
O=$GRAFT_REPO_ROOT/gpurun_out/r6c44; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/pandepth_amd; V=$GRAFT_REPO_ROOT/tools/ubench/var_lits4
run() { # name dir file
  local t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i $3 -o /tmp/o_t -t 16 > $O/cli.log 2>&1 )
  local t1=$(date +%s.%N)
  echo "$1: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter' $O/cli.log | tr -s ' ') | $(grep -E 'summed over' $O/cli.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 2
}
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5; do run "payload lits2 $rep" $P /tmp/s.bam; run "payload lits4 $rep" $V /tmp/s.bam; done
rm -f /tmp/s.bam*
tools/bamgen -o /tmp/q.bam -n 100000000 -Q 40 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/q.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5; do run "q40 lits2 $rep" $P /tmp/q.bam; run "q40 lits4 $rep" $V /tmp/q.bam; done
rm -f /tmp/q.bam* /tmp/o_* /tmp/warm* $O/cli.log
cat $O/summary.txt
