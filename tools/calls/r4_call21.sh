#!/bin/bash
# round 4, call 21: 23 instead of 20 one-wave inflate workgroups per CU (the kernel's LDS allows 23), isolated and in the executable (3e8-record file), with 64 MB batches / 8 feeders beside
O=$GRAFT_REPO_ROOT/gpurun_out/r4c21; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2> $O/gen.log
( cd $GRAFT_REPO_ROOT && MAX_BYTES=4e9 WAVES=20,23,22,20,23 REPS=5 timeout 200 python tools/ubench/inflate_ab.py /tmp/e2e/s.bam ) > $O/ab.txt 2>&1; cat $O/ab.txt
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
$P -i s.bam -o warm -t 16 > /dev/null 2>&1; sleep 1
for cfg in "20 6 32" "23 6 32" "20 8 64" "23 8 64" "20 6 32" "23 6 32" "23 8 64" "20 8 64"; do
  set -- $cfg
  TIMEFORMAT="wall %R s user %U sys %S"; ( time env PANDEPTH_TIMING=1 PANDEPTH_TUNE=inflate_waves=$1,dd_threads=$2,dd_batch_mb=$3 $P -i s.bam -o m -t 16 ) > $O/run.tmp 2>&1
  echo "waves $1 feeders $2 batch $3 MB: $(grep -E 'decode \+ scatter' $O/run.tmp | sed 's/  */ /g') | $(grep wall $O/run.tmp) | $(grep -o 'device ms summed over batches: [^;]*' $O/run.tmp)" >> $O/matrix.log
  cmp -s m.chr.stat.gz warm.chr.stat.gz || echo "  OUTPUT DIFFERS" >> $O/matrix.log
  sleep 1
done
cat $O/matrix.log
rm -rf /tmp/e2e
