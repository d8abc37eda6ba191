#!/bin/bash
# round 4, call 9: the same matrix with the batches' small tables staged through pinned memory
O=$GRAFT_REPO_ROOT/gpurun_out/r4c9; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
$P -i s.bam -o warm -t 16 > /dev/null 2>&1; sleep 1
for cfg in "6 32 20" "8 32 20" "12 32 20" "12 32 10" "8 64 20" "6 32 20" "12 64 20" "10 32 20"; do
  set -- $cfg
  TIMEFORMAT="wall %R s user %U sys %S"; ( time env PANDEPTH_TIMING=1 PANDEPTH_TUNE=dd_threads=$1,dd_batch_mb=$2,inflate_waves=$3 $P -i s.bam -o m -t 16 ) > $O/run.tmp 2>&1
  echo "feeders $1 batch $2 MB waves $3: $(grep -E 'decode \+ scatter' $O/run.tmp | sed 's/  */ /g') | $(grep wall $O/run.tmp) | $(grep -o 'device ms summed over batches: [^;]*' $O/run.tmp)" >> $O/matrix.log
  cmp -s m.chr.stat.gz warm.chr.stat.gz || echo "  OUTPUT DIFFERS" >> $O/matrix.log
  sleep 1
done
cat $O/matrix.log
rm -rf /tmp/e2e
