#!/bin/bash
# round 4, call 11: larger batches (an inflate launch of 3 100 members runs at 136 GB/s alone, one of 12 600 at 203)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c11; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
$P -i s.bam -o warm -t 16 > /dev/null 2>&1; sleep 1
for cfg in "6 128 20" "8 128 20" "4 128 20" "6 192 20" "8 96 20" "6 96 20" "4 256 20" "8 64 20" "6 128 20"; do
  set -- $cfg
  TIMEFORMAT="wall %R s user %U sys %S"; ( time env PANDEPTH_TIMING=1 PANDEPTH_TUNE=dd_threads=$1,dd_batch_mb=$2,inflate_waves=$3 $P -i s.bam -o m -t 16 ) > $O/run.tmp 2>&1
  echo "feeders $1 batch $2 MB waves $3: $(grep -E 'decode \+ scatter' $O/run.tmp | sed 's/  */ /g') | $(grep wall $O/run.tmp) | $(grep -o 'device ms summed over batches: [^;]*' $O/run.tmp)" >> $O/matrix.log
  cmp -s m.chr.stat.gz warm.chr.stat.gz || echo "  OUTPUT DIFFERS" >> $O/matrix.log
  sleep 1
done
cat $O/matrix.log
rm -rf /tmp/e2e
