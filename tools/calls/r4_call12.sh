#!/bin/bash
# round 4, call 12: hardware queues — the runtime maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues
O=$GRAFT_REPO_ROOT/gpurun_out/r4c12; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
$P -i s.bam -o warm -t 16 > /dev/null 2>&1; sleep 1
for cfg in "4 6 32" "8 6 32" "16 6 32" "8 8 64" "16 12 32" "16 8 64" "4 6 32" "12 8 32" "8 6 32" "16 12 64"; do
  set -- $cfg
  TIMEFORMAT="wall %R s user %U sys %S"; ( time env GPU_MAX_HW_QUEUES=$1 PANDEPTH_TIMING=1 PANDEPTH_TUNE=dd_threads=$2,dd_batch_mb=$3 $P -i s.bam -o m -t 16 ) > $O/run.tmp 2>&1
  echo "hwq $1 feeders $2 batch $3 MB: $(grep -E 'decode \+ scatter' $O/run.tmp | sed 's/  */ /g') | $(grep wall $O/run.tmp) | $(grep -o 'device ms summed over batches: [^;]*' $O/run.tmp)" >> $O/matrix.log
  cmp -s m.chr.stat.gz warm.chr.stat.gz || echo "  OUTPUT DIFFERS" >> $O/matrix.log
  sleep 1
done
cat $O/matrix.log
rm -rf /tmp/e2e
