#!/bin/bash
# round 6, call 10: what batch-sized inflate launches on K streams reach together on the device alone (wave_debug STREAMS=K BATCH=n: no host, no copies, no
# other kernels) — the ceiling of the decode pipeline's inflate stage by launch size and streams in flight
O=$GRAFT_REPO_ROOT/gpurun_out/r6c10; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
W=$GRAFT_REPO_ROOT/tools/ubench/wd_cur
echo "== one launch" >> $O/streams.txt; CHECK=100 timeout 120 $W w.bam 5120 1000000 60 2>&1 | grep -v " % " >> $O/streams.txt
for q in 8 16; do
for k in 1 2 3 4 6 8 12; do
  echo "== hw queues $q" >> $O/streams.txt
  GPU_MAX_HW_QUEUES=$q STREAMS=$k BATCH=3200 CHECK=100 timeout 120 $W w.bam 5120 1000000 60 2>&1 | grep -v " % " | head -1 >> $O/streams.txt
done
done
for b in 800 1600 6400 12800; do for k in 2 6; do
  GPU_MAX_HW_QUEUES=8 STREAMS=$k BATCH=$b CHECK=100 timeout 120 $W w.bam 5120 1000000 60 2>&1 | grep -v " % " | head -1 >> $O/streams.txt
done; done
cat $O/streams.txt
