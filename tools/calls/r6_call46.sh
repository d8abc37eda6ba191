#!/bin/bash
# round 6, call 46: a 32 MB host-to-device copy beside the decoder's kernels in the SAME process (wave_debug COPY=1 STREAMS=K REPEAT=n)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c46; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
for k in 1 3 6; do
  echo "== $k streams" >> $O/copy_beside.txt
  COPY=1 STREAMS=$k REPEAT=30 CHECK=10 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_cur w.bam 5120 1000000 100 2>&1 | grep -E "streams, launches|copies beside" >> $O/copy_beside.txt
done
cat $O/copy_beside.txt; rm -f w.bam
