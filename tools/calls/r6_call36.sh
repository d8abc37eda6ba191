#!/bin/bash
# round 6, call 37: two / three / four literals per trip of phase 2 (their bytes in one store) against none
O=$GRAFT_REPO_ROOT/gpurun_out/r6c37; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
$GRAFT_REPO_ROOT/tools/bamgen -o q.bam -n 5000000 -Q 40 -t 32 2>> $O/gen.log
for rep in 1 2 3; do for B in lit0 p2 p3 p4; do
  echo "== $B: $(CHECK=2000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v ' % ' | tail -1 | cut -c1-100) | q40: $(CHECK=2000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B q.bam 5120 1000000 60 2>&1 | grep -v ' % ' | tail -1 | cut -c1-100)" >> $O/ab.log
done; done
cat $O/ab.log; rm -f w.bam q.bam
