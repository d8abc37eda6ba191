#!/bin/bash
# round 6, call 38: PD_P2_LITS 2 / 3 / 4 again, six rounds, interleaved (call 37's box scattered by 5 % between rounds)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c38; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
for rep in 1 2 3 4 5 6; do for B in p2 p3 p4 lit0; do
  echo "== $B: $(CHECK=500 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v ' % ' | tail -1 | cut -c40-80)" >> $O/ab.log
done; done
cat $O/ab.log; rm -f w.bam
