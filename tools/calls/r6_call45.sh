#!/bin/bash
# round 6, call 45: does a host-to-device copy slow down beside the inflate kernels (another process keeps three streams of batch-sized launches busy) or beside reading threads?
# tools/ubench/h2d_fill alone, beside the kernels, beside the kernels + five readers
O=$GRAFT_REPO_ROOT/gpurun_out/r6c45; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
echo "== alone" > $O/h2d_beside.txt; timeout 120 $GRAFT_REPO_ROOT/tools/ubench/h2d_fill 0 20 2>&1 | grep -E "old|pread " >> $O/h2d_beside.txt
( STREAMS=3 REPEAT=400 CHECK=10 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_cur w.bam 5120 1000000 100 > $O/load.txt 2>&1 ) &
sleep 3
echo "== beside three streams of inflate launches" >> $O/h2d_beside.txt; timeout 60 $GRAFT_REPO_ROOT/tools/ubench/h2d_fill 0 20 2>&1 | grep -E "old|pread " >> $O/h2d_beside.txt
echo "== beside the launches and five reading threads" >> $O/h2d_beside.txt; timeout 60 $GRAFT_REPO_ROOT/tools/ubench/h2d_fill 5 20 2>&1 | grep -E "^5 other.*(old|pread )" >> $O/h2d_beside.txt
wait
cat $O/h2d_beside.txt; tail -2 $O/load.txt; rm -f w.bam
