#!/bin/bash
# round 6, call 49: the same with the readers reading INTO page-locked buffers (PINNED_READERS=1), as the decode's readers do
O=$GRAFT_REPO_ROOT/gpurun_out/r6c49; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 100000000 -t 32 2>> $O/gen.log
cat /tmp/s.bam > /dev/null
PINNED_READERS=1 timeout 120 tools/ubench/h2d_numa /tmp/s.bam 5 > $O/h2d_pinned_readers.txt 2>&1; cat $O/h2d_pinned_readers.txt
rm -f /tmp/s.bam*
