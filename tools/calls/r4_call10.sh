#!/bin/bash
# round 4, call 10: is the isolated inflate rate a burst figure?  5 launches (0.1 s) against 60 and 200 launches back to back (1.2 s / 4 s)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c10; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o /tmp/ab.bam -n 12000000 -t 32 2> $O/gen.log
U=$GRAFT_REPO_ROOT/tools/ubench
for r in 5 60 200 5; do REPS=$r WAVES=20 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/sustained.log 2>&1; done
# a 32 MB batch's worth of members per launch (3100), as the decoder launches them
MAX_BYTES=33e6 REPS=200 WAVES=20 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/sustained.log 2>&1
MAX_BYTES=66e6 REPS=200 WAVES=20 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/sustained.log 2>&1
MAX_BYTES=130e6 REPS=100 WAVES=20 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/sustained.log 2>&1
cat $O/sustained.log
