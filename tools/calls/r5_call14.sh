#!/bin/bash
# round 5, call 14: kernel + copy timeline of the decode phase with the round's final code (4 KiB lane stretches, k_copy_words, six readers x one buffer) on the 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r5c14; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$rep -t 16 2>&1 | grep -E 'decode \+ scatter' >> $O/plain.txt ); sleep 1; done
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o cli -- $CLI -i /tmp/s.bam -o /tmp/o_tr -t 16 > $O/tr.log 2>&1 )
grep -E "decode \+ scatter" $O/tr.log; cat $O/plain.txt
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
