#!/bin/bash
# round 5, call 31: kernel trace of the executable with the round's final build (whole-chromosome run on a 1e8-record file)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c31; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
tools/bamgen -o /tmp/s.bam -n 100000000 -t 32 2> $O/gen.log
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli -o run -- $CLI -i /tmp/s.bam -o /tmp/o_s -t 16 > $O/cli.log 2>&1 )
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
find $O -name "*kernel_stats.csv" | head -2
head -8 $(find $O -name "*kernel_stats.csv" | head -1) | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
