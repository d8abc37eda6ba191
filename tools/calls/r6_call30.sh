#!/bin/bash
# round 6, call 30: the round's evidence with the final pipeline: kernel + copy timeline of the decode (3e8-record file), kernel stats of the executable, kernel stats of
# the bench's device legs
O=$GRAFT_REPO_ROOT/gpurun_out/r6c30; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -o run -- $CLI -i /tmp/s.bam -o /tmp/o_tl -t 16 > $O/tl.log 2>&1 )
python tools/timeline.py $(dirname $(find $O/tl -name "*kernel_trace.csv" | head -1)) > $O/decode_timeline.txt 2>&1; rm -rf $O/tl
sleep 2
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli -o run -- $CLI -i /tmp/s.bam -o /tmp/o_s -t 16 > $O/cli_prof.log 2>&1 )
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/bench.py --e2e-records 0 --e2e-multi-records 0 > $O/kt_bench.json 2> $O/kt.log ); find $O/kt -name "*kernel_stats.csv" | head -2
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*memory_copy*" -delete
head -30 $O/decode_timeline.txt | cut -c1-200; grep "decode + scatter" $O/tl.log $O/cli_prof.log; du -sh $O
