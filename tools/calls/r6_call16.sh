#!/bin/bash
# round 6, call 16: where the first batches' 60-140 ms go (PANDEPTH_DEVTRACE: the parts of pd_decode_queue), 8 and 16 hardware queues
O=$GRAFT_REPO_ROOT/gpurun_out/r6c16; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 50000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for q in 8 16 8 16; do
( cd /tmp && GPU_MAX_HW_QUEUES=$q PANDEPTH_DEVTRACE=1 PANDEPTH_TUNE=dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/devtrace_q$q.log 2>&1 )
echo "== hw queues $q: $(grep 'decode + scatter' $O/devtrace_q$q.log)"; grep "pd_decode_queue:" $O/devtrace_q$q.log | head -8 | cut -c1-260; grep "\[trace\] batch [0-7] " $O/devtrace_q$q.log | cut -c1-200
grep "decode entry\|pd_create" $O/devtrace_q$q.log | cut -c1-330
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
