#!/bin/bash
# round 6, call 14: PANDEPTH_DEVTRACE — the host-clock times at which a batch's stages were seen to end (collect waits for the stage events one by one), six readers,
# copies first come first served: where the 2.5 ms lie that a batch spends between "queued" and "collected" beyond its device stages
O=$GRAFT_REPO_ROOT/gpurun_out/r6c14; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/s.bam -n 100000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for cfg in "6 1" "8 1" "1 1" "2 1"; do set -- $cfg
( cd /tmp && PANDEPTH_DEVTRACE=1 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/devtrace_t$1_d$2.log 2>&1 )
echo "== readers $1 x $2: $(grep 'decode + scatter' $O/devtrace_t$1_d$2.log)"; grep devtrace $O/devtrace_t$1_d$2.log | sed -n 60,80p | cut -c1-260
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
