#!/bin/bash
# round 5, call 30: the round's final build — smoke(), and the executable's timing lines on a 1e8-record file (three runs)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c30; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
tools/bamgen -o /tmp/s.bam -n 100000000 -t 32 2> $O/gen.log
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1
for k in 1 2 3; do ( cd /tmp && PANDEPTH_TIMING=1 timeout 120 $CLI -i /tmp/s.bam -o /tmp/o_$k -t 16 > $O/cli_$k.log 2>&1 ); cmp /tmp/o_$k.chr.stat.gz /tmp/warm.chr.stat.gz; grep -E "decode \+ scatter|device decode" $O/cli_$k.log | cut -c1-420; done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
