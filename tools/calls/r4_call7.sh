#!/bin/bash
# round 4, call 7: timeline of the executable's device decode (kernel + copy trace with timestamps) on a 1e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r4c7; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 100000000 -t 32 2> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
PANDEPTH_TIMING=1 $P -i s.bam -o warm -t 16 > /dev/null 2> $O/warm.log; sleep 1
cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o cli -- $P -i /tmp/e2e/s.bam -o /tmp/e2e/m -t 16 > $O/prof.log 2>&1
find $O/tr -name "*.csv" | while read f; do cp "$f" $O/$(basename "$f"); done; rm -rf $O/tr
ls -la $O; grep -E "decode \+ scatter|engine create" $O/warm.log $O/prof.log
rm -rf /tmp/e2e
