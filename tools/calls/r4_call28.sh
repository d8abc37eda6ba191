#!/bin/bash
# round 4, call 28: kernel + memory-copy timeline of the executable's decode on a 1e8-record file (where does a batch's 2.6 ms "H2D" stage go?)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c28; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
$GEN -o /tmp/s.bam -n 100000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o cli -- $CLI -i /tmp/s.bam -o /tmp/o_s -t 16 > $O/cli.log 2>&1 )
rm -f /tmp/s.bam /tmp/o_* /tmp/warm*
ls -la $O/tr | head; grep -E "decode \+ scatter|device decode" $O/cli.log | cut -c1-300
