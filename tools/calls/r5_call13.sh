#!/bin/bash
# round 5, call 13 (2 KiB, 4 KiB, 8 KiB against 1 KiB): the record walk's segment size — a lane owns 512 B / 1 KiB (default) / 2 KiB of a segment (one wave = 64 lanes): more, shorter waves or fewer, longer ones
# on the 3e8-record file; the same binary with the library preloaded; tables compared
O=$GRAFT_REPO_ROOT/gpurun_out/r5c13; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name preload
  ( cd /tmp && LD_PRELOAD=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'decode \+ scatter' $O/cli_$1.log) | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1)" >> $O/summary.txt
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run s1024_$rep ""
  run s4096_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt4096/libpandepth_amd.so
  run s2048_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt2048/libpandepth_amd.so
  run s8192_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt8192/libpandepth_amd.so
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
