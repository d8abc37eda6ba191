#!/bin/bash
# round 5, call 15: the inflate kernel with a 7-bit distance root (6 544 B of LDS per wave: 24 waves per CU instead of 22) against the 8-bit one, at 20 and 24 waves per CU
O=$GRAFT_REPO_ROOT/gpurun_out/r5c15; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; ALT=$GRAFT_REPO_ROOT/pandepth_amd/altd7/libpandepth_amd.so
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name preload tune
  ( cd /tmp && LD_PRELOAD=$2 PANDEPTH_TUNE="$3" PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'decode \+ scatter' $O/cli_$1.log) | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1) | handed back: $(grep -o '[0-9]* units handed back' $O/cli_$1.log | head -1)" >> $O/summary.txt
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run d8w20_$rep "" ""
  run d7w20_$rep $ALT ""
  run d7w24_$rep $ALT "inflate_waves=24"
  run d8w22_$rep "" "inflate_waves=22"
done
head -c 1000000000 /tmp/s.bam > /tmp/s1g.bam
for L in pandepth_amd/libpandepth_amd.so pandepth_amd/altd7/libpandepth_amd.so; do
  PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/$L WAVES=20,22,24 REPS=5 MAX_BYTES=9.9e8 timeout 300 python tools/ubench/inflate_ab.py /tmp/s1g.bam 2>&1 | grep -v amdgpu | sed "s|^|$L: |" >> $O/summary.txt
done
rm -f /tmp/s.bam* /tmp/s1g.bam /tmp/o_* /tmp/warm*
cat $O/summary.txt | cut -c1-260
