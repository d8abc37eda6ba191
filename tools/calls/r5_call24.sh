#!/bin/bash
# round 5, call 24: the new inflate kernel in the product: GPU tests of the BGZF path, then the CLI on the 3e8-record file with the library as built
# (new), with the inflate kernel of the commit before (old: alt_old/) and with registers for 5 waves per SIMD (w5: alt_w5/) preloaded; tables compared
O=$GRAFT_REPO_ROOT/gpurun_out/r5c24; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name preload
  ( cd /tmp && LD_PRELOAD=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'decode \+ scatter' $O/cli_$1.log) | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1)" >> $O/summary.txt
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run new_$rep ""
  run old_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt_old/libpandepth_amd.so
  run w5_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt_w5/libpandepth_amd.so
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
