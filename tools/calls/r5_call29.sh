#!/bin/bash
# round 5, call 29: the final inflate kernel (checkpoints of phase 1 in LDS, registers for 5 waves per SIMD) in the product: the BGZF GPU tests, the kernel alone
# (inflate_ab.py: old = d784431's kernel, new = the library as built), and the CLI on the 3e8-record file, new against old preloaded, tables compared
O=$GRAFT_REPO_ROOT/gpurun_out/r5c29; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py tests/test_host_generated.py -x -q -m gpu > $O/pytest.log 2>&1; tail -2 $O/pytest.log
tools/bamgen -o /tmp/w.bam -n 20000000 -t 32 2> $O/gen.log
for rep in 1 2; do
  PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/tools/ubench/libpd_inflate_old.so WAVES=20 REPS=5 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
  WAVES=20 REPS=5 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
done
PANDEPTH_AMD_LIB=$GRAFT_REPO_ROOT/tools/ubench/libpd_inflate_old.so WAVES=20 REPS=20 MAX_BYTES=32e6 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
WAVES=20 REPS=20 MAX_BYTES=32e6 timeout 200 python tools/ubench/inflate_ab.py /tmp/w.bam >> $O/ab.log 2>&1
rm -f /tmp/w.bam*; cat $O/ab.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name preload
  ( cd /tmp && LD_PRELOAD=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); echo "$1 rc $? $(grep -E 'decode \+ scatter' $O/cli_$1.log) | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1)" >> $O/summary.txt
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run new_$rep ""
  run old_$rep $GRAFT_REPO_ROOT/pandepth_amd/alt_old/libpandepth_amd.so
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt
