#!/bin/bash
# round 6, call 8: the balanced phase 2 without the occupancy it cost in call 7 (a checkpoint's counts packed into one word: three checkpoints in the old
# 1 600 bytes, four in 7 504 B of LDS) — wave_debug A/B, then the executable; page-locked buffers by how they are made (tools/ubench/pin_cost);
# readers x buffers x hardware queues with this round's kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r6c8; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py -m gpu -q --timeout 600 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
for m in 0 1 2 3 4 0 1 3; do
  t0=$(date +%s.%N); tools/ubench/pin_cost $m 6 32 2>> $O/pin_cost.txt; t1=$(date +%s.%N)
  echo "   ^ mode $m: process wall $(awk "BEGIN{print $t1-$t0}") s" >> $O/pin_cost.txt
done
cat /sys/kernel/mm/transparent_hugepage/enabled >> $O/pin_cost.txt
cat $O/pin_cost.txt
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
rm -f $O/ab.log
for B in r5 b3h16 b4h16 b3h8 b4h8 nb4 r5 b3h16 b4h16 b3h16_ticks b4h16_ticks; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " 0.0 %" >> $O/ab.log
done
cat $O/ab.log
rm -f w.bam
cd $GRAFT_REPO_ROOT
GEN=tools/bamgen
$GEN -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
run() { # name dir-of-the-executable tune [hw queues]
  local t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=${4:-8} PANDEPTH_TUNE=$3 PANDEPTH_TIMING=1 timeout 300 $2/pandepth -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s.%N)
  local en=$(grep 'main entered' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/'); local lv=$(grep 'main leaving' $O/cli_$1.log | sed 's/.* at \([0-9.]*\) .*/\1/')
  echo "$1 [$3 q${4:-8}] rc $rc wall $(awk "BEGIN{print $t1-$t0}") s: exec -> main $(awk "BEGIN{print $en-$t0}"), main $(awk "BEGIN{print $lv-$en}"), main's end -> reaped $(awk "BEGIN{print $t1-$lv}") | $(grep -E 'decode \+ scatter|engine create' $O/cli_$1.log | tr -s ' ' | tr '\n' ';') | $(grep -E 'summed over' $O/cli_$1.log | sed 's/.*device ms summed over batches: \([^;]*\);.*/\1/') | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4; do
  run r5_$rep $GRAFT_REPO_ROOT/tools/ubench/var_r5 x=1
  run b4h16_$rep $P x=1
  run b3h16_$rep $GRAFT_REPO_ROOT/tools/ubench/var_b3h16 x=1
done
for rep in 1 2 3; do
  for cfg in "6 1 8" "6 2 8" "8 1 8" "8 1 16" "12 1 16" "6 2 16" "4 3 8" "10 1 16"; do
    set -- $cfg
    run m_t$1_d$2_q$3_$rep $P dd_threads=$1,dd_depth=$2 $3
  done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
cat $O/summary.txt | cut -c1-420
