#!/bin/bash
# round 6, call 18: k_inflate_wave without scratch (lane-only values recomputed where they are used instead of kept from the kernel's first instruction:
# no per-queue scratch set-up, 7 ms per hardware queue at the first launch), isolated and in the executable: the first batches' times, readers x buffers
O=$GRAFT_REPO_ROOT/gpurun_out/r6c18; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py -m gpu -q --timeout 600 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
for B in r5 cur r5 cur; do
  echo "== $B" >> $O/ab.log
  CHECK=4000 timeout 120 $GRAFT_REPO_ROOT/tools/ubench/wd_$B w.bam 5120 1000000 60 2>&1 | grep -v " % " >> $O/ab.log
done
cat $O/ab.log; rm -f w.bam; cd $GRAFT_REPO_ROOT
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
for cfg in "6 1 8" "7 1 8" "5 1 8" "4 2 8" "3 2 8" "6 1 16"; do
  set -- $cfg
  t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_t$1_d$2_q$3_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "==== readers $1 x buffers $2 (hw queues $3), run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep 'decode + scatter' $O/trace_t$1_d$2_q$3_$rep.log | tr -s ' ') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8) first batch collected at $(grep '\[trace\] batch 0 ' $O/trace_t$1_d$2_q$3_$rep.log | awk '{print $15/1000}') ms" >> $O/summary.txt
  python tools/feeder_trace.py $O/trace_t$1_d$2_q$3_$rep.log | head -6 >> $O/summary.txt 2>&1
  sleep 1
done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
grep -E "====" $O/summary.txt | cut -c1-240
