#!/bin/bash
# round 5, call 2: (a) why the dense no-index file declines; (b) hardware queues x readers x depth on the 3e8-record file (twelve batch streams on
# eight hardware queues share queues: is that what makes deeper pipelines slower?); (c) kernel + copy timelines of two settings.
O=$GRAFT_REPO_ROOT/gpurun_out/r5c2; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen
timeout 600 python -m pytest "tests/test_cli_gpu.py::test_compact_session_over_many_batches" -x -q -m gpu > $O/pytest_dense.log 2>&1; echo "pytest rc $?" >> $O/pytest_dense.log
grep -E "DECLINED|passed|failed" $O/pytest_dense.log | cut -c1-400
[ -x $GEN ] || g++ -O2 -std=c++17 -pthread tools/bamgen.cpp -lz -ldl -o $GEN
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name hwq tune
  for rep in 1 2; do
    ( cd /tmp && GPU_MAX_HW_QUEUES=$2 PANDEPTH_TUNE="$3" PANDEPTH_TIMING=1 timeout 300 $CLI -i /tmp/s.bam -o /tmp/o_$1 -t 16 > $O/cli_$1_$rep.log 2>&1 ); echo "$1 rep $rep rc $? $(grep -E 'decode \+ scatter' $O/cli_$1_$rep.log)" >> $O/summary.txt
    sleep 1
  done
  cmp /tmp/o_$1.chr.stat.gz /tmp/warm.chr.stat.gz >> $O/summary.txt 2>&1 || echo "$1 DIFFERENT" >> $O/summary.txt
}
for q in 8 16 24; do
  run q${q}_6x1 $q "dd_threads=6,dd_depth=1"
  run q${q}_6x2 $q "dd_threads=6,dd_depth=2"
  run q${q}_4x3 $q "dd_threads=4,dd_depth=3"
  run q${q}_3x2 $q "dd_threads=3,dd_depth=2"
done
run q4_6x1 4 "dd_threads=6,dd_depth=1"
run q16_12x1 16 "dd_threads=12,dd_depth=1"
cat $O/summary.txt
for cfg in "q8_6x1 8 dd_threads=6,dd_depth=1" "q16_6x2 16 dd_threads=6,dd_depth=2"; do
  set -- $cfg
  ( cd /tmp && GPU_MAX_HW_QUEUES=$2 PANDEPTH_TUNE="$3" PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_$1 -o cli -- $CLI -i /tmp/s.bam -o /tmp/o_tr -t 16 > $O/tr_$1.log 2>&1 )
  grep -E "decode \+ scatter" $O/tr_$1.log
done
rm -f /tmp/s.bam /tmp/o_* /tmp/warm*
du -sh $O/tr_*; find $O -name '*.csv' | head
