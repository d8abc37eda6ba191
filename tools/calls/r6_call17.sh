#!/bin/bash
# round 6, call 17: one host-to-device copy per batch (tables behind the members) on the context's main stream, first come first served; the decode GPU tests;
# readers x hardware queues on the 3e8-record file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c17; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py tests/test_host_generated.py tests/test_long_reads.py tests/test_cli_gpu.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
for cfg in "6 1 8" "6 1 16" "8 1 8" "8 1 16" "10 1 16" "5 1 8"; do
  set -- $cfg
  t0=$(date +%s.%N)
  ( cd /tmp && GPU_MAX_HW_QUEUES=$3 PANDEPTH_TUNE=dd_threads=$1,dd_depth=$2,dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_t$1_d$2_q$3_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "==== readers $1 x buffers $2 (hw queues $3), run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep 'decode + scatter' $O/trace_t$1_d$2_q$3_$rep.log | tr -s ' ') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  python tools/feeder_trace.py $O/trace_t$1_d$2_q$3_$rep.log | head -6 >> $O/summary.txt 2>&1
  sleep 1
done
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam*
grep -E "====" $O/summary.txt | cut -c1-200
