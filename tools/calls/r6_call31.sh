#!/bin/bash
# round 6, call 32: the slots' page-locked buffers as huge pages registered with the runtime (pin_alloc) instead of hipHostMalloc: pd_decode_begin, first batches, the phase; GPU tests
# make): engine create / pd_decode_begin / first batches on the 3e8-record file; the CLI GPU tests
O=$GRAFT_REPO_ROOT/gpurun_out/r6c31; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_gpu_bgzf.py tests/test_gpu_engine.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 2
for rep in 1 2 3 4 5 6 7 8; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  lv=$(grep 'main leaving' $O/trace_$rep.log | sed 's/.* at \([0-9.]*\) .*/\1/')
  echo "run $rep: wall $(awk "BEGIN{print $t1-$t0}") exit $(awk "BEGIN{print $t1-$lv}") $(grep -E 'decode \+ scatter|engine create|pd_decode_begin|pd_create: streams|pd_create: other' $O/trace_$rep.log | tr -s ' ' | tr '\n' ';') first batch collected at $(grep '\[trace\] batch 0 ' $O/trace_$rep.log | awk '{printf "%.0f ", $15/1000}') ms" >> $O/summary.txt
  sleep 2
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
cat $O/summary.txt | cut -c1-560
