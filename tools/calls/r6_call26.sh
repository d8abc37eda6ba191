#!/bin/bash
# round 6, call 26: the decode slots made ready by a helper thread beside the caller (buffer, stream, hardware queue; a slot is handed out when it is ready), the room
# for the one-copy tables also in the buffers pinned up front: decode GPU tests, the 3e8-record file (first batches' times), the configs[1] file
O=$GRAFT_REPO_ROOT/gpurun_out/r6c26; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_gpu_bgzf.py tests/test_inflate_core.py tests/test_host_generated.py tests/test_long_reads.py tests/test_cli_gpu.py tests/test_comm_loopback_gpu.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
P=$GRAFT_REPO_ROOT/pandepth_amd
tools/bamgen -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.log
$P/pandepth -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TUNE=dd_trace=1 PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/s.bam -o /tmp/o_t -t 16 > $O/trace_3e8_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "3e8 run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter|pd_decode_begin' $O/trace_3e8_$rep.log | tr -s ' ' | tr '\n' ';') first batches collected at $(grep '\[trace\] batch [0-5] ' $O/trace_3e8_$rep.log | awk '{printf "%.0f ", $15/1000}') ms; $(python tools/feeder_trace.py $O/trace_3e8_$rep.log | head -1)" >> $O/summary.txt
  sleep 1
done
rm -f /tmp/s.bam* /tmp/o_* /tmp/warm*
( time tools/bamgen -o /tmp/b.bam -n 1000000000 -t 32 ) 2>> $O/gen.log
$P/pandepth -i /tmp/b.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3 4 5 6; do
  t0=$(date +%s.%N)
  ( cd /tmp && PANDEPTH_TIMING=1 timeout 300 $P/pandepth -i /tmp/b.bam -o /tmp/o_t -t 16 > $O/cli_1e9_$rep.log 2>&1 )
  t1=$(date +%s.%N)
  echo "1e9 run $rep: wall $(awk "BEGIN{print $t1-$t0}") $(grep -E 'decode \+ scatter|engine create|pd_decode_begin' $O/cli_1e9_$rep.log | tr -s ' ' | tr '\n' ';') $(zcat /tmp/o_t.chr.stat.gz | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
done
rm -f /tmp/o_* /tmp/warm* /tmp/b.bam*
cat $O/summary.txt | cut -c1-420
