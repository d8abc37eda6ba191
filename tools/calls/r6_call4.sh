#!/bin/bash
# round 6, call 4: after the in-process transport stopped loading librccl and the wave-cooperative CIGAR walk went in — the comm / list / long-read / BGZF tests,
# the 3e8-record list over the transports (3 runs each), eight contexts on one GPU, long reads, the short-read walk / emit times before and after; then the
# round's evidence: kernel trace of the bench's device legs, PMC traffic passes, kernel traces + timeline of the executable
O=$GRAFT_REPO_ROOT/gpurun_out/r6c4; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_comm_loopback_gpu.py tests/test_comm_gpu.py tests/test_long_reads.py tests/test_gpu_bgzf.py tests/test_host_generated.py tests/test_cli_gpu.py tests/test_z_cli_gpu_late.py -m gpu -q --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; SHIM=$GRAFT_REPO_ROOT/tests/harness/libpd_loopback_nccl.so
run() { # name tune input
  local t0=$(date +%s%N)
  ( cd /tmp && PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i $3 -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s%N)
  echo "$1 [$2] rc $rc wall $(( (t1 - t0) / 1000000 )) ms | $(grep -E 'decode \+ scatter' $O/cli_$1.log | tr -s ' ') | $(grep -E 'comm (init|ahead)' $O/cli_$1.log | tr -s ' ' | cut -c1-200 | tr '\n' ';') | $(grep -o 'inflate [0-9.]*, walk [0-9.]*, emit [0-9.]*' $O/cli_$1.log | head -1) | $(grep -E 'summed over|added into' $O/cli_$1.log | cut -c10-100) | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | tail -1 | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$GEN -o /tmp/l.bam -n 600000 --long -t 32 2>> $O/gen.txt
$CLI -i /tmp/l.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do run long_$rep "x=1" /tmp/l.bam; done
grep -h "device decode\|decode entry" $O/cli_long_1.log | cut -c1-900 >> $O/summary.txt
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli_long -o run -- $CLI -i /tmp/l.bam -o /tmp/o_lp -t 16 > $O/cli_long_prof.log 2>&1 )
( cd /tmp && $GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i /tmp/l.bam -o /tmp/ref_l -t 16 > /dev/null 2>&1; cmp /tmp/ref_l.chr.stat.gz /tmp/o_long_1.chr.stat.gz && echo "long: same as the reference" >> $O/summary.txt )
rm -f /tmp/o_* /tmp/warm* /tmp/l.bam* /tmp/ref_l*
$GEN -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.txt
echo /tmp/s.bam > /tmp/s.list
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
  run single_$rep "x=1" /tmp/s.bam
  run list_nocomm_$rep "x=1" /tmp/s.list
  run list_peer_early_$rep "comm=force" /tmp/s.list
  run list_peer_inline_$rep "comm=force,comm_early=0" /tmp/s.list
done
run list_rccl_1 "comm=force,transport=rccl" /tmp/s.list
# the decode phase's timeline on this file (kernels + copies)
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl -o run -- $CLI -i /tmp/s.bam -o /tmp/o_tl -t 16 > $O/tl.log 2>&1 )
python tools/timeline.py $(dirname $(find $O/tl -name "*kernel_trace.csv" | head -1)) > $O/decode_timeline.txt 2>&1; rm -rf $O/tl
( cd /tmp && PANDEPTH_TIMING=1 PANDEPTH_ORDERLY_EXIT=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cli -o run -- $CLI -i /tmp/s.bam -o /tmp/o_s -t 16 > $O/cli_prof.log 2>&1 )
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam* /tmp/s.list
: > /tmp/s8.list
for k in 0 1 2 3 4 5 6 7; do $GEN -o /tmp/m$k.bam -n 40000000 -S $((42+k)) -t 32 2>> $O/gen.txt; echo /tmp/m$k.bam >> /tmp/s8.list; done
$CLI -i /tmp/s8.list -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do
  run list8_one_ctx_$rep "gpus=1" /tmp/s8.list
  run list8_peer_$rep "gpus=8" /tmp/s8.list
  PANDEPTH_RCCL_LIB=$SHIM run list8_loopback_rccl_$rep "gpus=8,transport=rccl" /tmp/s8.list
done
rm -f /tmp/o_* /tmp/warm* /tmp/m?.bam* /tmp/s8.list
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/bench.py --e2e-records 0 --e2e-multi-records 0 > $O/kt_bench.json 2> $O/kt.log ); find $O/kt -name "*kernel_stats.csv" | head -2
timeout 600 bash tools/pmc_collect.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
mkdir -p $O/pmc && cp gpurun_out/pmc/*_pdk.csv $O/pmc/ 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete; find $O -name "*agent_info.csv" -delete
cat $O/gen.txt; cat $O/summary.txt | cut -c1-500; head -40 $O/decode_timeline.txt | cut -c1-200; du -sh $O
