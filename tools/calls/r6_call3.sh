#!/bin/bash
# round 6, call 3: what device allocations cost (tools/ubench/devalloc), the communicator + its exchange buffers beside the decode (-X comm_early) on the
# 3e8-record list, eight contexts on one GPU after the parallel set-up, long reads with the CG tag walked on the device, and the direct kernel's
# prefetch forms (registers / touch)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c3; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 120 tools/ubench/devalloc > $O/devalloc.txt 2>&1; cat $O/devalloc.txt
( time timeout 1500 python -m pytest tests/test_comm_loopback_gpu.py tests/test_comm_gpu.py tests/test_long_reads.py tests/test_gpu_bgzf.py tests/test_host_generated.py tests/test_cli_gpu.py -m gpu -q --timeout 900 -x ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; SHIM=$GRAFT_REPO_ROOT/tests/harness/libpd_loopback_nccl.so
run() { # name tune input
  local t0=$(date +%s%N)
  ( cd /tmp && PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i $3 -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s%N)
  echo "$1 [$2] rc $rc wall $(( (t1 - t0) / 1000000 )) ms | $(grep -E 'decode \+ scatter' $O/cli_$1.log | tr -s ' ') | $(grep -E 'comm (init|ahead)' $O/cli_$1.log | tr -s ' ' | cut -c1-200 | tr '\n' ';') | $(grep -E 'summed over|added into' $O/cli_$1.log | cut -c10-100) | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | tail -1 | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
$GEN -o /tmp/l.bam -n 600000 --long -t 32 2>> $O/gen.txt
$CLI -i /tmp/l.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do run long_$rep "x=1" /tmp/l.bam; done
run long_host_1 "device_decode=0" /tmp/l.bam
grep -h "device decode\|decode entry" $O/cli_long_1.log | cut -c1-900 >> $O/summary.txt
( cd /tmp && t0=$(date +%s%N); $GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i /tmp/l.bam -o /tmp/ref_l -t 16 > /dev/null 2>&1; t1=$(date +%s%N); echo "long: reference -t 16 wall $(( (t1 - t0) / 1000000 )) ms" >> $O/summary.txt; cmp /tmp/ref_l.chr.stat.gz /tmp/o_long_1.chr.stat.gz && echo "long: same as the reference" >> $O/summary.txt )
rm -f /tmp/o_* /tmp/warm* /tmp/l.bam* /tmp/ref_l*
$GEN -o /tmp/s.bam -n 300000000 -t 32 2>> $O/gen.txt
echo /tmp/s.bam > /tmp/s.list
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do
  run list_nocomm_$rep "x=1" /tmp/s.list
  run list_peer_early_$rep "comm=force" /tmp/s.list
  run list_peer_inline_$rep "comm=force,comm_early=0" /tmp/s.list
  run list_rccl_early_$rep "comm=force,transport=rccl" /tmp/s.list
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam* /tmp/s.list
: > /tmp/s8.list
for k in 0 1 2 3 4 5 6 7; do $GEN -o /tmp/m$k.bam -n 40000000 -S $((42+k)) -t 32 2>> $O/gen.txt; echo /tmp/m$k.bam >> /tmp/s8.list; done
$CLI -i /tmp/s8.list -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do
  run list8_one_ctx_$rep "gpus=1" /tmp/s8.list
  run list8_peer_$rep "gpus=8" /tmp/s8.list
  run list8_peer_inline_$rep "gpus=8,comm_early=0" /tmp/s8.list
done
rm -f /tmp/o_* /tmp/warm* /tmp/m?.bam* /tmp/s8.list
VARIANTS=c0,c2704,c3703,c3702,c3604,c4704,c4703,c3802,c0 timeout 600 python tools/ubench/direct_ab.py > $O/direct_ab.txt 2>&1; grep -E "variant" $O/direct_ab.txt | cut -c1-200
cat $O/gen.txt; cat $O/summary.txt | cut -c1-600
