#!/bin/bash
# round 6, call 2: the in-process transport after its all-reduce fix (the comm / list tests), the list path's transports A/B on a 3e8-record file
# (no communicator / peer / RCCL ahead of the contexts, 3 runs each), eight contexts on this one GPU, first contact with the long-read and
# 40-level-quality files, the H2D-by-kernel knob, and the kernel A/Bs (k_direct_c8 zero-on-read, k_sweep_i4_fast's covered count, export)
O=$GRAFT_REPO_ROOT/gpurun_out/r6c2; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/test_comm_loopback_gpu.py tests/test_comm_gpu.py tests/test_cli_gpu.py tests/test_z_cli_gpu_late.py tests/test_gpu_engine.py -m gpu -q --timeout 900 -k "list or comm or sliced or ranks or slice or export or overflow or chunked" ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; SHIM=$GRAFT_REPO_ROOT/tests/harness/libpd_loopback_nccl.so
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
echo /tmp/s.bam > /tmp/s.list
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name tune input
  local t0=$(date +%s%N)
  ( cd /tmp && PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i $3 -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 ); local rc=$?
  local t1=$(date +%s%N)
  echo "$1 [$2] rc $rc wall $(( (t1 - t0) / 1000000 )) ms | $(grep -E 'decode \+ scatter' $O/cli_$1.log | tr -s ' ') | $(grep -E 'comm (init|ahead)' $O/cli_$1.log | tr -s ' ' | cut -c1-110 | tr '\n' ';') | $(grep -E 'summed over|added into' $O/cli_$1.log | cut -c1-100) | $(zcat /tmp/o_$1.chr.stat.gz 2>/dev/null | tail -1 | md5sum | cut -c1-8)" >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run single_$rep "x=1" /tmp/s.bam
  run list_nocomm_$rep "x=1" /tmp/s.list
  run list_peer_$rep "comm=force" /tmp/s.list
  run list_rccl_$rep "comm=force,transport=rccl" /tmp/s.list
done
for rep in 1 2 3; do for k in 0 1 2 3; do run h2d${k}_$rep "h2d_kernel=$k" /tmp/s.bam; done; done
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
$GEN -o /tmp/l.bam -n 600000 --long -t 32 2>> $O/gen.txt; ls -la /tmp/l.bam >> $O/gen.txt
$CLI -i /tmp/l.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do run long_$rep "x=1" /tmp/l.bam; run long_host_$rep "device_decode=0" /tmp/l.bam; done
grep -h "device decode" $O/cli_long_1.log | cut -c1-700 >> $O/summary.txt
( cd /tmp && t0=$(date +%s%N); $GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i /tmp/l.bam -o /tmp/ref_l -t 16 > /dev/null 2>&1; t1=$(date +%s%N); echo "long: reference -t 16 wall $(( (t1 - t0) / 1000000 )) ms" >> $O/summary.txt; cmp /tmp/ref_l.chr.stat.gz /tmp/o_long_1.chr.stat.gz && echo "long: same as the reference" >> $O/summary.txt )
rm -f /tmp/o_* /tmp/warm* /tmp/l.bam* /tmp/ref_l*
$GEN -o /tmp/q.bam -n 200000000 -Q 40 -t 32 2>> $O/gen.txt; ls -la /tmp/q.bam >> $O/gen.txt
$CLI -i /tmp/q.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do run q40_$rep "x=1" /tmp/q.bam; done
grep -h "device decode" $O/cli_q40_1.log | cut -c1-700 >> $O/summary.txt
rm -f /tmp/o_* /tmp/warm* /tmp/q.bam*
# kernels
VARIANTS=c0,c2704,c2703,c2804,c1704,c0 timeout 600 python tools/ubench/direct_ab.py > $O/direct_ab.txt 2>&1; tail -12 $O/direct_ab.txt
for lib in "" $GRAFT_REPO_ROOT/pandepth_amd/alt_old/libpandepth_amd.so; do
  PANDEPTH_AMD_LIB=$lib PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --e2e-records 0 --e2e-multi-records 0 2> $O/bench_dist.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('lib=[$lib]', 'ms_per_step', round(d['device_step']['ms_per_step'] if 'device_step' in d else d['ms_per_step'],3), {n:(k[n]['avg_ms'] if k.get(n) else None) for n in ('direct_export','slice_sweep','gather_windows')})" >> $O/kernels_ab.txt 2>&1
done
PD_BENCH_DIRECT_UN=2704 PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --e2e-records 0 --e2e-multi-records 0 2>> $O/bench_dist.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernels']; print('direct_un=2704', 'ms_per_step', round(d['device_step']['ms_per_step'],3), {n:(k[n]['avg_ms'] if k.get(n) else None) for n in ('direct_export','slice_sweep','gather_windows')})" >> $O/kernels_ab.txt 2>&1
cat $O/gen.txt; cat $O/summary.txt; cat $O/kernels_ab.txt
