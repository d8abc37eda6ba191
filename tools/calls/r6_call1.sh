#!/bin/bash
# (this call lost its CLI legs: /usr/bin/time does not exist on the GPU box — the suite ran: 591 passed, 12 failed, all of them the in-process transport, whose
# all-reduce dropped the last count % 4 words; fixed and re-run in call 2)
# round 6, call 1: smoke + the whole gpu suite with the round's first changes (in-process transport, RCCL ahead of the contexts, own stdout, the patched
# reference RUN on the engine, long-read fixtures), then the list path's transports A/B on a 3e8-record file (no communicator / peer / RCCL, 3 runs each),
# an 8-context list run on this one GPU over both transports, and first contact with the long-read and 40-level-quality files
O=$GRAFT_REPO_ROOT/gpurun_out/r6c1; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 ) > $O/pytest_gpu.log 2>&1; echo rc=$? >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; GEN=tools/bamgen; SHIM=$GRAFT_REPO_ROOT/tests/harness/libpd_loopback_nccl.so
$GEN -o /tmp/s.bam -n 300000000 -t 32 2> $O/gen.txt
echo /tmp/s.bam > /tmp/s.list
$CLI -i /tmp/s.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
run() { # name tune input
  ( cd /tmp && /usr/bin/time -f "%e s wall" env PANDEPTH_TUNE=$2 PANDEPTH_TIMING=1 timeout 300 $CLI -i $3 -o /tmp/o_$1 -t 16 > $O/cli_$1.log 2>&1 )
  echo "$1 [$2] $(tail -1 $O/cli_$1.log) | $(grep -E 'decode \+ scatter' $O/cli_$1.log | tr -s ' ') | $(grep -E 'comm (init|ahead)' $O/cli_$1.log | tr -s ' ' | cut -c1-120 | tr '\n' ';') | $(grep -E 'summed over|added into' $O/cli_$1.log | cut -c1-100)" >> $O/summary.txt
  zcat /tmp/o_$1.chr.stat.gz | tail -1 | md5sum | cut -c1-8 >> $O/summary.txt
  sleep 1
}
for rep in 1 2 3; do
  run single_$rep "x=1" /tmp/s.bam
  run list_nocomm_$rep "x=1" /tmp/s.list
  run list_peer_$rep "comm=force" /tmp/s.list
  run list_rccl_$rep "comm=force,transport=rccl" /tmp/s.list
done
rm -f /tmp/o_* /tmp/warm* /tmp/s.bam* /tmp/s.list
# eight contexts on this one GPU: 8 BAMs of 4e7 records
: > /tmp/s8.list
for k in 0 1 2 3 4 5 6 7; do $GEN -o /tmp/m$k.bam -n 40000000 -S $((42+k)) -t 32 2>> $O/gen.txt; echo /tmp/m$k.bam >> /tmp/s8.list; done
$CLI -i /tmp/s8.list -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do
  run list8_one_ctx_$rep "gpus=1" /tmp/s8.list
  run list8_peer_$rep "gpus=8" /tmp/s8.list
  PANDEPTH_RCCL_LIB=$SHIM run list8_loopback_rccl_$rep "gpus=8,transport=rccl" /tmp/s8.list
done
rm -f /tmp/o_* /tmp/warm* /tmp/m?.bam* /tmp/s8.list
# long reads and unbinned qualities
$GEN -o /tmp/l.bam -n 600000 --long -t 32 2>> $O/gen.txt; ls -la /tmp/l.bam >> $O/gen.txt
$CLI -i /tmp/l.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2; do run long_$rep "x=1" /tmp/l.bam; run long_host_$rep "device_decode=0" /tmp/l.bam; done
grep -h "device decode" $O/cli_long_1.log | cut -c1-600 >> $O/summary.txt
( cd /tmp && /usr/bin/time -f "%e s wall (reference -t 16)" $GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref -i /tmp/l.bam -o /tmp/ref_l -t 16 > /dev/null 2>> $O/summary.txt; cmp /tmp/ref_l.chr.stat.gz /tmp/o_long_1.chr.stat.gz && echo "long: same as the reference" >> $O/summary.txt )
rm -f /tmp/o_* /tmp/warm* /tmp/l.bam* /tmp/ref_l*
$GEN -o /tmp/q.bam -n 200000000 -Q 40 -t 32 2>> $O/gen.txt; ls -la /tmp/q.bam >> $O/gen.txt
$CLI -i /tmp/q.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for rep in 1 2 3; do run q40_$rep "x=1" /tmp/q.bam; done
grep -h "device decode" $O/cli_q40_1.log | cut -c1-600 >> $O/summary.txt
rm -f /tmp/o_* /tmp/warm* /tmp/q.bam*
cat $O/gen.txt; cat $O/summary.txt
