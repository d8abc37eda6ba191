#!/bin/bash
# round 4, call 4: the wave inflate kernel before / after the balanced piece copies, and with more waves per CU (9-bit root)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/bamgen -o /tmp/ab.bam -n 12000000 -t 32 2> $O/gen.log
U=$GRAFT_REPO_ROOT/tools/ubench
for pass in 1 2; do
  PANDEPTH_AMD_LIB=$U/libpd_inflate_old.so WAVES=16 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v1.so WAVES=12,16 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v2.so WAVES=16,20,22 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
  PANDEPTH_AMD_LIB=$U/libpd_inflate_v3.so WAVES=16,20,22 timeout 200 python $U/inflate_ab.py /tmp/ab.bam >> $O/ab.log 2>&1
done
cat $O/ab.log
# the GPU tests of the decoder with the working tree's library (v1)
cd $GRAFT_REPO_ROOT && VARIANTS=c803,c703,c704,c802 EXPORT=0 timeout 300 python tools/ubench/direct_ab.py > $O/direct_ab.log 2>&1; tail -5 $O/direct_ab.log; timeout 900 python -m pytest tests/test_gpu_bgzf.py tests/test_cli_gpu.py tests/test_gpu_engine.py -q -x --timeout 300 > $O/pytest_bgzf.log 2>&1; tail -3 $O/pytest_bgzf.log
