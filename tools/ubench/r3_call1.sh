# round 3, first GPU call: new tests, the default bench line, CLI kernel trace at 3e8 records, feeder / batch sweeps
O=$GRAFT_REPO_ROOT/gpurun_out/r3c1; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_comm_loopback_gpu.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > $O/tests_comm.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_engine.log 2>&1
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err
# CLI at 3e8: timing, sweeps, kernel trace
mkdir -p /tmp/e2e && cd /tmp/e2e
( time timeout 900 $GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 300000000 -t 32 ) > $O/gen.log 2>&1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
for cfg in "6 32" "6 32" "8 32" "12 32" "6 64" "8 64" "6 16" "12 16"; do set -- $cfg; sleep 1
  PANDEPTH_DD_THREADS=$1 PANDEPTH_DD_BATCH_MB=$2 PANDEPTH_TIMING=1 python3 -c "
import subprocess,time
t0=time.time(); p=subprocess.run(['$P','-i','b.bam','-o','m','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE); dt=time.time()-t0
e=p.stderr.decode(); print('feeders $1 batchMB $2 wall %.3f' % dt); print('\n'.join(l[:400] for l in e.splitlines() if 'engine create' in l or 'decode + scatter' in l or 'device decode' in l or 'decode entry' in l or 'pd_create' in l))" >> $O/sweep.log 2>&1
done
cd /tmp && PANDEPTH_ORDERLY_EXIT=1 rocprofv3 --kernel-trace --stats -d $O/prof -o cli -- $P -i /tmp/e2e/b.bam -o /tmp/e2e/dd -t 16 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/cli_kernel_stats.csv 2>/dev/null; rm -rf $O/prof
rm -rf /tmp/e2e
