# round 3, call 22: inflate phase 3 with sources redirected through earlier matches of the batch, against a build without (libpandepth_amd_nr.so)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c22; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bgzf.py -x -q -m gpu > $O/pytest.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 300000000 -t 32 2> $O/gen.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
cp $GRAFT_REPO_ROOT/pandepth_amd/libpandepth_amd.so /tmp/e2e/new.so
run() { for k in 1 2 3; do python3 -c "
import subprocess,time,os
t0=time.time(); p=subprocess.run(['$CLI','-i','b.bam','-o','m','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=dict(os.environ,PANDEPTH_TIMING='1')); dt=time.time()-t0
l=[x for x in p.stderr.decode().splitlines() if 'decode + scatter' in x]
print('$1: wall %.3f s; %s' % (dt, l[0].strip() if l else ''))" >> $O/e2e.log; sleep 1; done; }
run "redirect"
cp $GRAFT_REPO_ROOT/tools/ubench/libpandepth_amd_nr.so $GRAFT_REPO_ROOT/pandepth_amd/libpandepth_amd.so
run "no redirect"
cp /tmp/e2e/new.so $GRAFT_REPO_ROOT/pandepth_amd/libpandepth_amd.so
run "redirect again"
md5sum m.chr.stat.gz >> $O/e2e.log
rm -rf /tmp/e2e
