mkdir -p gpurun_out/e2et; cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/e2et
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 300 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 100000000 -t 32 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
for T in 1 2 3; do sleep 1; PANDEPTH_TIMING=1 python3 -c "
import subprocess,time
t0=time.time(); p=subprocess.run(['$P','-i','s.bam','-o','m','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE); dt=time.time()-t0
e=p.stderr.decode(); print('wall %.3f' % dt); print('\n'.join(l[:110] for l in e.splitlines() if l.startswith('[timing] ')))"; done
sleep 1; python3 -c "
import subprocess,time
t0=time.time(); subprocess.run(['$P','-h'],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); print('pandepth -h wall %.3f' % (time.time()-t0))"
