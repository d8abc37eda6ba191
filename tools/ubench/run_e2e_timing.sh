mkdir -p gpurun_out/e2et; cd /tmp && export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 300 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 100000000 -t 32 2>&1 | tail -1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
for T in 6 6 8 8 10 12; do sleep 1; PANDEPTH_DD_THREADS=$T PANDEPTH_TIMING=1 python3 -c "
import subprocess,time
t0=time.time(); p=subprocess.run(['$P','-i','s.bam','-o','m','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE); dt=time.time()-t0
e=p.stderr.decode(); print('feeders $T wall %.3f' % dt, [l[36:70] for l in e.splitlines() if 'engine create' in l or 'decode + scatter' in l])"; done
