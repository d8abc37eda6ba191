mkdir -p gpurun_out/site; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/site
(timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "format_sites" 2>&1 | tail -15) > $O/t1.log; tail -5 $O/t1.log
(timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "_a or site or a-" 2>&1 | tail -5) > $O/t2.log; tail -3 $O/t2.log
mkdir -p /tmp/e2e && cd /tmp/e2e && timeout 200 $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 20000000 -t 32 2>&1 | tail -1
for i in 1 2; do PANDEPTH_TIMING=1 timeout 300 $GRAFT_REPO_ROOT/pandepth_amd/pandepth -i s.bam -w 100 -a -o mine -t 16 > $O/cli$i.out 2> $O/cli$i.err; grep -E "per-site|decode \+|total|table" $O/cli$i.err | cut -c1-150; done
python3 - <<'PY'
import subprocess, time, os
t0=time.time(); subprocess.run([os.environ["GRAFT_REPO_ROOT"]+"/pandepth_amd/pandepth","-i","s.bam","-w","100","-a","-o","mine2","-t","16"],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); print("pandepth -w 100 -a wall %.3f s" % (time.time()-t0))
t0=time.time(); subprocess.run([os.environ["GRAFT_REPO_ROOT"]+"/oracle/_ref/pandepth_ref","-i","s.bam","-w","100","-a","-o","ref","-t","36"],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); print("reference wall %.3f s" % (time.time()-t0))
for f in ("win.stat.gz","SiteDepth.gz"):
    a=open("mine2."+f,"rb").read(); b=open("ref."+f,"rb").read(); print(f, len(a), "identical" if a==b else "DIFFERENT")
PY
