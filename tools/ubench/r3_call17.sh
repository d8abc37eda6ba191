# round 3, call 17: what the process's exit waits for (0.2 s in most config-4 runs, 2 ms in some)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c17; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"
def run(cmd, env=None, tag="", keys=("pd_destroy",)):
    t0=time.time()
    p=subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    last=None; lines=[]
    for l in p.stderr:
        last=time.time()-t0; lines.append(l.decode().rstrip())
    p.wait(); dt=time.time()-t0
    print("%s: wall %.3f s rc %d; last stderr line at %.3f s, exit %.3f s later" % (tag, dt, p.returncode, last or 0, dt-(last or 0)))
    for l in lines:
        if any(k in l for k in keys): print("    %s" % l[:230])
for tag,env,args in (("quick exit",{},["-w","100","-a"]),("quick exit after freeing the text rings and parse buffers",{"PANDEPTH_EXIT_FREE":"1"},["-w","100","-a"]),
                     ("orderly",{"PANDEPTH_ORDERLY_EXIT":"1"},["-w","100","-a"])):
    for k in range(5):
        time.sleep(0.7)
        run([cli,"-i","w.bam"]+args+["-o","dev","-t","16"], dict({"PANDEPTH_TIMING":"1"}, **env), "%s #%d" % (tag,k), ("per-site writer (","pd_destroy"))
PY
rm -rf /tmp/e2e
