# round 3, call 23: the process's exit: _exit with everything still allocated, _exit after pd_destroy, orderly return — config 4 and the 1e8-record whole-chromosome run
O=$GRAFT_REPO_ROOT/gpurun_out/r3c23; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
$GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 100000000 -t 32 2>> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"
def run(cmd, env, tag):
    t0=time.time()
    p=subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    last=None; lines=[]
    for l in p.stderr:
        last=time.time()-t0; lines.append(l.decode().rstrip())
    p.wait(); dt=time.time()-t0
    d=[x for x in lines if "pd_destroy" in x]
    print("%s: wall %.3f s; last stderr line at %.3f s, exit %.3f s later %s" % (tag, dt, last or 0, dt-(last or 0), d[0].strip()[:150] if d else ""))
for name,args in (("config 4",["-i","w.bam","-w","100","-a"]),("1e8 whole-chromosome",["-i","b.bam"])):
    for tag,env in (("_exit",{}),("pd_destroy then _exit",{"PANDEPTH_EXIT_DESTROY":"1"}),("orderly",{"PANDEPTH_ORDERLY_EXIT":"1"})):
        for k in range(4):
            time.sleep(0.7)
            run([cli]+args+["-o","dev","-t","16"], dict({"PANDEPTH_TIMING":"1"}, **env), "%s, %s #%d" % (name,tag,k))
PY
rm -rf /tmp/e2e
