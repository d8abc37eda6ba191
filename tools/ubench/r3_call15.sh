# round 3, call 15: where the wall time outside the program's own phase timers goes (start-up before main's first timer, teardown after the last)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c15; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"
def run(cmd, env=None, tag=""):
    t0=time.time()
    p=subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    first=None; last=None; lines=[]
    for l in p.stderr:
        now=time.time()-t0
        if first is None: first=now
        last=now; lines.append((now,l.decode().rstrip()))
    p.wait(); dt=time.time()-t0
    print("%s: wall %.3f s rc %d; first stderr line at %.3f s, last at %.3f s, exit %.3f s later" % (tag, dt, p.returncode, first or 0, last or 0, dt-(last or 0)))
    for t,l in lines:
        if any(k in l for k in ("engine create","per-site file")): print("    @%.3f %s" % (t,l[:160]))
def dirty():
    d={}
    for l in open("/proc/meminfo"):
        k,v=l.split(":"); d[k]=v.strip()
    return "Dirty %s, Writeback %s, MemFree %s, Cached %s" % (d["Dirty"], d["Writeback"], d["MemFree"], d["Cached"])
for k in range(4):
    print(dirty()); run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1"}, "device parse, no sync #%d" % k)
t0=time.time(); os.sync(); print("sync: %.3f s" % (time.time()-t0))
for k in range(4):
    print(dirty()); run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1"}, "device parse, after sync #%d" % k)
    if k == 1: t0=time.time(); os.sync(); print("sync: %.3f s" % (time.time()-t0))
t0=time.time(); subprocess.run([cli,"-h"],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); print("pandepth -h: %.3f s" % (time.time()-t0))
t0=time.time(); subprocess.run(["/bin/true"]); print("/bin/true: %.3f s" % (time.time()-t0))
PY
rm -rf /tmp/e2e
