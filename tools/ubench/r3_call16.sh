# round 3, call 16: the per-site file with its text resident on the device (pd_text_*): tests, then config 4 end to end with exit times
O=$GRAFT_REPO_ROOT/gpurun_out/r3c16; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lz77.py tests/test_gpu_engine.py -x -q -m gpu -k "lz or text_stream or format_sites or device_parse or cli" > $O/pytest.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time, hashlib
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"; ref=R+"/oracle/_ref/pandepth_ref"
def run(cmd, env=None, tag="", keys=("per-site writer","engine create","decode + scatter","scan + stat","table gzip","per-site file")):
    t0=time.time()
    p=subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
    last=None; lines=[]
    for l in p.stderr:
        last=time.time()-t0; lines.append(l.decode().rstrip())
    p.wait(); dt=time.time()-t0
    print("%s: wall %.3f s rc %d; last stderr line at %.3f s, exit %.3f s later" % (tag, dt, p.returncode, last or 0, dt-(last or 0)))
    for l in lines:
        if any(k in l for k in keys): print("    %s" % l[:230])
    return dt
sha={}
for tag,env in (("resident",{}),("resident 32/8/128",{"PGZ_DEV_CHUNK_KB":"32","PGZ_DEV_BATCH_MB":"128"}),("resident 16/4/96",{"PGZ_DEV_CHUNK_KB":"16","PGZ_DEV_TAIL_KB":"4","PGZ_DEV_BATCH_MB":"96"}),("host text",{"PANDEPTH_SITE_RESIDENT":"0","PGZ_DEV_CHUNK_KB":"32","PGZ_DEV_BATCH_MB":"128"})):
    for k in range(3):
        time.sleep(1.0)
        run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], dict(env, PANDEPTH_TIMING="1"), "%s #%d" % (tag,k), ("per-site writer","[pgz]","[lz]") if k==0 else ("per-site writer",))
    sha[tag]=hashlib.sha256(open("dev.SiteDepth.gz","rb").read()).hexdigest()
run([ref,"-i","w.bam","-w","100","-a","-o","ref","-t","36"], None, "reference")
h=hashlib.sha256(open("ref.SiteDepth.gz","rb").read()).hexdigest()
print({k: v==h for k,v in sha.items()}, "win:", open("dev.win.stat.gz","rb").read()==open("ref.win.stat.gz","rb").read())
PY
rm -rf /tmp/e2e
