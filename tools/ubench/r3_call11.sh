# round 3, call 11: config 4 end to end (2e7 records, 6e7 per-site lines) with the gzip streams' LZ77 parse on the device / on the host threads / the reference
O=$GRAFT_REPO_ROOT/gpurun_out/r3c11; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
python3 - > $O/e2e.log 2>&1 <<'PY'
import os, subprocess, time
R=os.environ["GRAFT_REPO_ROOT"]; cli=R+"/pandepth_amd/pandepth"; ref=R+"/oracle/_ref/pandepth_ref"
def run(cmd, env=None, tag=""):
    t0=time.time(); p=subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {}))); dt=time.time()-t0
    print("%s: wall %.3f s rc %d" % (tag, dt, p.returncode))
    for l in p.stderr.decode().splitlines():
        if any(k in l for k in ("per-site writer","per-site file","table gzip","deflate_parse","[pgz]","decode + scatter","engine create","scan + statistics")): print("   ", l[:230])
    return dt
for k in range(2):
    time.sleep(1); run([cli,"-i","w.bam","-w","100","-a","-o","dev","-t","16"], {"PANDEPTH_TIMING":"1","PGZ_DEBUG":"1"}, "device parse #%d" % k)
time.sleep(1); run([cli,"-i","w.bam","-w","100","-a","-o","host","-t","16"], {"PANDEPTH_TIMING":"1","PGZ_DEBUG":"1","PANDEPTH_DEVICE_DEFLATE":"0"}, "host parse")
run([ref,"-i","w.bam","-w","100","-a","-o","ref","-t","36"], None, "reference")
for a,b in (("dev.SiteDepth.gz","ref.SiteDepth.gz"),("dev.win.stat.gz","ref.win.stat.gz"),("host.SiteDepth.gz","ref.SiteDepth.gz")):
    print(a, "==", b, open(a,"rb").read()==open(b,"rb").read())
PY
rm -rf /tmp/e2e
