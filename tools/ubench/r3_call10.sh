# round 3, call 10: pd_deflate_parse on the device: correctness against zlib and timing on per-site text
O=$GRAFT_REPO_ROOT/gpurun_out/r3c10; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_lz77.py -x -q -m gpu 2>&1 | tail -12 ) > $O/tests_lz77.log 2>&1
python3 - <<'PY'
import random
random.seed(3)
d=50; buf=[]
with open('/tmp/site.txt','w') as out:
    for j in range(7000000):
        if random.random()<0.3: d=max(0,d+random.choice((-1,1,1,-1,2,-2)))
        buf.append("Chr01\t%d\t%d\n" % (j,d))
        if len(buf)>=200000: out.write("".join(buf)); buf=[]
    out.write("".join(buf))
PY
for geo in "262144 16384" "65536 4096" "32768 4096" "1048576 65536"; do
  ( timeout 600 tests/harness/lz77_gpu_check /tmp/site.txt $geo ) >> $O/timing.log 2>&1
done
