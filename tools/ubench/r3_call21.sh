# round 3, call 21: k_lz_parse with XCD-aware chunk placement: waves per XCD (PD_LZ_PER_XCD) on config 4
O=$GRAFT_REPO_ROOT/gpurun_out/r3c21; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_lz77.py -x -q -m gpu > $O/pytest.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
for P in 1024; do
  echo "== PD_LZ_PER_XCD=$P" >> $O/e2e.log
  PD_LZ_PER_XCD=$P PANDEPTH_TIMING=1 PD_LZ_DEBUG=1 $CLI -i w.bam -w 100 -a -o dev -t 16 2>&1 >/dev/null | grep "\[lz\]" | awk '{p+=$(NF-6)+0} {print} ' | sed -n 3,14p | cut -c1-200 >> $O/e2e.log
  for k in 1 2 3; do python3 -c "
import subprocess,time,os
t0=time.time(); subprocess.run(['$CLI','-i','w.bam','-w','100','-a','-o','dev','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL,env=dict(os.environ,PD_LZ_PER_XCD='$P')); print('wall %.3f s' % (time.time()-t0))" >> $O/e2e.log; done
done
rm -rf /tmp/e2e
