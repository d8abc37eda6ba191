# round 3, call 6: inflate kernel with the LDS span (isolated rate + device tests), sweep_i4 split, e2e timing at 3e8
O=$GRAFT_REPO_ROOT/gpurun_out/r3c6; mkdir -p $O; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bgzf.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_bgzf.log 2>&1
( BGZF_VARIANTS=258,2,32770 timeout 600 python tools/bgzf_gpu_bench.py 6e6 > $O/inflate_l6.log 2>&1 )
( BGZF_LEVEL=1 BGZF_VARIANTS=258 timeout 600 python tools/bgzf_gpu_bench.py 6e6 > $O/inflate_l1.log 2>&1 )
( PD_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_1rank.json 2> $O/bench_1rank.err )
mkdir -p /tmp/e2e && cd /tmp/e2e
( timeout 900 $GRAFT_REPO_ROOT/tools/bamgen -o b.bam -n 300000000 -t 32 ) > $O/gen.log 2>&1
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; R=$GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref
for k in 1 2 3; do sleep 1; PANDEPTH_TIMING=1 python3 -c "
import subprocess,time
t0=time.time(); p=subprocess.run(['$P','-i','b.bam','-o','m','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE); dt=time.time()-t0
e=p.stderr.decode(); print('wall %.3f' % dt); print('\n'.join(l[:420] for l in e.splitlines() if 'engine create' in l or 'decode + scatter' in l or 'device decode' in l))" >> $O/e2e.log 2>&1; done
$R -i b.bam -o r -t 36 > /dev/null 2>&1; cmp m.chr.stat.gz r.chr.stat.gz && echo "byte-identical with the reference" >> $O/e2e.log
rm -rf /tmp/e2e
