# round 3, call 18: config 4 with the resident per-site text: stage timers per round, then a kernel trace of the same command
O=$GRAFT_REPO_ROOT/gpurun_out/r3c18; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e
$GRAFT_REPO_ROOT/tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
CLI=$GRAFT_REPO_ROOT/pandepth_amd/pandepth
PANDEPTH_TIMING=1 PGZ_DEBUG=1 PD_LZ_DEBUG=1 $CLI -i w.bam -w 100 -a -o dev -t 16 > /dev/null 2> $O/debug.log
for k in 1 2 3; do python3 -c "
import subprocess,time
t0=time.time(); subprocess.run(['$CLI','-i','w.bam','-w','100','-a','-o','dev','-t','16'],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL); print('wall %.3f s' % (time.time()-t0))" >> $O/wall.log; sleep 1; done
cd /tmp && PANDEPTH_ORDERLY_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cfg4 -- $CLI -i /tmp/e2e/w.bam -w 100 -a -o /tmp/e2e/prof -t 16 > $O/prof.log 2>&1
rm -rf /tmp/e2e
