# round 3, call 11: the per-site / window gzip streams with the LZ77 parse on the device: tests, and config 4 end to end (2e7 records, 6e7 per-site lines)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c11; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_lz77.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_lz77.log 2>&1
mkdir -p /tmp/e2e && cd /tmp/e2e
tools=$GRAFT_REPO_ROOT/tools; $tools/bamgen -o w.bam -n 20000000 -t 32 2> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; R=$GRAFT_REPO_ROOT/oracle/_ref/pandepth_ref
for k in 1 2 3; do sleep 1; ( /usr/bin/time -f "wall %e s" env PANDEPTH_TIMING=1 PGZ_DEBUG=1 $P -i w.bam -w 100 -a -o dev -t 16 ) > /dev/null 2>> $O/dev.log; done
( /usr/bin/time -f "wall %e s" env PANDEPTH_TIMING=1 PANDEPTH_DEVICE_DEFLATE=0 $P -i w.bam -w 100 -a -o host -t 16 ) > /dev/null 2> $O/host.log
( /usr/bin/time -f "wall %e s" $R -i w.bam -w 100 -a -o ref -t 36 ) > /dev/null 2> $O/ref.log
cmp dev.SiteDepth.gz ref.SiteDepth.gz && cmp dev.win.stat.gz ref.win.stat.gz && echo "device-parse files byte-identical with the reference" >> $O/dev.log
cmp host.SiteDepth.gz ref.SiteDepth.gz && echo "host-parse file byte-identical with the reference" >> $O/host.log
rm -rf /tmp/e2e
