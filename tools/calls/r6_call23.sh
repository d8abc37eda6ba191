#!/bin/bash
# round 6, call 23: PGZ_DEBUG on the per-site writer (`-w 100 -a`, 1e8 records): a round's stages against the clock
O=$GRAFT_REPO_ROOT/gpurun_out/r6c23; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tools/bamgen -o /tmp/m.bam -n 100000000 -t 32 2>> $O/gen.log
P=$GRAFT_REPO_ROOT/pandepth_amd
$P/pandepth -i /tmp/m.bam -o /tmp/warm -t 16 > /dev/null 2>&1; sleep 1
for tn in lz_calls=2 lz_calls=4; do
( cd /tmp && PGZ_DEBUG=1 PANDEPTH_TUNE=$tn PANDEPTH_TIMING=1 timeout 600 $P/pandepth -i /tmp/m.bam -w 100 -a -o /tmp/o_s -t 16 > $O/pgz_${tn//=/_}.log 2>&1 )
echo "== $tn"; grep "per-site writer" $O/pgz_${tn//=/_}.log | cut -c1-200; grep "\[pgz\]" $O/pgz_${tn//=/_}.log | sed -n 60,84p | cut -c1-260
done
rm -f /tmp/o_* /tmp/warm* /tmp/m.bam*
