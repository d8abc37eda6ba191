#!/bin/bash
# round 4, call 29: readers and submitters instead of feeders that do both (host/pipeline.cpp) — the executable before (tools/ubench/pandepth_prev, same library)
# and after on one 3e8-record file, alternating; then the tests that run the executable on the GPU
O=$GRAFT_REPO_ROOT/gpurun_out/r4c29; mkdir -p $O; export TMPDIR=/tmp
mkdir -p /tmp/e2e && cd /tmp/e2e && $GRAFT_REPO_ROOT/tools/bamgen -o s.bam -n 300000000 -t 32 2> $O/gen.log
NEW=$GRAFT_REPO_ROOT/pandepth_amd/pandepth; OLD=$GRAFT_REPO_ROOT/tools/ubench/pandepth_prev; export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/pandepth_amd
$NEW -i s.bam -o warm -t 16 > /dev/null 2>&1; sleep 1
run() { tag=$1; bin=$2; shift 2; TIMEFORMAT="wall %R s user %U sys %S"; ( time env PANDEPTH_TIMING=1 "$@" $bin -i s.bam -o m -t 16 ) > $O/run.tmp 2>&1
  echo "$tag: $(grep -E 'decode \+ scatter' $O/run.tmp | sed 's/  */ /g') | $(grep wall $O/run.tmp) | $(grep -o 'feeder thread-seconds: [^;]*' $O/run.tmp) | $(grep -o 'device ms summed over batches: [^;]*' $O/run.tmp)" >> $O/ab.log
  cmp -s m.chr.stat.gz warm.chr.stat.gz || echo "  OUTPUT DIFFERS" >> $O/ab.log; sleep 0.5; }
for r in 1 2 3; do run "before      " $OLD X=1; run "after       " $NEW X=1; done
run "after, 8 sub" $NEW PANDEPTH_TUNE=dd_threads=8
run "after, 4 sub" $NEW PANDEPTH_TUNE=dd_threads=4
run "after, 64 MB" $NEW PANDEPTH_TUNE=dd_batch_mb=64
run "before      " $OLD X=1; run "after       " $NEW X=1
cat $O/ab.log
rm -rf /tmp/e2e
cd $GRAFT_REPO_ROOT; ( time timeout 400 python -m pytest tests/test_cli_gpu.py tests/test_z_cli_gpu_late.py -m gpu -q -x --timeout 300 ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
