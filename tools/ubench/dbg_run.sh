mkdir -p gpurun_out/r2b; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r2b/gen.log 2>&1
import sys, os
sys.path.insert(0,'.')
from tools import synth
names, lens = synth.genome_c2(scale=0.006)
rec = synth.gen_records_numpy(lens, 2000000, seed=8)
for level in (1,6):
    synth.write_bam('/tmp/p%d.bam'%level, names, lens, rec, procs=16, payload=True, level=level)
PY
(timeout 60 tools/ubench/wave_debug /tmp/p6.bam 3584 100000 10; echo rc=$?) > gpurun_out/r2b/d4.log 2>&1
(timeout 60 tools/ubench/wave_debug /tmp/p1.bam 3584 100000 10; echo rc=$?) > gpurun_out/r2b/d5.log 2>&1
head -12 gpurun_out/r2b/d[45].log
V6=226,258 bash tools/ubench/run_inflate_bench.sh
