mkdir -p gpurun_out/abpmc; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VARIANTS=0,3504
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/abpmc/p1 -- python tools/ubench/direct_ab.py > gpurun_out/abpmc/p1.log 2>&1
tail -3 gpurun_out/abpmc/p1.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/abpmc/p1/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if 'direct' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    for k, d in agg.items():
        print(k)
        for c, v in sorted(d.items()): print('   %-24s %.4g' % (c, v))
PY
find gpurun_out/abpmc -name "*.csv" -size +2M -delete
