mkdir -p gpurun_out/legs; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/legs
(timeout 300 python bench.py --config gff > $O/bench_gff.json 2> $O/g.err; echo "rc=$?" >> $O/g.err)
(timeout 300 python bench.py --config w100a > $O/bench_w100a.json 2> $O/w.err; echo "rc=$?" >> $O/w.err)
tail -1 $O/g.err $O/w.err; wc -l $O/*.json
python3 -c "
import json
for f in ('bench_gff','bench_w100a'):
    d=json.load(open('$O/%s.json'%f)); print(f, '%.3g'%d['value'], round(d['ms_per_step'],2), json.dumps(d['roofline'])[:260]); print('  ', json.dumps(d['kernels']['scatter_tiles']))"
