# GPU box: the whole -m gpu suite, then the round's bench lines
mkdir -p gpurun_out/r2s; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r2s/pytest.log
tail -6 gpurun_out/r2s/pytest.log | cut -c1-400
O=gpurun_out/r2s
(timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
(PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err)
tail -2 $O/bench.err; wc -l $O/bench.json $O/bench_dist1.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])); print(json.dumps(d['e2e'])[-700:]); print(json.dumps(d['cpu_baseline'])[:200])"
