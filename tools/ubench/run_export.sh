mkdir -p gpurun_out/exp; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/exp
(timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -4)
(PD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 6 --warmup 2 --e2e-records 0 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err); tail -1 $O/bench_dist1.err
python3 -c "
import json; d=json.load(open('$O/bench_dist1.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:300]); print(json.dumps(d['kernels']['direct_export']))"
(timeout 300 python bench.py --config w100a > $O/bench_w100a.json 2> $O/w.err); python3 -c "
import json; d=json.load(open('$O/bench_w100a.json')); print(d['value'], d['ms_per_step'])"
