mkdir -p gpurun_out/r2g; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 400 python bench.py --steps 10 --warmup 3 --e2e-records 0 > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "rc=$?" >> gpurun_out/r2g/bench.err)
(PANDEPTH_TIMING=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "direct or gff or w100" 2>&1 | tail -25) > gpurun_out/r2g/pytest.log
tail -3 gpurun_out/r2g/bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r2g/bench.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']), json.dumps(d['arrays_path']))"
tail -22 gpurun_out/r2g/pytest.log
