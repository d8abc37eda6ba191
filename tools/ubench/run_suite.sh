# GPU box: the whole -m gpu suite, then bench.py (short) with the e2e leg
mkdir -p gpurun_out/r2f; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r2f/pytest.log
true
tail -5 gpurun_out/r2f/pytest.log; tail -3 gpurun_out/r2f/bench.err; cat gpurun_out/r2f/bench.json | cut -c1-1500
