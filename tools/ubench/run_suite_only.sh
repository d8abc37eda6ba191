mkdir -p gpurun_out/r2s; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 1700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -25) > gpurun_out/r2s/pytest.log
tail -6 gpurun_out/r2s/pytest.log | cut -c1-400
