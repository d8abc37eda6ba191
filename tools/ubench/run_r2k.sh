mkdir -p gpurun_out/r2k; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 400 python tools/ubench/dbg_direct_narrow.py > gpurun_out/r2k/narrow.log 2>&1)
tail -40 gpurun_out/r2k/narrow.log | cut -c1-300
