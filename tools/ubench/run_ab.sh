mkdir -p gpurun_out/ab; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export VARIANTS=0,0/s128,0/s64,0/l160,0/s64/l160,0/s32/l160,0
(timeout 500 python tools/ubench/direct_ab.py > gpurun_out/ab/ab.log 2>&1); tail -6 gpurun_out/ab/ab.log | cut -c1-400
