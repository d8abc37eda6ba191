#!/bin/bash
# round 5, call 9: what a 1-rank RCCL communicator costs in a fresh process, under a few environments (call 8 lost the lines to a `tail`)
O=$GRAFT_REPO_ROOT/gpurun_out/r5c9; mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { ( cd /tmp && env "$@" timeout 120 python $GRAFT_REPO_ROOT/tools/ubench/comm_init_time.py > $O/out_$1.txt 2>&1; grep -h "comm init" $O/out_$1.txt >> $O/comm.txt ); }
run TAG=default
run TAG=default_again
run TAG=ib_disabled NCCL_IB_DISABLE=1
run TAG=socket_lo NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo NCCL_NET=Socket
run TAG=no_plugin NCCL_IB_DISABLE=1 NCCL_SOCKET_IFNAME=lo NCCL_NET_PLUGIN=none RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0
run TAG=debug_info NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV,NET
cat $O/comm.txt | cut -c1-300
grep -c . $O/out_TAG=debug_info.txt; grep -E "NCCL INFO" $O/out_TAG=debug_info.txt | cut -c1-200 | head -60
