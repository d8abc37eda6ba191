#!/bin/bash
# round 4: SQ counters + kernel trace of the ISOLATED wave inflate kernel (pd_x_bgzf_inflate through tools/bgzf_gpu_bench.py,
# variant 258 = wave per member, 16 waves per CU, CRC check on) — one rocprofv3 --pmc pass per counter set, no tracing with them
# usage: r4_inflate_sq.sh <out dir under gpurun_out> [records] [variant: 258 = 16 waves per CU (round 3), 322 = 20 (round 4)]
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-inflate_sq}; R=${2:-2e6}
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
export BGZF_VARIANTS=${3:-258}
timeout 200 python $GRAFT_REPO_ROOT/tools/bgzf_gpu_bench.py $R > $O/bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o run -- python $GRAFT_REPO_ROOT/tools/bgzf_gpu_bench.py $R > $O/kt.log 2>&1
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $O/kernel_stats.csv; rm -rf $O/kt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/p$i -o run -- python $GRAFT_REPO_ROOT/tools/bgzf_gpu_bench.py $R > $O/p$i.log 2>&1 || true
  F=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then head -1 "$F" > $O/sq_pass$i.csv; grep "inflate" "$F" >> $O/sq_pass$i.csv || true; fi
  rm -rf $O/p$i
done
ls -la $O
