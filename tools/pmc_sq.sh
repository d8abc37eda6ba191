#!/bin/bash
# tools/pmc_sq.sh — SQ counters of the direct window kernel, one rocprofv3 --pmc pass per counter set (8 SQ slots)
# usage: pmc_sq.sh <records> <direct_un> ["COUNTERS ..." ...]
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
R=${1:-1e9}; UNV=${2:-504}; shift 2 || true
if [ $# -eq 0 ]; then
  set -- "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM"
fi
i=0
for C in "$@"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o run -- python $GRAFT_REPO_ROOT/tools/pmc_direct.py $R $UNV > $OUT/p$i.log 2>&1 || true
  F=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then head -1 "$F" > $OUT/sq_pass$i.csv; grep "k_direct_tiles\|k_index" "$F" >> $OUT/sq_pass$i.csv || true; fi
  rm -rf $OUT/p$i
done
ls $OUT
