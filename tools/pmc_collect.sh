#!/bin/bash
# tools/pmc_collect.sh — HBM traffic of this repo's kernels from rocprofv3 PMC counters, one
# counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with tracing).
# Run on the GPU box from the repo root; writes gpurun_out/pmc/{fetch,write}_pdk.csv.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o run -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py ${1:-1e9} > $OUT/$C.log 2>&1 || true
  F=$(find $OUT/$C -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then head -1 "$F" > $OUT/${C}_pdk.csv; grep "pdk::" "$F" >> $OUT/${C}_pdk.csv || true; fi
  rm -rf $OUT/$C
done
ls -la $OUT; head -3 $OUT/FETCH_SIZE_pdk.csv | cut -c1-400
