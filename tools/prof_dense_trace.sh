#!/bin/bash
# Kernel trace of the batched matcher on the dense-overlap batch (tools/dense_sweep.py: 96 images x 2000, all 4560 pairs).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_dense_trace
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 tools/dense_sweep.py > $OUT/out.txt 2>&1
python3 tools/trace_top.py $OUT/trace/t_results.db | head -20
cat $OUT/out.txt | tail -6
