#!/bin/bash
# Kernel trace of the batched matcher on the database with euroc_V1's corner counts (tools/ragged_probe.py).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_ragged
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 tools/ragged_probe.py first > $OUT/out.txt 2>&1
python3 tools/trace_top.py $OUT/trace/t_results.db | head -14
