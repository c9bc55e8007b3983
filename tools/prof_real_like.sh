#!/bin/bash
# Kernel trace of the batched matcher on a real-like database (tools/real_like_probe.py 200 2000).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_rl
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 tools/real_like_probe.py 200 2000 > $OUT/out.txt 2>&1
python3 tools/trace_top.py $OUT/trace/t_results.db | head -20
