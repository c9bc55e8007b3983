#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_dense
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc -o p -- python3 tools/dense_sweep.py > $OUT/out.txt 2>&1
python3 - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_dense/pmc/p_results.db')
rows=db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
for r in rows:
    if 'scan_kernel' in r[0]: print(r[0].split('(')[0], r[1], r[2], r[3])
PY
