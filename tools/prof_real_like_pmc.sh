#!/bin/bash
# PMC counters of the consumer kernels on a real-like database (tools/real_like_probe.py 200 2000).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_rl_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc -o p -- python3 tools/real_like_probe.py 200 2000 > $OUT/out.txt 2>&1
python3 - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_rl_pmc/pmc/p_results.db')
rows=db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
for r in rows:
    if 'nominate' in r[0] or 'count_list' in r[0]: print(r[0].split('(')[0][:40], r[1], r[2], round(r[3]))
PY
