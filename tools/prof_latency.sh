#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_lat
rm -rf $OUT; mkdir -p $OUT
python3 tools/latency_probe.py 1500 0x401 0x801 0x1001 0x2001 0x404 0x804 2>&1 | grep -v amdgpu.ids
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/t -o t -- python3 tools/latency_probe.py 1500 > $OUT/out.txt 2>&1
python3 - <<'PY'
import sqlite3
db=sqlite3.connect('gpurun_out/prof_lat/t/t_results.db')
rows=db.execute("select name, start, end from kernels order by start").fetchall()
# last 40 kernel launches: durations and gaps
rows=rows[-24:]
prev=None
for name,s,e in rows:
    gap = (s-prev)/1e3 if prev else 0
    print(f"{name.split('(')[0][:40]:40s} dur {(e-s)/1e3:8.2f} us  gap_before {gap:8.2f} us")
    prev=e
PY
