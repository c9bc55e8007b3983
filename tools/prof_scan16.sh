#!/bin/bash
# Counters of the scan harness (tools/bin/mfma_scan_prod): the shipped 32x32x64 kernel against the 16x16x128 prototypes,
# per kernel: duration, clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA pipe busy, VALU and MFMA instructions.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
N=${1:-2048}
OUT=gpurun_out/r03/prof_scan16
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT/pmc -o p -- tools/bin/mfma_scan_prod $N 8192 83 4 > $OUT/out.txt 2>&1
python3 - <<'PY'
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('gpurun_out/r03/prof_scan16/pmc/*results.db')[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cc = [t for t in tabs if t.startswith('counters_collection')][0]
kd = [t for t in tabs if t == 'kernels' or t.startswith('kernels')][0]
cols = [r[1] for r in db.execute(f"pragma table_info({cc})")]
print(cols)
rows = db.execute(f"select kernel_name, counter_name, dispatch_id, value, start, end from {cc}").fetchall() if 'start' in cols else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
if rows:
    for k, c, d, v, s, e in rows:
        name = k.split('(')[0][-60:]
        acc[name][c].append(v)
        dur[name].append((d, e - s))
for name, cs in acc.items():
    if 'scan' not in name: continue
    n = len(cs['GRBM_GUI_ACTIVE'])
    ds = dict(dur[name])
    t = sum(ds.values()) / len(ds) * 1e-9
    g = sum(cs['GRBM_GUI_ACTIVE']) / n
    f = lambda c: sum(cs[c]) / n
    clk = g / 8 / t / 1e9
    print(f"{name:62s} n={n} {t*1e3:7.3f} ms clock {clk:5.3f} GHz mfma_busy {f('SQ_VALU_MFMA_BUSY_CYCLES')/1024/(g/8):.3f} valu/mfma {f('SQ_INSTS_VALU')/max(f('SQ_INSTS_MFMA'),1):.2f} "
          f"mfma {f('SQ_INSTS_MFMA')/1e6:.1f} M wait_any/wave_cycles {f('SQ_WAIT_INST_ANY')/f('SQ_WAVE_CYCLES'):.3f} active_valu/busy {f('SQ_ACTIVE_INST_VALU')/f('SQ_BUSY_CYCLES'):.3f}")
PY
