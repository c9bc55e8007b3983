#!/usr/bin/env python3
"""The last N device events (kernels and copies) of a rocprofv3 --kernel-trace --memory-copy-trace database
in time order, with the gap to the one before: where a latency-bound call spends its time.
usage: trace_timeline.py <t_results.db> [N=40]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ev = [(s, e, name.split("(")[0][:48]) for s, e, name in db.execute("select start, end, name from kernels")]
ev += [(s, e, f"copy {name} {size} B") for s, e, name, size in db.execute("select start, end, name, size from memory_copies")]
ev.sort()
ev = ev[-n:]
t0, prev = ev[0][0], ev[0][0]
for s, e, name in ev:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(s - prev) / 1e3:7.1f} gap  {(e - s) / 1e3:8.1f} us  {name}")
    prev = e
