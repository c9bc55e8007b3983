#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace --stats database.  usage: trace_top.py <t_results.db>"""
import sqlite3, sys
for name, calls, total, avg, pct in sqlite3.connect(sys.argv[1]).execute("select * from top_kernels").fetchall():
    print(f"{name.split('(')[0][:70]:70s} {calls:6d} {total:12.1f} us total {avg:10.2f} us avg {pct:7.2f} %")
