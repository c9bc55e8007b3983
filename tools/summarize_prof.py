#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd databases written by tools/profile_bench.sh into the
text summaries committed under profiles/.  usage: summarize_prof.py <prof dir> <out dir>"""
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()


with open(os.path.join(dst, "kernel_trace_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-extra --cpu-pairs 0 --pairs 16384\n")
    f.write("# columns: kernel | calls | total_us | avg_us | percent\n")
    for name, calls, total, avg, pct in q(os.path.join(src, "trace", "t_results.db"), "select * from top_kernels"):
        f.write(f"{name.split('(')[0]:60s} {calls:6d} {total:14.1f} {avg:12.1f} {pct:8.3f}\n")

ex = os.path.join(src, "trace_extras", "t_results.db")
if os.path.exists(ex):
    with open(os.path.join(dst, "kernel_trace_stats_extras.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --pairs 4096\n")
        f.write("# (with the extras: stereo pair, find_matches_landmarks frame, ORB frame, config-3 shape, dense overlap, 1M x 1M)\n")
        f.write("# columns: kernel | calls | total_us | avg_us | percent\n")
        for name, calls, total, avg, pct in q(ex, "select * from top_kernels"):
            f.write(f"{name.split('(')[0]:60s} {calls:6d} {total:14.1f} {avg:12.1f} {pct:8.3f}\n")

with open(os.path.join(dst, "pmc_counters.txt"), "w") as f:
    f.write("# rocprofv3 --pmc <counters> (one run per group) -- same command; per-dispatch averages\n")
    f.write("# columns: kernel | counter | dispatches | avg value per dispatch\n")
    for d in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_mfma"):
        p = os.path.join(src, d, "p_results.db")
        if not os.path.exists(p):
            continue
        for name, ctr, n, avg in q(p, "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                      "group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write(f"{name.split('(')[0]:50s} {ctr:22s} {n:4d} {avg:20.1f}\n")
print(open(os.path.join(dst, "kernel_trace_stats.txt")).read())
print(open(os.path.join(dst, "pmc_counters.txt")).read())

# HBM traffic of the dominant kernel per image pair, for bench.py's roofline.traffic
import json
vals = {}
# the dominant kernel = the scan kernel with the largest summed FETCH_SIZE; WRITE_SIZE of the same kernel
r = q(os.path.join(src, "pmc_fetch", "p_results.db"),
      "select kernel_name, avg(value) from counters_collection where counter_name='FETCH_SIZE' and "
      "kernel_name like '%scan_kernel%' group by kernel_name order by sum(value) desc")
full_name, vals["FETCH_SIZE"] = r[0]
kname = full_name.split("(")[0].replace("void vsm::", "")
r = q(os.path.join(src, "pmc_write", "p_results.db"),
      "select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' group by kernel_name")
vals["WRITE_SIZE"] = [v for n, v in r if n == full_name][0]
pairs = 16384
json.dump({"kernel": kname, "pairs_per_launch": pairs,
           "FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"],
           "bytes_per_pair": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / pairs,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_bench.sh); raw counter "
                   "values in KiB. bytes_per_pair = (2 x FETCH_SIZE + WRITE_SIZE) / pairs: the x2 gfx950 correction of "
                   "MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of 16 B/lane global loads) applies, the "
                   "matrix-core scan reads every descriptor through global_load_dwordx4. WRITE_SIZE is exact "
                   "(8 B per query)."},
          open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
