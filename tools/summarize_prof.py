#!/usr/bin/env python3
"""Turns the rocprofv3 rocpd databases written by tools/profile_bench.sh into the
text summaries committed under profiles/ and into pmc_summary.json, which bench.py reads for
roofline.traffic / sustained_clock_ghz / mfma_busy_frac.  usage: summarize_prof.py <prof dir> <out dir>"""
import hashlib
import json
import os
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
XCDS, SIMDS = 8, 1024


def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()


cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "bench.py"
line = json.loads(open(os.path.join(src, "trace.json")).readline())
pairs = int(line["config"]["pairs_per_rank_per_step"])
top = q(os.path.join(src, "trace", "t_results.db"), "select * from top_kernels")
# the forward scan of the batch: the 16x16x128 kernel (csrc/mfma16_scan.h) or the 32x32x64 one, whichever the library chose
DOMINANT = max(("mfma16_scan_kernel", "mfma_scan_kernel<8, 1"), key=lambda k: sum(t[2] for t in top if k in t[0]))
with open(os.path.join(dst, "kernel_trace_stats.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 {cmd}   ({pairs} pairs per launch, the bench default)\n")
    f.write("# columns: kernel | calls | total_us | avg_us | percent\n")
    for name, calls, total, avg, pct in top:
        f.write(f"{name.split('(')[0]:60s} {calls:6d} {total:14.1f} {avg:12.1f} {pct:8.3f}\n")
    r = line["roofline"]
    # --stats averages every launch; bench.py's HIP events average the timed ones only (the first
    # `warmup` launches run at a lower clock: same cycle count, see pmc_counters.txt)
    durs = [(e - b) / 1e3 for b, e in q(os.path.join(src, "trace", "t_results.db"),
            f"select start, end from kernels where name like '%{DOMINANT}%' order by start")]
    warm = int(line["warmup"]) + int(line.get("setup", {}).get("clock_settle_calls_before_warmup", 0))  # every untimed launch
    timed_us = sum(durs[warm:]) / max(1, len(durs[warm:]))
    f.write(f"# {DOMINANT}, every launch in us: {' '.join(f'{d:.1f}' for d in durs)}  (the first {warm} = set-up and warm-up, untimed by bench.py)\n")
    f.write(f"# average of the {len(durs) - warm} timed launches: {timed_us:.1f} us by the kernel trace; the bench line of this same "
            f"(profiled) run: avg_launch_ms {r['avg_launch_ms']:.3f} by HIP events, kernel {r['kernel_gcomps']:.0f} G comparisons/s, "
            f"roofline.frac {r['frac']:.4f}\n")

ex = os.path.join(src, "trace_extras", "t_results.db")
if os.path.exists(ex):
    with open(os.path.join(dst, "kernel_trace_stats_extras.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --pairs 4096\n")
        f.write("# (with the extras: stereo pair, find_matches_landmarks frame, ORB frame, config-3 shape, dense overlap, real-like, 1M x 1M)\n")
        f.write("# columns: kernel | calls | total_us | avg_us | percent\n")
        for name, calls, total, avg, pct in q(ex, "select * from top_kernels"):
            f.write(f"{name.split('(')[0]:60s} {calls:6d} {total:14.1f} {avg:12.1f} {pct:8.3f}\n")

ctr = {}
with open(os.path.join(dst, "pmc_counters.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc <counters> (one run per group) -- python3 {cmd}; per-dispatch averages\n")
    f.write("# columns: kernel | counter | dispatches | avg value per dispatch\n")
    for d in ("pmc_mfma", "pmc_sq", "pmc_fetch", "pmc_write"):
        p = os.path.join(src, d, "p_results.db")
        if not os.path.exists(p):
            continue
        for name, c, n, avg in q(p, "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                    "group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write(f"{name.split('(')[0]:50s} {c:26s} {n:4d} {avg:20.1f}\n")
        # the dominant kernel: timed launches only, with the duration of each launch IN THIS PASS
        rows = q(p, f"select dispatch_id, counter_name, value, end - start from counters_collection where kernel_name like '%{DOMINANT}%' "
                    "order by dispatch_id")
        ids = sorted({r[0] for r in rows})[warm:]
        for c in {r[1] for r in rows}:
            vals = [r[2] for r in rows if r[1] == c and r[0] in ids]
            ctr[(d, c)] = sum(vals) / len(vals)
        if d == "pmc_mfma":
            per = {i: {r[1]: (r[2], r[3]) for r in rows if r[0] == i} for i in ids}
            f.write(f"# {DOMINANT} in the pmc_mfma pass, launch by launch (all {len(ids) + warm}, the first {warm} = warm-up): duration us | GRBM_GUI_ACTIVE / 8 XCDs | clock GHz\n")
            for i in sorted({r[0] for r in rows}):
                g_, ns = [(r[2], r[3]) for r in rows if r[0] == i and r[1] == "GRBM_GUI_ACTIVE"][0]
                f.write(f"#   {ns / 1e3:10.1f} {g_ / XCDS:14.0f} {g_ / XCDS / ns:8.3f}\n")
            clk = [per[i]["GRBM_GUI_ACTIVE"][0] / XCDS / per[i]["GRBM_GUI_ACTIVE"][1] for i in ids]
            ctr[("pmc_mfma", "clock_ghz")] = sum(clk) / len(clk)

avg_us = [avg for name, calls, total, avg, pct in top if DOMINANT in name][0]
s = {"kernel": DOMINANT, "command": cmd, "pairs_per_launch": pairs, "avg_launch_us_kernel_trace": timed_us,
     "avg_launch_us_kernel_trace_with_warmup_launches": avg_us,
     "avg_launch_ms_hip_events_same_run": line["roofline"]["avg_launch_ms"]}
if ("pmc_fetch", "FETCH_SIZE") in ctr and ("pmc_write", "WRITE_SIZE") in ctr:
    s["FETCH_SIZE_KB_per_launch"] = ctr[("pmc_fetch", "FETCH_SIZE")]
    s["WRITE_SIZE_KB_per_launch"] = ctr[("pmc_write", "WRITE_SIZE")]
    s["hbm_bytes_per_pair"] = (2 * s["FETCH_SIZE_KB_per_launch"] + s["WRITE_SIZE_KB_per_launch"]) * 1024 / pairs
g = ctr.get(("pmc_mfma", "GRBM_GUI_ACTIVE"))
b = ctr.get(("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES"))
if g and b:
    s["GRBM_GUI_ACTIVE_per_launch"] = g
    s["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] = b
    s["SQ_INSTS_MFMA_per_launch"] = ctr.get(("pmc_mfma", "SQ_INSTS_MFMA"))
    s["sustained_clock_ghz"] = ctr[("pmc_mfma", "clock_ghz")]
    s["mfma_busy_frac"] = (b / SIMDS) / (g / XCDS)
v = ctr.get(("pmc_sq", "SQ_INSTS_VALU"))
if v and s.get("SQ_INSTS_MFMA_per_launch"):
    s["valu_instructions_per_mfma"] = v / s["SQ_INSTS_MFMA_per_launch"]
# what code the counters were taken from: bench.py marks them stale when the sources have changed since
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for name in ("mfma_scan.h", "mfma16_scan.h", "vsmatch_kernels.h", "vsmatch.hip"):
    with open(os.path.join(ROOT, "visual-slam_amd", "csrc", name), "rb") as f:
        h.update(f.read())
s["kernel_sources_sha256"] = h.hexdigest()
s["kernel_sources"] = "visual-slam_amd/csrc/{mfma_scan.h, mfma16_scan.h, vsmatch_kernels.h, vsmatch.hip}, concatenated in this order"

# the popcount formulation (north_star's: v_xor + v_bcnt, knob bit 25) under the same command: VALU utilisation evidence
pc = {}
POP = "scan_kernel_adaptive"
for d in ("pop_trace", "pop_sq", "pop_fetch", "pop_write"):
    p = os.path.join(src, d, "p_results.db" if d != "pop_trace" else "t_results.db")
    if not os.path.exists(p):
        continue
    if d == "pop_trace":
        rows = q(p, f"select start, end from kernels where name like '%{POP}%' order by start")
        durs_p = [(e - b) / 1e3 for b, e in rows][warm:]
        if durs_p:
            pc["avg_launch_us_kernel_trace"] = sum(durs_p) / len(durs_p)
        continue
    rows = q(p, f"select dispatch_id, counter_name, value, end - start from counters_collection where kernel_name like '%{POP}%' order by dispatch_id")
    ids = sorted({r[0] for r in rows})[warm:]
    for c in {r[1] for r in rows}:
        vals = [r[2] for r in rows if r[1] == c and r[0] in ids]
        if vals:
            pc[c] = sum(vals) / len(vals)
    if d == "pop_sq" and rows:
        clk = [r[2] / XCDS / r[3] for r in rows if r[1] == "GRBM_GUI_ACTIVE" and r[0] in ids]
        if clk:
            pc["sustained_clock_ghz"] = sum(clk) / len(clk)
if pc:
    if "SQ_ACTIVE_INST_VALU" in pc and "GRBM_GUI_ACTIVE" in pc:
        # SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md), here one per VALU instruction (= SQ_INSTS_VALU):
        # x4 / SIMDs / cycles per XCD = nominal four-cycle VALU passes per SIMD cycle.  Above 1: gfx950 issues the
        # scan's v_xor / v_bcnt in 2.3-4.6 cycles per wave64 instruction, not 4 (profiles/r01/valu_rate2_set1_opcodes.txt)
        # -- the utilisation against the MEASURED issue rate is bench.py's popcount_path.roofline.frac (0.90).
        pc["valu_nominal_passes_per_simd_cycle"] = pc["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (pc["GRBM_GUI_ACTIVE"] / XCDS)
        pc["valu_instructions_per_simd_cycle"] = pc["SQ_INSTS_VALU"] / SIMDS / (pc["GRBM_GUI_ACTIVE"] / XCDS)
    if "FETCH_SIZE" in pc and "WRITE_SIZE" in pc:
        pc["hbm_bytes_per_pair"] = (2 * pc["FETCH_SIZE"] + pc["WRITE_SIZE"]) * 1024 / pairs
    if "SQ_INSTS_VALU" in pc:
        pc["valu_instructions_per_wave_comparison"] = pc["SQ_INSTS_VALU"] / (pairs * float(line["config"]["desc_per_image"]) ** 2 / 64)
    pc["command"] = cmd + " --variant 0x2000000"
    s["popcount_path"] = pc

s["note"] = ("rocprofv3 --pmc, one pass per counter group (tools/profile_bench.sh); raw FETCH_SIZE / WRITE_SIZE in KiB. "
             "hbm_bytes_per_pair = (2 x FETCH_SIZE + WRITE_SIZE) / pairs: the x2 gfx950 correction of MI355X_MICROARCH.md "
             "(FETCH_SIZE reports half the bytes of 16 B/lane global loads) applies, the scan reads every descriptor through "
             "global_load_dwordx4; WRITE_SIZE is exact (8 B per query).  Timed launches only (bench.py's warm-up launches run the same cycle count at a lower clock).  "
             "sustained_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / the duration of the same dispatch in the counter pass; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs).")
json.dump(s, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(open(os.path.join(dst, "kernel_trace_stats.txt")).read())
print(json.dumps(s, indent=1))
