"""The committed bench line of this round (profiles/r02/bench_round2_final.json, written by
`python bench.py` on an MI355X) carries every field of the driver's contract and the ones
VERDICT r1 asked for (real_like with its own roofline, sustained clock, MFMA busy fraction, the
host's core count and CPU model)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("rel", ["r02/bench_round2_final.json", "r03/bench_round3_final.json"])
def test_committed_bench_line_has_the_contract_fields(rel):
    with open(os.path.join(ROOT, "profiles", *rel.split("/"))) as f:
        d = json.loads(f.readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["pairs_per_rank_per_step"] == 32768
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "sustained_clock_ghz", "mfma_busy_frac"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.6 < r["frac"] < 1.0
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 9000 < r["peak"] < 11000  # dense FP4 peak
    assert r["pairs_per_launch"] == 32768 and 1.5 < r["sustained_clock_ghz"] < 2.4 and 0.7 < r["mfma_busy_frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] < r["hbm_algorithmic_bytes_per_pair"] * r["pairs_per_launch"]
    # the popcount formulation of north_star, measured in the same run: lists identical, its own roofline
    p = d["popcount_path"]
    assert p["lists_identical_to_matrix_core_path"] is True and p["roofline"]["bound"] == "valu"
    assert 0.7 < p["roofline"]["frac"] < 1.0 and d["pairs_per_s"] > 3 * p["pairs_per_s"]
    # descriptors with the statistics of real ORB output: the same scan at the same rate
    x = d["real_like"]
    assert x["sample_of_24_pairs_equals_cpu_reference"] is True
    assert x["matrix_core"]["forward_gcomps"] >= 12000 and 0.6 < x["roofline"]["frac"] < 1.0
    assert abs(x["roofline"]["frac"] - x["roofline"]["achieved"] / x["roofline"]["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample", "nproc", "cpu_model"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["gpu_lists_bit_exact"] is True
    assert d["value"] > 100 * c["value"]
    assert d["extra"]["stereo_1500x1500"]["latency_ms_host_to_host"] < 0.040
    if rel.startswith("r03"):  # the counters belong to the running code; which kernel the line is about
        assert r["counters_stale"] is False and "frac_of_peak_at_sustained_clock" in r
        assert "16x16x128" in r["kernel"] and r["frac"] > 0.76
        assert d["extra"]["nn2_1Mx1M"]["roofline_frac"] > 0.8
        x = d["real_like"]["matrix_core"]  # consumer kernels: the step outside the scans
        assert x["total_ms"] < 1.10 * (x["forward_scan_ms"] + x["backward_scan_ms"])
