"""The committed bench line of this round (profiles/r01/bench_round1_final.json, written by
`python bench.py` on an MI355X) carries every field of the driver's contract."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r01", "bench_round1_final.json")) as f:
        d = json.loads(f.readline())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.5 < r["frac"] < 1.0
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 9000 < r["peak"] < 11000  # dense FP4 peak
    # the popcount formulation of north_star, measured in the same run: lists identical, its own roofline
    p = d["popcount_path"]
    assert p["lists_identical_to_matrix_core_path"] is True and p["roofline"]["bound"] == "valu"
    assert 0.7 < p["roofline"]["frac"] < 1.0 and d["pairs_per_s"] > 3 * p["pairs_per_s"]
    x = d["extra"]["real_orb_statistics_200img_19900pairs"]
    assert x["sample_of_24_pairs_equals_cpu_reference"] is True and x["chosen"]["tracking_form"] == "distances"
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["gpu_lists_bit_exact"] is True
    assert d["value"] > 100 * c["value"]
