"""bench.py end to end on the GPU box, each run in a fresh child process: the RCCL branch with one
rank (--force-dist), the self-launch of N ranks (`python bench.py --gpus 2`, no launcher) and the
strong-scaling mode (whole pair list sharded block-cyclically + device-side gather), the latter two
as a rehearsal of two ranks on the one GPU with gloo collectives."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--images", "48", "--desc", "600", "--no-extra"]


def _bench(*args, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rccl_branch_with_one_rank():
    d = _bench("--force-dist", "--steps", "1", "--warmup", "1", "--pairs", "1024", "--cpu-pairs", "64", *SMALL)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["setup"]["gathered_lists_consistent"] is True and d["setup"]["gather_last_step_ms"] > 0
    assert d["cpu_baseline"]["gpu_lists_bit_exact"] is True  # here: a sample of the GATHERED lists against the reference
    assert d["cpu_baseline"]["nproc"] >= 1 and d["cpu_baseline"]["cpu_model"]
    assert 1 <= d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["nproc"]
    ss = d["strong_scaling"]  # the collective branch also runs the whole list once
    assert ss["pairs"] == 48 * 47 // 2 and ss["gathered_lists_consistent"] is True and ss["gpu_lists_bit_exact"] is True
    assert "counters_stale" in d["roofline"]


def test_bench_starts_its_own_ranks_and_strong_scaling_adds_up():
    one = _bench("--strong", "--steps", "1", "--warmup", "1", "--cpu-pairs", "0", *SMALL)
    two = _bench("--gpus", "2", "--rehearse", "--strong", "--steps", "1", "--warmup", "1", "--cpu-pairs", "0", *SMALL)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["pairs_per_step_all_ranks"] == two["config"]["pairs_per_step_all_ranks"] == 48 * 47 // 2
    assert two["setup"]["gathered_lists_consistent"] is True
    assert one["matches_per_step"] == two["matches_per_step"] > 0  # the same lists, however they are sharded
    assert two["cpu_baseline"] is None  # --cpu-pairs 0
    checked = _bench("--gpus", "2", "--rehearse", "--strong", "--steps", "1", "--warmup", "1", "--cpu-pairs-multi", "300", *SMALL)
    assert checked["cpu_baseline"]["gpu_lists_bit_exact"] is True and "GATHERED" in checked["cpu_baseline"]["sample"]


def test_two_rank_weak_line_carries_parity_cpu_baseline_and_a_strong_scaling_step():
    """the default --gpus N line (what the driver's scaling run records) holds the
    reference against a sample of the gathered lists and times one whole-list step."""
    weak = _bench("--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1", "--pairs", "256", "--cpu-pairs-multi", "200", *SMALL)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["setup"]["gathered_lists_consistent"] is True
    assert weak["config"]["pairs_per_step_all_ranks"] == 512
    cb = weak["cpu_baseline"]
    assert cb["gpu_lists_bit_exact"] is True and cb["value"] > 0 and cb["kind"] in ("reference", "port")
    ss = weak["strong_scaling"]
    assert ss["pairs"] == 48 * 47 // 2 and ss["seconds"] > 0 and ss["pairs_per_s"] > 0 and ss["gather_ms"] >= 0
    assert ss["gathered_lists_consistent"] is True and ss["gpu_lists_bit_exact"] is True
    one = _bench("--strong", "--steps", "1", "--warmup", "1", "--cpu-pairs", "0", *SMALL)
    assert ss["matches"] == one["matches_per_step"]  # the same lists as one rank's


def test_nn2_workload_one_rank_and_two_rank_rehearsal():
    """config 4 through bench.py, queries sharded, results all-gathered."""
    one = _bench("--workload", "nn2", "--nn2-size", "20000", "--steps", "2", "--warmup", "1")
    two = _bench("--workload", "nn2", "--nn2-size", "20000", "--steps", "2", "--warmup", "1", "--gpus", "2", "--rehearse")
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["config"]["workload"] == "nn2_one_direction" and d["scaling"] == "strong"
        assert d["cpu_baseline"]["gpu_lists_bit_exact"] is True and d["cpu_baseline"]["kind"] == "port"
        assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0
    assert one["accepted_thr70_ratio1.2"] == two["accepted_thr70_ratio1.2"] > 0
