"""bench.py keeps its contract: one JSON line with the agreed keys, the roofline and cpu_baseline
objects, consistent arithmetic between value / ms_per_step / the workload size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def total_credit(d):
    return d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * d["steps"] * 1.01


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the line is the DOMINANT sweep kernel's (named, with its share of the sweep time); the
    # nominal aggregate, the executed-work figure and the PMC traffic sit beside it
    for k in ("kernel", "launch_us", "alg_bytes_per_launch", "share_of_sweep_time", "frac_nominal", "achieved_executed",
              "frac_executed", "traffic_source", "hbm_real_frac", "valu_frac", "active_pixel_ratio", "per_kernel"):
        assert k in rf, k
    assert rf["frac_executed"] <= rf["frac_nominal"] and 0 < rf["valu_frac"] < 1 and 0 < rf["active_pixel_ratio"] <= 1
    assert len(rf["per_kernel"]) >= 2
    for e in rf["per_kernel"]:
        for k in ("kernel", "launches", "avg_us", "alg_bytes_per_launch", "nominal_frac"):
            assert k in e, k
    dom = max(rf["per_kernel"], key=lambda e: e["share_of_sweep_time"])
    assert rf["kernel"] == dom["kernel"] and rf["frac"] == dom["nominal_frac"] and rf["launch_us"] == dom["avg_us"]
    assert abs(rf["achieved"] - rf["alg_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) < 0.02 * rf["achieved"] + 0.01
    # what ran vs what is credited: fixed work credits 500 sweeps per level, executed counts the
    # sweeps up to each level's convergence
    assert "skips" in d["config"]["semantics"]
    assert 0 < d["executed_pixel_iters"] <= total_credit(d) and 0 < d["value_executed"] <= d["value"]
    assert d["config"]["iters_executed_per_level_fine_to_coarse"][-1] <= 500
    assert abs(sum(e["share_of_sweep_time"] for e in rf["per_kernel"]) - 1.0) < 0.02
    assert abs(sum(e["launches"] for e in rf["per_kernel"]) - rf["launches"]) <= 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["gpu_same_sample"]["value"] > cb["value"]        # like for like: same levels, starts, iteration counts
    # ... and the EXACT arithmetic on that sample is the oracle's result bit for bit
    assert cb["parity_same_sample"]["bit_identical"] is True and cb["parity_same_sample"]["max_abs_dv"] == 0.0
    # value = pixel-iterations of one fixed-work 1080p solve / time of the step
    sizes = [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]
    total = 500 * sum(w * h for w, h in sizes)
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / total - 1.0) < 0.01
    assert d["value"] > 50 * cb["value"] / cb["cores"]      # sanity: the GPU path is not the CPU path


def test_bench_gpus_2_self_launches_two_ranks():
    """`python bench.py --gpus 2` (no torch.distributed.run in front, the driver's form) must
    run TWO ranks: config[2]'s pairs sharded over them, each rank one vm_solve_batch per step,
    n_gpus == 2 in the line.  This box has one GPU, so the two collectives go over gloo and the
    ranks share the device (--backend gloo); with RCCL the same code path runs one rank per GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "4", "--inflight", "1",
                        "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert "config[2]" in d["config"]["workload"] and d["config"]["pairs_per_launch"] == 2
    sizes = [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]
    total = 4 * 500 * sum(w * h for w, h in sizes)          # all 4 pairs of the job, both ranks
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / total - 1.0) < 0.01


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
