"""bench.py keeps its contract: one JSON line with the agreed keys, the roofline and cpu_baseline
objects, consistent arithmetic between value / ms_per_step / the workload size."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SIZES = [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68)]


def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the line is the DOMINANT sweep kernel's (named, with its share of the sweep time); the
    # nominal aggregate, the executed-work figure and the PMC traffic sit beside it
    for k in ("kernel", "launch_us", "alg_bytes_per_launch", "share_of_sweep_time", "survey_formula_executed", "credited_not_moved",
              "achieved_executed", "frac_executed", "traffic_source", "hbm_real_frac", "valu_frac", "active_pixel_ratio", "per_kernel"):
        assert k in rf, k
    assert "frac_nominal" not in rf and "achieved_nominal" not in rf
    assert rf["traffic_stale"] is False       # the committed profile was taken with the sweep kernels' sources as they are
    assert rf["frac_executed"] <= rf["credited_not_moved"]["frac_of_peak"] and 0 < rf["valu_frac"] < 1 and 0 < rf["active_pixel_ratio"] <= 1
    # SURVEY 8(d)'s formula on the executed rate: value x 282.7 B / 8 TB/s
    assert abs(rf["survey_formula_executed"] - d["value"] * 1e6 * 282.7 / 8e12) < 2e-5
    assert len(rf["per_kernel"]) >= 2
    for e in rf["per_kernel"]:
        for k in ("kernel", "launches", "avg_us", "alg_bytes_per_launch", "nominal_frac"):
            assert k in e, k
    dom = max(rf["per_kernel"], key=lambda e: e["share_of_sweep_time"])
    assert rf["kernel"] == dom["kernel"] and rf["frac"] == dom["nominal_frac"] and rf["launch_us"] == dom["avg_us"]
    assert abs(rf["achieved"] - rf["alg_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) < 0.02 * rf["achieved"] + 0.01
    # the headline is what RAN: executed pixel*iters / wall time; the 500-sweeps-per-level credit sits beside it
    assert "EXECUTED" in d["config"]["semantics"] and "value_executed" not in d
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * d["steps"] / d["executed_pixel_iters"] - 1.0) < 0.01
    ex = d["config"]["iters_executed_per_level_fine_to_coarse"]
    assert len(ex) == 5 and all(1 <= k <= 500 for k in ex)
    # (two different frames per run: each contributes between one sweep and 500 sweeps of every level)
    assert 2 * sum(w * h for w, h in SIZES) <= d["executed_pixel_iters"] <= 2 * 500 * sum(w * h for w, h in SIZES)
    assert d["value"] <= d["value_nominal"]
    assert abs(d["value_nominal"] * 1e6 * d["ms_per_step"] * 1e-3 / (500 * sum(w * h for w, h in SIZES)) - 1.0) < 0.01
    assert abs(sum(d["step_executed_mpix_iters"]) * 1e6 / d["executed_pixel_iters"] - 1.0) < 0.01
    assert abs(sum(e["share_of_sweep_time"] for e in rf["per_kernel"]) - 1.0) < 0.02
    assert abs(sum(e["launches"] for e in rf["per_kernel"]) - rf["launches"]) <= 0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["gpu_same_sample"]["value"] > cb["value"]        # like for like: same levels, starts, iteration counts
    # ... and the EXACT arithmetic on that sample is the oracle's result bit for bit
    assert cb["parity_same_sample"]["bit_identical"] is True and cb["parity_same_sample"]["max_abs_dv"] == 0.0
    assert d["value"] > 50 * cb["value"] / cb["cores"]      # sanity: the GPU path is not the CPU path
    # round 5: what the chip did per second whatever units are credited, the clock it held inside the chain-bound
    # kernels (in-kernel probe), and the two kinds of step apart with the frames they are
    assert d["evals_per_s"] > d["line_searches_per_s"] > 1e6 and 0 < d["valu_frac"] < 1 and d["valu_frac"] == rf["valu_frac"]
    sclk = d["sclk_mhz_observed"]
    assert 500 < sclk["k_pass"] <= 2600 and 500 < sclk["dense_tile_kernel"] <= 2600 and sclk["assumed_by_valu_peak"] == 2400
    assert d["config"]["frame_ids"] == [1, 2] and set(d["config"]["cycling_frame_ids"]) <= {1, 2}
    assert "ms_converging_steps" in d and "ms_cycling_steps" in d
    n_cyc = len(d["config"]["cycling_frame_ids"])
    assert (d["ms_cycling_steps"] is None) == (n_cyc == 0) and (d["ms_converging_steps"] is None) == (n_cyc == 2)


def test_bench_scale_reference_in_the_n1_line():
    """the N = 1 line carries config[2]'s job -- what an N > 1 run shards -- on this one GPU (here: 6 pairs)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-extras-but-scale-ref", "--scale-ref-pairs", "6"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    sr = d["scale_reference"]
    assert "config[2]" in sr["workload"] and "6 independent" in sr["workload"] and "2 stream(s) x batches of 3" in sr["workload"]
    assert 0 < sr["value"] <= sr["value_nominal"] and sr["unit"] == d["unit"]
    assert abs(sr["value"] * 1e6 * sr["ms_per_step"] * 1e-3 / sr["executed_pixel_iters"] - 1.0) < 0.01
    assert abs(sr["value_nominal"] * 1e6 * sr["ms_per_step"] * 1e-3 / (6 * 500 * sum(w * h for w, h in SIZES)) - 1.0) < 0.01
    # the dense regime's figures belong in the driver's record too (round 5)
    assert sr["evals_per_s"] > sr["line_searches_per_s"] > 1e6 and 0 < sr["valu_frac"] < 1
    assert 500 < sr["sclk_mhz_observed"]["dense_tile_kernel"] <= 2600


def test_bench_extras_of_round_5():
    """the driver's default line carries config[3] in one step, config[4] as stated (30 frames: solve with constraints,
    batched Poisson extension, render; round 6: three runs, fastest + median) with Metric 2 both ways, and the Poisson
    extension's own figures at the fixture-verified tolerances; here with the development switch that runs only those extras"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-scale-ref", "--extras", "render,poisson,pipeline30,config3"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    c3 = d["config3_4k"]
    assert "3840x2160" in c3["workload"] and len(c3["iters_executed_per_level_fine_to_coarse"]) == 6 and c3["ms_per_step"] > 0
    assert abs(c3["value"] * 1e6 * c3["ms_per_step"] * 1e-3 / c3["executed_pixel_iters"] - 1.0) < 0.01
    p30 = d["pipeline_config4_30_frames"]
    assert "30 1080p pairs" in p30["workload"] and "BCOND_BORDER" in p30["workload"]
    split = p30["compositor_split_ms_per_frame"]
    assert abs(sum(split.values()) - p30["compositor_ms_per_frame"]) < 0.1 * p30["compositor_ms_per_frame"]   # the lanes' own clocks / lanes vs the wall
    assert p30["compositor_lanes"] == 2
    assert abs(p30["solve_ms_per_pair"] + p30["compositor_ms_per_frame"] - p30["ms_per_pair"]) < 0.15 * p30["ms_per_pair"]
    lo, hi = p30["pcg_iterations_min_max"]
    assert 8 <= lo <= hi <= 40                                     # tol 1e-5: 20 iterations on the synthetic frames
    m2 = p30["render_frames_per_s"]
    assert m2["render_only"] > m2["with_poisson_amortised"] > m2["whole_pipeline_incl_solve"] > 0
    ov = p30["stages_overlapped"]       # job N's compositor beside job N + 1's solve: never slower than 1.15 x the stages in a row
    assert "error" not in ov and 0 < ov["ms_per_pair"] < 1.15 * p30["ms_per_pair"]
    assert m2["whole_pipeline_stages_overlapped"] == ov["rendered_frames_per_s"]
    pe = d["poisson_extend_1080p_ex192"]
    assert pe["tol_1e-05"]["cg_iterations"] == pe["tol_1e-05_one_side_at_a_time"]["cg_iterations"]
    assert pe["tol_1e-05"]["ms_per_frame"] < pe["tol_1e-05_one_side_at_a_time"]["ms_per_frame"]
    prf = pe["roofline"]        # the compositor's HBM-bound kernel family has its own roofline entry
    assert prf["bound"] == "hbm" and prf["unit"] == "GB/s" and abs(prf["frac"] - prf["achieved"] / prf["peak"]) < 1e-3
    assert prf["unknowns_per_side"] == 2304 * 1464 - 1918 * 1078 and 0.05 < prf["frac"] < 1.0
    # PMC traffic (committed profile x this run's iterations) is at least the algorithmic bytes of those iterations
    its = sum(pe["tol_1e-05"]["cg_iterations"])
    assert prf["traffic"] is None or prf["traffic"] > prf["unknowns_per_side"] * its * prf["alg_bytes_per_unknown_iteration"]
    assert set(k for k in pe if k.startswith("tol_")) == {"tol_1e-05", "tol_1e-06", "tol_1e-05_one_side_at_a_time", "tol_1e-05_four_frames_per_batch",
                                                           "tol_1e-06_four_frames_per_batch"}          # nothing timed at an unverified tolerance
    assert pe["tol_1e-05_four_frames_per_batch"]["ms_per_frame"] < pe["tol_1e-05"]["ms_per_frame"] and prf["four_frames_per_batch"]["frac"] > prf["frac"]
    dk = prf["dominant_kernel"]       # measured live with HIP events (vm_dbg_poisson_profile)
    # (a 4-frame batch: the PCG update rides in the level-0 restriction; VM_MGB_FUSE_MIN_SYS=0 would make it k_mgb_update)
    assert dk["kernel"] in ("k_mgb_restrict<true, true>", "k_mgb_update") and dk["launches"] >= 6 and 20 < dk["launch_us"] < 400 and 0.3 < dk["frac"] < 1.0
    assert abs(dk["frac"] - dk["achieved"] / dk["peak"]) < 1e-3 and (dk["traffic"] is None or dk["traffic"] > 0.9 * dk["alg_bytes_per_launch"])
    assert p30["second_lane_gain"] >= 1.0 and 0 <= p30["second_lane_streams_rejected"] <= 4      # the lanes are chosen by measurement
    assert p30["runs"] == 3 and len(p30["ms_per_pair_each_run"]) == 3 and p30["ms_per_pair"] <= p30["ms_per_pair_median"]


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
    total = 4 * 500 * sum(w * h for w, h in SIZES)          # all 4 pairs of the job, both ranks
    assert abs(d["value_nominal"] * 1e6 * d["ms_per_step"] * 1e-3 / total - 1.0) < 0.01
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / d["executed_pixel_iters"] - 1.0) < 0.01


def test_bench_gpus_8_gloo_one_device():
    """N = 8 readiness without the hardware: `bench.py --gpus 8` as the driver starts it (self-launch: eight
    ranks, config[2]'s 60 pairs sharded 8/7/8/7/..., two streams per rank), the two collectives over gloo,
    all eight ranks sharing this box's one device -- one JSON line, n_gpus 8, every pair of the job
    accounted for, no environment switch needed for the shared device."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VM_NO_PASS")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--config", "2", "--pairs", "60",
                        "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 1
    assert "60 independent" in d["config"]["workload"] and "rank 0: 8 pairs" in d["config"]["workload"]
    assert d["config"]["pairs_per_launch"] == 4 and d["config"]["pairs_in_flight_per_gpu"] == 2
    total = 60 * 500 * sum(w * h for w, h in SIZES)          # all 60 pairs, all eight ranks
    assert abs(d["value_nominal"] * 1e6 * d["ms_per_step"] * 1e-3 / total - 1.0) < 0.01
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 / d["executed_pixel_iters"] - 1.0) < 0.01
    assert 60 * sum(w * h for w, h in SIZES) <= d["executed_pixel_iters"] <= total


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_config4_on_two_ranks_equals_one_rank(tmp_path):
    """BASELINE config[4] under N > 1 (`for config 5: 30 pairs/4 GPUs`): `bench.py --config 4 --gpus 2` block-shards
    the pairs over two ranks (here 4 pairs of 480x270 on this box's one device, the two collectives over gloo), every rank
    runs solve -> Poisson -> nine renders for ITS pairs, the point constraints arrive inside the one broadcast.  Against
    the same job on ONE rank: every pair's halfway field is identical bit for bit (a pair's solve does not depend on its
    batch-mates or its rank), its frames are identical up to the Poisson solver's atomics (dot products accumulated by
    double-precision atomics in run-dependent order: colours within one level, most frames identical to the byte)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    docs = {}
    for n in (1, 2):
        dg = str(tmp_path / ("dig%d.json" % n))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "gloo", "--config", "4", "--pairs", "4",
                            "--size", "480x270", "--steps", "1", "--warmup", "0", "--digest", dg],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["unit"] == "frames/s" and d["scaling"] == "strong" and d["config"]["pairs"] == 4
        assert "BCOND_BORDER" in d["config"]["workload"] and "8 point constraints" in d["config"]["workload"]
        pl = d["pipeline"]
        assert [p["pairs"] for p in pl["per_rank"]] == ([4] if n == 1 else [2, 2])
        assert abs(d["value"] * d["ms_per_step"] * 1e-3 / (4 * 9) - 1.0) < 0.01          # frames/s x s per step = the job's 36 frames
        assert pl["solve_mpix_iters_per_s"] > 0 and 5 <= pl["rank0"]["pcg_iterations_min_max"][0] <= pl["rank0"]["pcg_iterations_min_max"][1] <= 40
        assert "160 bytes" in d["config"]["collectives"]            # 8 constraints x 5 floats ride in the broadcast
        files = [dg] if n == 1 else [dg + ".0", dg + ".1"]
        merged = {"fields": {}, "frames": {}, "frame_bytes": {}}
        for f in files:
            doc = json.load(open(f))
            for k in merged:
                merged[k].update(doc[k])
        docs[n] = merged
    assert sorted(docs[1]["fields"]) == sorted(docs[2]["fields"]) == ["0", "1", "2", "3"]
    assert docs[1]["fields"] == docs[2]["fields"]                     # halfway fields: bit-identical
    assert sorted(docs[1]["frames"]) == sorted(docs[2]["frames"]) and len(docs[1]["frames"]) == 36
    same = sum(docs[1]["frames"][k] == docs[2]["frames"][k] for k in docs[1]["frames"])
    assert same >= 4, same             # (observed: most of the 36; the bound only says the frames are the same frames)
    import numpy as np
    for k in docs[1]["frame_bytes"]:
        a, b = np.asarray(docs[1]["frame_bytes"][k], int), np.asarray(docs[2]["frame_bytes"][k], int)
        assert np.abs(a - b).max() <= 1
