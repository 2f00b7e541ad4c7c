#!/bin/bash
# r05 second GPU visit: the batched ring-only Poisson solver -- tests, A/B timing against round 4's solver, counters
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "poisson or pipeline or chaos_floor_of_config1 or inside_the_family" > gpurun_out/pytest_r05b_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r05b_new.log
tail -15 gpurun_out/pytest_r05b_new.log
VM_POISSON_SOLVER=mg1 timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05b_mg1.txt 2>&1; cat gpurun_out/poisson_r05b_mg1.txt
timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05b_mgb.txt 2>&1; cat gpurun_out/poisson_r05b_mgb.txt
bash tools/prof_pmc.sh r05b_compositor "k_" tools/prof_compositor.py
python - > gpurun_out/sclk_r05b.txt 2>&1 <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
prm = morph.Parameters(); prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
ctx.set_params(morph.KernParameters(prm))
i0, i1 = synth.make_pair(1920, 1080)
for rep in range(3):
    pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
    prog = (capi.Progress * 5)()
    capi.check(pyr._L.vm_solve(pyr._h, 500.0, 1.0, None, 0, None, 1, prog))
    for el in range(5):
        p = prog[el]
        print(rep, "level", el, "iters", p.iters, "clk dense", p.clk_shader_ticks[0], p.clk_wall_ticks[0], "->", (p.clk_shader_ticks[0] / p.clk_wall_ticks[0] * 100 if p.clk_wall_ticks[0] else 0), "MHz;  pass", p.clk_shader_ticks[1], p.clk_wall_ticks[1], "->", (p.clk_shader_ticks[1] / p.clk_wall_ticks[1] * 100 if p.clk_wall_ticks[1] else 0), "MHz")
PY
cat gpurun_out/sclk_r05b.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "per_pair_constraints or bench" > gpurun_out/pytest_r05b_more.log 2>&1; echo "more rc=$?"; tail -5 gpurun_out/pytest_r05b_more.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05b.json 2> gpurun_out/bench_r05b.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r05b.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05b.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "evals_per_s", "line_searches_per_s", "valu_frac", "sclk_mhz_observed", "ms_converging_steps", "ms_cycling_steps"):
    print(k, d.get(k))
print(d["config"].get("cycling_frame_ids"))
for k in ("scale_reference", "config3_4k", "pipeline_config4_30_frames", "pipeline_config4_8_pairs", "poisson_extend_1080p_ex192", "video_pipeline_5_frames"):
    print(k, json.dumps(d.get(k)))
PY
