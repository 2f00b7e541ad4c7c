#!/bin/bash
# r05 closing records: compositor counters of the final solver, the 20-step bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/prof_pmc.sh r05k_compositor "k_" tools/prof_compositor.py > gpurun_out/prof_r05k.log 2>&1; tail -2 gpurun_out/prof_r05k.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05k.json 2> gpurun_out/bench_r05k.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05k.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "ms_converging_steps", "ms_cycling_steps", "render_frames_per_s", "render_frames_per_s_with_quadratic_path"):
    print(k, d.get(k))
for k in ("pipeline_config4_30_frames", "poisson_extend_1080p_ex192", "video_pipeline_5_frames"):
    print(k, json.dumps(d.get(k))[:1000])
PY
