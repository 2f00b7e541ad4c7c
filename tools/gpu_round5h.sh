#!/bin/bash
# r05: render fast path test, final chaos-floor tables (final key names), final default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "render or bench" > gpurun_out/pytest_r05h.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r05h.log
timeout 1500 python tools/dev_chaos_floor.py --trunc > gpurun_out/chaos_r05h.jsonl 2> gpurun_out/chaos_r05h.err; echo "chaos rc=$?"
timeout 1500 python tools/dev_chaos_floor.py --4k --trunc 0 3 > gpurun_out/chaos_r05h_4k.jsonl 2> gpurun_out/chaos_r05h_4k.err; echo "chaos4k rc=$?"
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05h.json 2> gpurun_out/bench_r05h.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench_r05h.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05h.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "ms_converging_steps", "ms_cycling_steps", "render_frames_per_s", "render_frames_per_s_with_quadratic_path"):
    print(k, d.get(k))
for k in ("scale_reference", "pipeline_config4_30_frames", "poisson_extend_1080p_ex192"):
    print(k, json.dumps(d.get(k))[:900])
PY
