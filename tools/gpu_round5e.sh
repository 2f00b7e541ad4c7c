#!/bin/bash
# r05 GPU visit: the whole GPU suite, the default bench line, the scale_reference bisect
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r05e.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r05e.log
tail -8 gpurun_out/pytest_r05e.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05e.json 2> gpurun_out/bench_r05e.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench_r05e.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05e.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "evals_per_s", "line_searches_per_s", "valu_frac", "sclk_mhz_observed", "ms_converging_steps", "ms_cycling_steps"):
    print(k, d.get(k))
print(d["config"].get("cycling_frame_ids"))
for k in ("scale_reference", "config3_4k", "pipeline_config4_30_frames", "pipeline_config4_8_pairs", "poisson_extend_1080p_ex192", "video_pipeline_5_frames"):
    print(k, json.dumps(d.get(k)))
PY
bash tools/dev_scale_ref_bisect.sh > /dev/null 2>&1; cat gpurun_out/scale_ref_bisect.txt
