#!/bin/bash
# r05 third GPU visit: Poisson solver v2 (1-byte level-0 operator, in-wave restriction), C++ shard driver, bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "poisson or pipeline or cpp or per_pair_constraints or bcast or rccl or quadratic" > gpurun_out/pytest_r05c_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r05c_new.log
tail -15 gpurun_out/pytest_r05c_new.log
timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05c_mgb.txt 2>&1; cat gpurun_out/poisson_r05c_mgb.txt
bash tools/prof_pmc.sh r05c_compositor "k_" tools/prof_compositor.py
timeout 900 python -m pytest tests -m gpu -x -q -k "bench" > gpurun_out/pytest_r05c_bench.log 2>&1; echo "benchtests rc=$?"; tail -8 gpurun_out/pytest_r05c_bench.log
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05c.json 2> gpurun_out/bench_r05c.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_r05c.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05c.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "evals_per_s", "line_searches_per_s", "valu_frac", "sclk_mhz_observed", "ms_converging_steps", "ms_cycling_steps"):
    print(k, d.get(k))
print(d["config"].get("cycling_frame_ids"))
for k in ("scale_reference", "config3_4k", "pipeline_config4_30_frames", "pipeline_config4_8_pairs", "poisson_extend_1080p_ex192", "video_pipeline_5_frames"):
    print(k, json.dumps(d.get(k)))
PY
