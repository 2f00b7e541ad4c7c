#!/bin/bash
# r05 closing visit: what the driver will run -- the whole GPU suite, smoke(), the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_r05i.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r05i.log
tail -4 gpurun_out/pytest_r05i.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r05i.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke_r05i.log
( time python bench.py ) > gpurun_out/bench_r05i_default.json 2> gpurun_out/bench_r05i_default.err; echo "bench rc=$?"; tail -4 gpurun_out/bench_r05i_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05i_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "evals_per_s", "sclk_mhz_observed", "render_frames_per_s")})
print(d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_same_sample"]["bit_identical"])
PY
