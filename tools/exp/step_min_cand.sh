#!/bin/bash
# dev experiment: AUTO's STEP -> TILE/SPARSE crossover (candidates per iteration and pair), config[1], 20 frames
mkdir -p gpurun_out
for t in ${THRESHOLDS:-200 300 450 600 800 1000}; do
  VM_STEP_MIN_CAND=$t timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-scale-ref > /tmp/smc_$t.json 2>/dev/null
  python - $t /tmp/smc_$t.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
pk = {e["kernel"].split(" ")[0]: (e["launches"], e["avg_us"], e["share_of_sweep_time"]) for e in d["roofline"]["per_kernel"]}
print("VM_STEP_MIN_CAND=%s: ms_per_step %.2f median %.2f conv %s cyc %s cycling %s  %s" % (sys.argv[1], d["ms_per_step"], d["step_ms"]["median"], d.get("ms_converging_steps"), d.get("ms_cycling_steps"), d["config"].get("cycling_frame_ids"), pk))
PY
done
