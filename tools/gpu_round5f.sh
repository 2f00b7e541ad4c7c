#!/bin/bash
# r05: what in the temporal extra slows a LATER scale_reference down?  the lanes' stream priorities, three ways
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/scale_ref_bisect_priority.txt; : > $out
for m in 2 1 0; do
  VM_LANE_PRIORITY=$m timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --scale-ref-last --extras temporal > /tmp/sr_p$m.json 2> /tmp/sr_p$m.err
  python - $m /tmp/sr_p$m.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("VM_LANE_PRIORITY=%s: temporal video %s ms, video pipeline %s; scale_reference after it %.1f ms" % (
    sys.argv[1], d["temporal_video_5_frames"].get("ms_per_video"), d.get("video_pipeline_5_frames", {}).get("solve_plus_compositor_ms_per_video"), d["scale_reference"]["ms_per_step"]))
PY
  tail -1 $out
done
timeout 900 python -m pytest tests -m gpu -q -x -k "temporal or video or bcast or cpp" > gpurun_out/pytest_r05f.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_r05f.log
