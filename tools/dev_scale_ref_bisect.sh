#!/bin/bash
# dev: VERDICT r4 item 7 -- bench.py's scale_reference (config[2]'s 60 pairs on one GPU) took 970 ms when it ran AFTER the
# other extras and 838 ms before them.  Which extra does it?  One bench.py process per candidate: config[1] headline
# (1 step), then ONLY that extra, then scale_reference; "none" = no extra in between, "first" = scale_reference before
# all extras (the shipping order), "all" = after all of them.  Prints ms_per_step of scale_reference per case.
mkdir -p gpurun_out
out=gpurun_out/scale_ref_bisect.txt; : > $out
run() { # label, extra args
  local label=$1; shift
  timeout 900 python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /tmp/sr_$label.json 2> /tmp/sr_$label.err
  python - "$label" /tmp/sr_$label.json >> $out <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    sr = d["scale_reference"]
    print("%-12s scale_reference %8.1f ms  (%s)" % (sys.argv[1], sr.get("ms_per_step", -1), sr.get("error", "ok")))
except Exception as e:
    print("%-12s FAILED %s" % (sys.argv[1], e))
PY
  tail -1 $out
}
run first --extras render
run none --scale-ref-last --extras nothing
for x in semantics math batched temporal sync render poisson qpath pipeline8 pipeline30 config3; do
  run $x --scale-ref-last --extras $x
done
run all --scale-ref-last
