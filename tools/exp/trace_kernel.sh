#!/bin/bash
# dev: per-dispatch durations of one kernel (substring) from a rocprofv3 kernel trace: tools/exp/trace_kernel.sh <substr> <script> [args]
k=$1; shift
O=gpurun_out/prof_trace
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 "$@" > $O/run.log 2> $O/run.err; echo "rc=$?"
python3 - $O "$k" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(len(d), "dispatches; us:", " ".join("%.0f" % x for x in d[:80]))
PY
tail -2 $O/run.log
find $O -name "*kernel_trace.csv" -delete
