#!/bin/bash
# dev: rocprofv3 kernel trace of any script, durations grouped by kernel:  tools/prof_any.sh <tag> <script> [args]
tag=$1; shift
O=gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 "$@" > $O/run.log 2> $O/run.err; echo "rc=$?"
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s n %6s avg %9.2f us total %8.2f ms %5s%%" % (r["Name"].split("(anonymous namespace)::")[-1][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -4 $O/run.log
find $O -name "*kernel_trace.csv" -delete
