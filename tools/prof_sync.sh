#!/bin/bash
# dev: kernel trace of the sync stage (tools/dev_sync.py), durations grouped by kernel and grid
tag=${1:-sync}
O=gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 tools/dev_sync.py ${2:-40} ${3:-60} 0 > $O/dev.log 2> $O/dev.err; echo "rc=$?"
python3 - $O <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    import re
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    agg[(name, r["Grid_Size_X"], r["Grid_Size_Y"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg, key=lambda k: -sum(agg[k])):
    v = agg[k]
    print("%-42s grid %8s x %s  n %6d  avg %9.2f us  total %9.2f ms" % (k[0], k[1], k[2], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
PY
tail -8 $O/dev.log
find $O -name "*kernel_trace.csv" -delete
