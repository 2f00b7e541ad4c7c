#!/bin/bash
# static partition study: every rank's shard of config[2] (60 pairs over G ranks), each run
# alone on this one GPU; per-rank ms, max / mean.  usage: bash tools/prof_partition.sh <G> [inflight]
G=${1:-8}; INF=${2:-0}   # 0 = bench.default_streams()
O=gpurun_out/partition_G$G
mkdir -p $O
for k in $(seq 0 $((G-1))); do
  timeout 600 python3 bench.py --config 2 --pairs 60 --as-rank $k --of $G --inflight $INF --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/rank$k.json 2> $O/rank$k.err
done
python3 - $O $G <<'PY'
import json, sys, glob
d, G = sys.argv[1], int(sys.argv[2])
ms, pairs, ex = [], [], []
for k in range(G):
    j = json.loads(open("%s/rank%d.json" % (d, k)).read().strip().splitlines()[-1])
    ms.append(j["ms_per_step"]); pairs.append(j["as_rank"]["pairs"]); ex.append(j["executed_pixel_iters"] / j["steps"])
out = {"G": G, "pairs_per_rank": pairs, "ms_per_rank": ms, "max_over_mean": max(ms) / (sum(ms) / len(ms)),
       "job_ms_static_partition": max(ms),
       "job_executed_G_pixel_iters_per_s": sum(ex) / (max(ms) * 1e-3) / 1e9,
       "job_nominal_G_pixel_iters_per_s": 60 * 500 * (1920*1080 + 960*540 + 480*270 + 240*135 + 120*68) / max(ms) / 1e6}
print(json.dumps(out))
open(d + "/summary.json", "w").write(json.dumps(out, indent=1))
PY
