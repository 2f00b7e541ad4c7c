#!/bin/bash
# static partition study of config[4] (30 pairs: constrained solve + Poisson + 9 renders per pair): the whole job on this one
# GPU, then every rank's shard of a G-rank job alone (bench.py --config 4 --as-rank k --of G).
# usage: bash tools/prof_config4_partition.sh "<G list, e.g. 2 4>"
O=gpurun_out/config4_partition
mkdir -p $O
timeout 900 python3 bench.py --config 4 --steps 3 --warmup 1 < /dev/null > $O/G1.json 2> $O/G1.err
for G in ${1:-2 4}; do
  for k in $(seq 0 $((G-1))); do
    timeout 900 python3 bench.py --config 4 --as-rank $k --of $G --steps 3 --warmup 1 < /dev/null > $O/G${G}_rank$k.json 2> $O/G${G}_rank$k.err
  done
done
python3 - $O "${1:-2 4}" <<'PY'
import json, sys
d = sys.argv[1]
rd = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
one = rd(d + "/G1.json")
out = {"G1": {"ms_per_job": one["ms_per_step"], "frames_per_s": one["value"], "pipeline": one["pipeline"]}}
for G in [int(x) for x in sys.argv[2].split()]:
    rs = [rd("%s/G%d_rank%d.json" % (d, G, k)) for k in range(G)]
    ms = [r["ms_per_step"] for r in rs]
    out["G%d" % G] = {"pairs_per_rank": [r["as_rank"]["pairs"] for r in rs], "ms_per_rank": ms, "max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 3),
                      "job_ms_static_partition": max(ms), "frames_per_s_projected": round(30 * 9 / (max(ms) * 1e-3), 1),
                      "fraction_of_linear": round(one["ms_per_step"] / max(ms) / G, 3),
                      "solve_s_per_rank": [r["pipeline"]["per_rank"][0]["solve_s_per_step"] for r in rs],
                      "compositor_s_per_rank": [r["pipeline"]["per_rank"][0]["compositor_s_per_step"] for r in rs]}
print(json.dumps(out))
open(d + "/summary.json", "w").write(json.dumps(out, indent=1))
PY
