#!/bin/bash
# copy the summaries of a tools/prof_sweeps.sh visit (gpurun_out/prof_<tag>) into profiles/ as <R>_* and merge the PMC
# bytes / SQ ratios into profiles/traffic_latest.json (one entry per workload shape = config, pairs per launch).
# usage (build container, repo root): bash tools/collect_profiles.sh <tag> <R>
tag=${1:?tag}; R=${2:?round prefix, e.g. r05}
S=gpurun_out/prof_$tag
for n in n1 b8 c2; do
  N=$n; [ $n = b8 ] && N=batch8; [ $n = c2 ] && N=config2_60
  cp $S/$n/${n}_kernel_stats.csv profiles/${R}_${N}_kernel_stats.csv
  cp $S/${n}_pmc_summary.csv profiles/${R}_${N}_pmc_summary.csv
  cp $S/${n}_sq_table.csv profiles/${R}_${N}_sq_counters.csv
  tail -1 $S/${n}_out.txt > profiles/${R}_${N}_bench_under_rocprof.json
  cfg=$(python3 -c "import json;d=json.loads(open('profiles/${R}_${N}_bench_under_rocprof.json').read());print(1 if 'config[1]' in d['config']['workload'] else 2, d['config']['pairs_per_launch'])")
  python3 tools/pmc_summary.py --merge profiles/traffic_latest.json $S/${n}_traffic.json $cfg
  python3 tools/pmc_summary.py --merge-sq profiles/traffic_latest.json profiles/${R}_${N}_sq_counters.csv $cfg
done
cp $S/config3_bench.json profiles/${R}_config3_bench.json
cp $S/config2_1gpu_bench.json profiles/${R}_config2_1gpu_bench.json
ls profiles | grep "^$R"
