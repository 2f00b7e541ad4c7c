#!/bin/bash
# copy the summaries of a tools/prof_round3.sh visit (gpurun_out/prof_<tag>) into profiles/
tag=${1:?tag}; R=${2:-r03}
S=gpurun_out/prof_$tag
for n in n1 b8 c2; do
  N=$n; [ $n = b8 ] && N=batch8; [ $n = c2 ] && N=config2_60
  cp $S/$n/${n}_kernel_stats.csv profiles/${R}_${N}_kernel_stats.csv
  cp $S/${n}_pmc_summary.csv profiles/${R}_${N}_pmc_summary.csv
  cp $S/${n}_sq_table.csv profiles/${R}_${N}_sq_counters.csv
  tail -1 $S/${n}_out.txt > profiles/${R}_${N}_bench_under_rocprof.json
done
cp $S/comp/comp_kernel_stats.csv profiles/${R}_compositor_kernel_stats.csv
cp $S/comp_out.txt profiles/${R}_compositor.log
cp $S/config3_bench.json profiles/${R}_config3_bench.json
cp $S/config2_1gpu_bench.json profiles/${R}_config2_1gpu_bench.json
python3 tools/pmc_summary.py --merge profiles/traffic_latest.json $S/n1_traffic.json 1 1
python3 tools/pmc_summary.py --merge profiles/traffic_latest.json $S/b8_traffic.json 2 4
python3 tools/pmc_summary.py --merge profiles/traffic_latest.json $S/c2_traffic.json 2 30
python3 tools/pmc_summary.py --merge-sq profiles/traffic_latest.json profiles/${R}_n1_sq_counters.csv 1 1
python3 tools/pmc_summary.py --merge-sq profiles/traffic_latest.json profiles/${R}_batch8_sq_counters.csv 2 4
python3 tools/pmc_summary.py --merge-sq profiles/traffic_latest.json profiles/${R}_config2_60_sq_counters.csv 2 30
ls profiles | grep $R
