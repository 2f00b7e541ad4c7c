#!/bin/bash
# copy the summaries of a tools/prof_round.sh visit (gpurun_out/prof_<tag>) into profiles/
tag=${1:?tag}; R=${2:-r02}
S=gpurun_out/prof_$tag
cp $S/n1/n1_kernel_stats.csv profiles/${R}_n1_kernel_stats.csv
cp $S/b8/b8_kernel_stats.csv profiles/${R}_batch8_kernel_stats.csv
cp $S/comp/comp_kernel_stats.csv profiles/${R}_compositor_kernel_stats.csv
cp $S/n1_pmc_summary.csv profiles/${R}_n1_pmc_summary.csv
cp $S/b8_pmc_summary.csv profiles/${R}_batch8_pmc_summary.csv
cp $S/n1_traffic.json profiles/traffic_latest.json
cp $S/b8_traffic.json profiles/${R}_batch8_traffic.json
cp $S/n1_bench.json profiles/${R}_n1_bench_under_rocprof.json
cp $S/b8_bench.json profiles/${R}_batch8_bench_under_rocprof.json
cp $S/config3_bench.json profiles/${R}_config3_bench.json
cp $S/comp.log profiles/${R}_compositor.log
ls profiles
