#!/bin/bash
# round-6 profile visit: (i) kernel-trace stats of the default headline workload (config[1]; the sweep kernels are unchanged
# since round 5: their PMC / SQ entries in profiles/traffic_latest.json stay valid, bench.py checks the fingerprint),
# (ii) the compositor of one frame and (iii) the Poisson extension of four frames per batch with PMC + SQ passes.
# usage (on the box, repo root): bash tools/prof_round6.sh     -> gpurun_out/prof_r06*/
export TMPDIR=/tmp
O=gpurun_out/prof_r06
mkdir -p $O
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/n1 -o n1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras < /dev/null > $O/n1_out.txt 2> $O/n1.err; echo "kt n1 rc=$?"
cp $(find $O/n1 -name "*kernel_stats.csv" | head -1) $O/r06_n1_kernel_stats.csv; tail -1 $O/n1_out.txt > $O/r06_n1_bench_under_rocprof.json
find $O -name "*kernel_trace.csv" -delete
timeout 900 bash tools/prof_pmc.sh r06comp k_mgb,k_render,k_fill,k_classify,k_setup,k_paste,k_crop,k_qp tools/prof_compositor.py < /dev/null
timeout 900 bash tools/prof_pmc.sh r06p4 k_mgb tools/prof_poisson4.py 4 1e-5 < /dev/null
