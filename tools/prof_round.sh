#!/bin/bash
# one GPU-box visit for profiles: kernel-trace stats of (i) the N=1 config[1] solve, (ii) an
# 8-pair 1080p batch (config[2]'s per-GPU unit), (iii) the compositor; PMC passes (FETCH_SIZE,
# WRITE_SIZE, separately, kernel-trace only) of (i) and (ii); the config[3] bench line.
# usage (on the box, repo root): bash tools/prof_round.sh <tag>
tag=${1:-r02}
O=gpurun_out/prof_$tag
mkdir -p $O
cd /tmp 2>/dev/null; cd - >/dev/null
export TMPDIR=/tmp
R=$PWD
P1="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
P2="bench.py --config 2 --pairs 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/n1 -o n1 -- python3 $P1 > $O/n1_bench.json 2> $O/n1.err; echo "n1 rc=$?"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/b8 -o b8 -- python3 $P2 > $O/b8_bench.json 2> $O/b8.err; echo "b8 rc=$?"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/comp -o comp -- python3 tools/prof_compositor.py > $O/comp.log 2> $O/comp.err; echo "comp rc=$?"
Q1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
Q2="bench.py --config 2 --pairs 8 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/n1_fetch -o f -- python3 $Q1 > /dev/null 2> $O/n1_fetch.err; echo "n1 fetch rc=$?"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/n1_write -o w -- python3 $Q1 > /dev/null 2> $O/n1_write.err; echo "n1 write rc=$?"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/b8_fetch -o f -- python3 $Q2 > /dev/null 2> $O/b8_fetch.err; echo "b8 fetch rc=$?"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/b8_write -o w -- python3 $Q2 > /dev/null 2> $O/b8_write.err; echo "b8 write rc=$?"
python3 tools/pmc_summary.py $O/n1_fetch $O/n1_write $O/n1_pmc_summary.csv $O/n1_traffic.json "python3 $Q1" > /dev/null
python3 tools/pmc_summary.py $O/b8_fetch $O/b8_write $O/b8_pmc_summary.csv $O/b8_traffic.json "python3 $Q2" > /dev/null
# keep the summaries, drop the bulky per-dispatch traces
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
timeout 600 python3 bench.py --config 3 --steps 3 --warmup 1 --no-extras > $O/config3_bench.json 2> $O/config3.err; echo "config3 rc=$?"
ls -la $O; du -sh $O
