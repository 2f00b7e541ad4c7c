#!/bin/bash
# one GPU-box visit: parity tests, the default bench line, a kernel-trace profile of one solve
# usage (on the box, from the repo root): bash tools/gpu_round.sh <tag> [pytest-args]
tag=${1:-r02}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/pytest_$tag.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_$tag.log
tail -5 gpurun_out/pytest_$tag.log
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/bench_$tag.json
