#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "pipeline or another_context or extras_of_round" > gpurun_out/pytest_r05l.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r05l.log
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-ref --extras render,pipeline30 2>gpurun_out/bench_r05l.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['pipeline_config4_30_frames'])[:1300])"; tail -3 gpurun_out/bench_r05l.err
