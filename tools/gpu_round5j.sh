#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "poisson or pipeline" > gpurun_out/pytest_r05j.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r05j.log
timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05j.txt 2>&1; cat gpurun_out/poisson_r05j.txt
timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-ref --extras render,poisson,pipeline30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['pipeline_config4_30_frames'])[:1200]); print(json.dumps(d['poisson_extend_1080p_ex192'])[:500])"
