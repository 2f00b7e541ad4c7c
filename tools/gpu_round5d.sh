#!/bin/bash
# r05 GPU visit: Poisson solver v3 (12-byte vectors, unconditional loads) -- tests, timing, counters
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "poisson or pipeline" > gpurun_out/pytest_r05d_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r05d_new.log
tail -15 gpurun_out/pytest_r05d_new.log
timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05d_mgb.txt 2>&1; cat gpurun_out/poisson_r05d_mgb.txt
bash tools/prof_pmc.sh r05d_compositor "k_" tools/prof_compositor.py
