#!/bin/bash
# r05 second GPU visit: the batched ring-only Poisson solver -- tests, A/B timing against round 4's solver, counters
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "poisson or pipeline or chaos_floor_of_config1 or inside_the_family" > gpurun_out/pytest_r05b_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r05b_new.log
tail -15 gpurun_out/pytest_r05b_new.log
VM_POISSON_SOLVER=mg1 timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05b_mg1.txt 2>&1; cat gpurun_out/poisson_r05b_mg1.txt
timeout 300 python tools/dev_poisson_batch.py > gpurun_out/poisson_r05b_mgb.txt 2>&1; cat gpurun_out/poisson_r05b_mgb.txt
bash tools/prof_pmc.sh r05b_compositor "k_" tools/prof_compositor.py
