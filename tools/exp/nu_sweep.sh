#!/bin/bash
# dev: the cycle's sweeps per level (VM_MGB_NU; tools/exp/mg_prototype.py is the CPU model) -- ms per frame and iterations of
# the 1080p Poisson extension per setting, the quadratic path, and the Poisson parity tests under V(2,2) on every level
# usage (GPU box): bash tools/exp/nu_sweep.sh "1 1,1,2 2" > gpurun_out/nu_sweep.log 2>&1
cd "$(dirname "$0")/../.."
for nu in ${1:-1 1,1,2 2}; do
    for tol in 1e-5 1e-6; do
        echo "== VM_MGB_NU=$nu tol $tol"
        VM_MGB_NU=$nu timeout 300 python tools/dev_poisson_batch.py $tol < /dev/null 2>&1 | tail -4
    done
    echo "== VM_MGB_NU=$nu quadratic path"
    VM_MGB_NU=$nu timeout 300 python tools/dev_qpath.py < /dev/null 2>&1 | grep converged
done
VM_MGB_NU=2 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_fullsize_compositor.py -m gpu -x -q < /dev/null 2>&1 | tail -3
