#!/bin/bash
# dev: A/B on one box of the PCG update riding in the level-0 restriction against the separate k_mgb_update, for every batch size
# (VM_MGB_FUSE_MIN_SYS: the smallest batch that fuses; 0 = never, 1 = always)
# usage (GPU box): bash tools/exp/fuse_ab.sh [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${1:-2}); do
  for tol in 1e-5 1e-6; do
    echo "== always fused, tol $tol";  VM_MGB_FUSE_MIN_SYS=1 timeout 300 python tools/dev_poisson_batch.py $tol < /dev/null 2>&1 | tail -4 | cut -c1-150
    echo "== never fused, tol $tol"; VM_MGB_FUSE_MIN_SYS=0 timeout 300 python tools/dev_poisson_batch.py $tol < /dev/null 2>&1 | tail -4 | cut -c1-150
  done
done
