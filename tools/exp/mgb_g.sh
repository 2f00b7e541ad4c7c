#!/bin/bash
# dev experiment: list entries per workgroup in the Poisson solver's dot-product kernels (MGB_G)
for g in ${GS:-8 16}; do
  echo "== MGB_G=$g"
  bash tools/gpu_variant.sh "-DMGB_G=$g" python tools/dev_poisson_batch.py 2>&1 | grep -v "^variant" | tail -4
done
echo "== default (4)"; python tools/dev_poisson_batch.py | tail -4
timeout 600 python -m pytest tests -m gpu -x -q -k "poisson or pipeline" 2>&1 | tail -2
