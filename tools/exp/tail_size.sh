#!/bin/bash
# dev: where the one-workgroup tail of the Poisson cycle should start: VM_MGB_TAIL_X / _B (cells of LDS) -> ms per frame, per-kernel us
for d in "" "-DVM_MGB_TAIL_X=1280 -DVM_MGB_TAIL_B=512" "-DVM_MGB_TAIL_X=384 -DVM_MGB_TAIL_B=128"; do
  echo "== $d"
  for nf in 4 1; do
  if [ -z "$d" ]; then timeout 300 bash tools/prof_any.sh t tools/prof_poisson4.py $nf 1e-5 < /dev/null 2>&1 | grep -E "tail|restrict<false>|prolong<false>|frames per"
  else timeout 600 bash tools/gpu_variant.sh "$d" bash tools/prof_any.sh t tools/prof_poisson4.py $nf 1e-5 < /dev/null 2>&1 | grep -E "tail|restrict<false>|prolong<false>|frames per"; fi
  done
done
