#!/bin/bash
# dev: A/B of the cycle's sweeps per level (VM_MGB_NU) on ONE box: the bench line's Poisson / quadratic-path extras and the config[4]
# pipeline's compositor, alternating settings
# usage (GPU box): bash tools/exp/nu_ab.sh "1 1,1,2" [rounds]
cd "$(dirname "$0")/../.."
for r in $(seq 1 ${2:-2}); do
for nu in ${1:-1 1,1,2}; do
VM_MGB_NU=$nu timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-scale-ref --extras poisson,qpath,pipeline30 < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline_config4_30_frames']; q=d['poisson_extend_1080p_ex192']
print('nu $nu:', 'poisson', {k[4:]: (v['ms_per_frame'], v['cg_iterations'][0]) for k, v in q.items() if k.startswith('tol_')}, 'qpath', d['quadratic_path_1080p']['ms_per_frame'], d['quadratic_path_1080p']['pcg_iterations'])
print('      pipeline30', p['ms_per_pair_each_run'], p['compositor_ms_per_frame'], p['compositor_split_ms_per_frame'], p['pcg_iterations_min_max'])"
done
done
