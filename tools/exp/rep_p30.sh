#!/bin/bash
# dev: are the config[4] pipeline's compositor figures reproducible across processes and beside the other extras?
# usage: bash tools/exp/rep_p30.sh "<bench args>" [repeats]
for i in $(seq 1 ${2:-3}); do
timeout 600 python bench.py $1 < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline_config4_30_frames']
print(p['ms_per_pair_each_run'], p['compositor_ms_per_frame'], p['compositor_split_ms_per_frame'])"
done
