#!/bin/bash
# dev: spread of one rank's 8-pair shard of config[2] (two solver streams) across fresh processes
# usage: bash tools/exp/rep_shard.sh [repeats] [extra bench args]
for i in $(seq 1 ${1:-6}); do
timeout 600 python bench.py --config 2 --pairs 60 --as-rank 0 --of 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras $2 < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['each'] if 'step_ms' in d else '')"
done
