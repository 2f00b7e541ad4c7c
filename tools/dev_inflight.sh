#!/bin/bash
# dev: config[2] on one GPU -- pairs in lockstep per launch (batch) vs concurrently on several streams (inflight)
for combo in "32 1 32" "32 2 16" "32 4 8" "32 8 4" "8 1 8" "8 2 4" "8 4 2" "8 8 1" "60 1 30" "60 2 30" "60 4 15"; do
  set -- $combo
  python bench.py --config 2 --pairs $1 --inflight $2 --max-batch $3 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pairs $1 inflight $2 batch $3: value %.0f ms/step %.1f' % (d['value'], d['ms_per_step']))"
done
