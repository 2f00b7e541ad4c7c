#!/bin/bash
# dev: solver streams per GPU revisited with side-by-side streams (bench.solver_contexts): ms per step of the 8-pair shard and of the
# 60-pair job per --inflight; GPU_MAX_HW_QUEUES from the environment (bench.py's default: 8).  usage: bash tools/exp/inflight.sh "<8-pair list>" "<60-pair list>"
for inf in ${1:-1 2 3 4 8}; do echo -n "8 pairs, inflight $inf: "; timeout 600 python bench.py --config 2 --pairs 60 --as-rank 0 --of 8 --inflight $inf --steps 2 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['solver_streams_rejected_for_sharing_a_hardware_queue'], d['config'].get('pairs_per_launch'))"; done
for inf in ${2:-2 3 4 6}; do echo -n "60 pairs, inflight $inf: "; timeout 600 python bench.py --config 2 --pairs 60 --inflight $inf --steps 1 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['solver_streams_rejected_for_sharing_a_hardware_queue'], d['config'].get('pairs_per_launch'))"; done
