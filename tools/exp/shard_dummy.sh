#!/bin/bash
# dev: do the solver streams of a multi-stream job depend on which hardware queues the runtime deals them?  k idle contexts are
# created first (VM_DEV_DUMMY_STREAMS); with VM_NO_STREAM_PROBE=1 the job takes the first streams it gets (rounds 1-5).
for k in ${1:-0 5 6}; do echo -n "8 pairs, 2 streams, dummy $k: "; VM_DEV_DUMMY_STREAMS=$k timeout 600 python bench.py --config 2 --pairs 60 --as-rank 0 --of 8 --steps 1 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2> /tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['solver_streams_rejected_for_sharing_a_hardware_queue'])"; grep "solver streams" /tmp/e.txt; done
for k in ${2:-0 5}; do echo -n "60 pairs, 3 streams, dummy $k: "; VM_DEV_DUMMY_STREAMS=$k timeout 600 python bench.py --config 2 --pairs 60 --steps 1 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2> /tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['solver_streams_rejected_for_sharing_a_hardware_queue'])"; grep "solver streams" /tmp/e.txt; done
