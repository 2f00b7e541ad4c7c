for of in 8 4 2; do for inf in 2 3 4; do echo -n "rank 0 of $of, inflight $inf: "; timeout 600 python bench.py --config 2 --pairs 60 --as-rank 0 --of $of --inflight $inf --steps 2 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['as_rank']['pairs'], d['ms_per_step'], d['config']['solver_streams_rejected_for_sharing_a_hardware_queue'], d['config'].get('pairs_per_launch'))"; done; done
echo -n "rank 1 of 8 (7 pairs), inflight 3: "; timeout 600 python bench.py --config 2 --pairs 60 --as-rank 1 --of 8 --inflight 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['as_rank']['pairs'], d['ms_per_step'])"
