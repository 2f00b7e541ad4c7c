#!/bin/bash
# r04 first GPU visit: the new/changed tests first, then the chaos-floor table, then the whole suite, then bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "commit_order or pass_timeout or pass_token or pass_schedule_is_bit or bench_json or scale_reference" > gpurun_out/pytest_r04a_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r04a_new.log
tail -15 gpurun_out/pytest_r04a_new.log
timeout 900 python tools/dev_chaos_floor.py > gpurun_out/chaos_r04a.jsonl 2> gpurun_out/chaos_r04a.err; echo "chaos rc=$?"
tail -c 1500 gpurun_out/chaos_r04a.jsonl
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_r04a.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r04a.log
tail -8 gpurun_out/pytest_r04a.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/bench_r04a.json
