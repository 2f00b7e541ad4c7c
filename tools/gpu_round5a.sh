#!/bin/bash
# r05 first GPU visit: host input fingerprints, the new tests (TEX8 parity, hash fixtures), the chaos-floor table
# with the TEX8 members (config[1] x 6 frames, config[3] x 2), the compositor's counters
mkdir -p gpurun_out
export TMPDIR=/tmp
lscpu | grep "Model name" > gpurun_out/r05a_host.txt; nproc >> gpurun_out/r05a_host.txt
python - > gpurun_out/r05a_inputs.txt 2>&1 <<'PY'
import json, sys
sys.path.insert(0, "tests")
import fullsize_hash as FH
from videomorphing_amd import synth
doc = json.load(open("tests/golden/full_solve_hashes.json"))
for k, f in sorted(doc["solves"].items()):
    i0, i1 = synth.make_pair(f["size"][0], f["size"][1], frame=f["frame"])
    print(k, "inputs match" if FH.input_hash(i0, i1) == f["inputs"] else "INPUTS DIFFER")
PY
cat gpurun_out/r05a_host.txt gpurun_out/r05a_inputs.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "tex8 or oracle_hashes" > gpurun_out/pytest_r05a_new.log 2>&1; echo "new rc=$?" | tee -a gpurun_out/pytest_r05a_new.log
tail -15 gpurun_out/pytest_r05a_new.log
timeout 1500 python tools/dev_chaos_floor.py --trunc > gpurun_out/chaos_r05a.jsonl 2> gpurun_out/chaos_r05a.err; echo "chaos rc=$?"
tail -c 600 gpurun_out/chaos_r05a.jsonl
timeout 1500 python tools/dev_chaos_floor.py --4k --trunc 0 3 > gpurun_out/chaos_r05a_4k.jsonl 2> gpurun_out/chaos_r05a_4k.err; echo "chaos4k rc=$?"
bash tools/prof_pmc.sh r05_compositor "k_" tools/prof_compositor.py
