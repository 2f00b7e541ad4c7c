#!/bin/bash
# r05: the whole GPU suite once more (exit code!), then the sweep kernels' profile visit for this round's records
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_r05g.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_r05g.log
tail -6 gpurun_out/pytest_r05g.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r05g.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke_r05g.log
bash tools/prof_sweeps.sh r05 > gpurun_out/prof_sweeps_r05.log 2>&1; tail -5 gpurun_out/prof_sweeps_r05.log
