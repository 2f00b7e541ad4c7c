#!/bin/bash
# config[2] on one GPU (60 pairs, two streams x batches of 30): kernel-trace stats + SQ / TCC counter
# passes (each PMC pass on its own, kernel-trace only).  usage (on the box, repo root): bash tools/prof_config2.sh <tag> [pairs]
tag=${1:-r03}; pairs=${2:-60}
O=gpurun_out/prof_c2_$tag
mkdir -p $O
export TMPDIR=/tmp
P="bench.py --config 2 --pairs $pairs --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
Q="bench.py --config 2 --pairs $pairs --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|GRBM_[A-Z_0-9]+" | sort -u > $O/counters_available.txt
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 $P > $O/bench.json 2> $O/kt.err; echo "kt rc=$?"
run_pmc() { # name, counters...
  n=$1; shift
  timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $O/pmc_$n -o p -- python3 $Q > /dev/null 2> $O/pmc_$n.err; echo "pmc $n rc=$?"
}
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
python3 tools/pmc_table.py $O > $O/pmc_table.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
cat $O/pmc_table.txt | head -60
