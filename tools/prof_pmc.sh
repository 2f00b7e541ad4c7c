#!/bin/bash
# rocprofv3 visit of any python workload: kernel-trace stats, then the PMC passes (each on its own, with
# --kernel-trace only, as gpurun requires): FETCH_SIZE, WRITE_SIZE, two SQ groups, GRBM_GUI_ACTIVE.
# Leaves <tag>_kernel_stats.csv, <tag>_pmc_summary.csv (HBM bytes per launch, FETCH_SIZE doubled per the gfx950
# note), <tag>_sq_table.csv under gpurun_out/prof_<tag>/ -- copy what is to be judged into profiles/.
# usage (on the box, repo root): bash tools/prof_pmc.sh <tag> <kernel,substrings,for,the,summary> <script> [args]
tag=$1; kern=$2; shift 2
O=gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 "$@" > $O/run.log 2> $O/run.err; echo "kt rc=$?"
pmc() { local nm=$1; shift; timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc $PMCS -d $O/$nm -o p -- python3 "$@" > /dev/null 2> $O/$nm.err; echo "pmc $nm rc=$?"; }
PMCS="FETCH_SIZE" pmc fetch "$@"
PMCS="WRITE_SIZE" pmc write "$@"
PMCS="$SQ1" pmc pmc_sq1 "$@"
PMCS="$SQ2" pmc pmc_sq2 "$@"
PMCS="GRBM_GUI_ACTIVE" pmc pmc_grbm "$@"
python3 tools/pmc_summary.py $O/fetch $O/write $O/${tag}_pmc_summary.csv $O/${tag}_traffic.json "python3 $*" "$kern" > /dev/null
python3 tools/pmc_table.py $O > $O/${tag}_sq_table.csv 2>> $O/run.err
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
rm -rf $O/fetch $O/write $O/pmc_sq1 $O/pmc_sq2 $O/pmc_grbm $O/kt
head -25 $O/${tag}_pmc_summary.csv; tail -3 $O/run.log
