#!/bin/bash
# profile visit of the sweep kernels (rounds 3-5): kernel-trace stats + PMC passes (each on its own, kernel-trace only) of
# (i) config[1] (one 1080p pair per step), (ii) an 8-pair batch (config[2]'s per-GPU share at N = 8:
# two streams x 4 pairs per launch), (iii) config[2] on one GPU (60 pairs, 2 x 30 per launch),
# (iv) the compositor; the config[3] bench line.  usage (on the box, repo root): bash tools/prof_sweeps.sh <tag>
tag=${1:-r03}
O=gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
pmc() { # out-name, counters..., then "--" and the command
  local nm=$1; shift; local cs=(); while [ "$1" != "--" ]; do cs+=("$1"); shift; done; shift
  timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc "${cs[@]}" -d $O/$nm -o p -- python3 "$@" > /dev/null 2> $O/$nm.err; echo "pmc $nm rc=$?"
}
kt() { local nm=$1; shift; timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/$nm -o $nm -- python3 "$@" > $O/${nm}_out.txt 2> $O/$nm.err; echo "kt $nm rc=$?"; }
P1="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
Q1="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
P2="bench.py --config 2 --pairs 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
Q2="bench.py --config 2 --pairs 8 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
P3="bench.py --config 2 --pairs 60 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
Q3="bench.py --config 2 --pairs 60 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
if [ -z "$PMC_ONLY" ]; then
kt n1 $P1
kt b8 $P2
kt c2 $P3
kt comp tools/prof_compositor.py
fi
for w in n1:"$Q1" b8:"$Q2" c2:"$Q3"; do
  n=${w%%:*}; q=${w#*:}
  pmc ${n}_fetch FETCH_SIZE -- $q
  pmc ${n}_write WRITE_SIZE -- $q
  pmc pmc_${n}sq1 $SQ1 -- $q
  pmc pmc_${n}sq2 $SQ2 -- $q
  pmc pmc_${n}grbm GRBM_GUI_ACTIVE -- $q
  python3 tools/pmc_summary.py $O/${n}_fetch $O/${n}_write $O/${n}_pmc_summary.csv $O/${n}_traffic.json "python3 $q" > /dev/null
  mkdir -p $O/sq_$n; mv $O/pmc_${n}sq1 $O/sq_$n/pmc_sq1; mv $O/pmc_${n}sq2 $O/sq_$n/pmc_sq2; mv $O/pmc_${n}grbm $O/sq_$n/pmc_grbm; cp -r $O/$n $O/sq_$n/kt; [ -f $O/sq_$n/kt/${n}_kernel_stats.csv ] && cp $O/sq_$n/kt/${n}_kernel_stats.csv $O/sq_$n/kt/kt_kernel_stats.csv
  python3 tools/pmc_table.py $O/sq_$n > $O/${n}_sq_table.csv 2>> $O/$n.err
  rm -rf $O/sq_$n
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
if [ -z "$PMC_ONLY" ]; then
timeout 600 python3 bench.py --config 3 --steps 3 --warmup 1 --no-extras > $O/config3_bench.json 2> $O/config3.err; echo "config3 rc=$?"
timeout 600 python3 bench.py --config 2 --pairs 60 --gpus 1 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/config2_1gpu_bench.json 2> $O/config2.err; echo "config2 rc=$?"
fi
ls $O; du -sh $O
