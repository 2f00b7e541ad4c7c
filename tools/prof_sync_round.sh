#!/bin/bash
# one GPU-box visit for the sync stage's profiles: (i) rocprofv3 kernel-trace stats of a whole
# 1080p x 60 sync solve (500 -> the reference's schedule), (ii) the finest level alone (300
# iterations): kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only) -> HBM bytes
# per launch of k_sync_A / k_sync_B.   usage (on the box, repo root): bash tools/prof_sync_round.sh <tag>
tag=${1:-r02c}
O=gpurun_out/prof_$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/sync_all -o s -- python3 tools/dev_sync.py 500 60 0 > $O/sync_all.log 2> $O/sync_all.err; echo "all rc=$?"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/sync_l1 -o s -- python3 tools/dev_sync.py 299 60 0 1 > $O/sync_l1.log 2> $O/sync_l1.err; echo "l1 rc=$?"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/sync_l1_fetch -o f -- python3 tools/dev_sync.py 49 60 0 1 > /dev/null 2> $O/sync_l1_fetch.err; echo "fetch rc=$?"
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/sync_l1_write -o w -- python3 tools/dev_sync.py 49 60 0 1 > /dev/null 2> $O/sync_l1_write.err; echo "write rc=$?"
python3 tools/pmc_summary.py $O/sync_l1_fetch $O/sync_l1_write $O/sync_l1_pmc_summary.csv $O/sync_l1_traffic.json "python3 tools/dev_sync.py 49 60 0 1" "k_sync_A<false,k_sync_B<false" > /dev/null; echo "pmc rc=$?"
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -40; tail -9 $O/sync_all.log; tail -4 $O/sync_l1.log; cat $O/sync_l1_pmc_summary.csv | head -12
