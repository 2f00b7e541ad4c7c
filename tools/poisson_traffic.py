"""profiles/poisson_traffic_latest.json from the PMC summary of tools/prof_poisson4.py (tools/prof_pmc.sh r06p4 ...): HBM bytes
per system and PCG iteration = sum over the kernels of the iteration (update, dirspmv, the cycle's restrictions / prolongations
of every level, the tail) of launches x bytes per launch / PCG iterations (= launches of k_mgb_dirspmv) / systems per launch.
usage: python tools/poisson_traffic.py gpurun_out/prof_r06p4/r06p4_pmc_summary.csv profiles/r06_poisson4_pmc_summary.csv [systems per launch]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, committed = sys.argv[1], sys.argv[2]
nsys = int(sys.argv[3]) if len(sys.argv) > 3 else 8
rows = {r["kernel"]: r for r in csv.DictReader(open(src))}
it_kernels = [k for k in rows if any(t in k for t in ("k_mgb_update", "k_mgb_dirspmv", "k_mgb_prolong", "k_mgb_restrict", "k_mgb_tail", "k_mgb_dot_rz"))]
# PCG iterations profiled = launches of the kernel that applies the operator (every iteration has exactly one)
iters = sum(int(rows[k]["launches"]) for k in rows if "k_mgb_dirspmv" in k)
total = sum(float(rows[k]["HBM_bytes_per_launch_corrected"]) * int(rows[k]["launches"]) for k in it_kernels)
doc = {
    "bytes_per_system_iteration": total / iters / nsys,
    "systems_per_launch": nsys,
    "pcg_iterations_profiled": iters,
    "source": "%s (tools/prof_poisson4.py: four 1080p frames = eight systems per batch, tol 1e-5; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
              "separate passes): sum over the kernels of the PCG iterations (update, dirspmv, every level's restriction / prolongation, the "
              "tail) of launches x bytes per launch / iterations / 8 systems; FETCH_SIZE doubled per the gfx950 note" % committed,
    "per_kernel_bytes_per_launch": {k: float(rows[k]["HBM_bytes_per_launch_corrected"]) for k in sorted(it_kernels)},
    "per_kernel_launches": {k: int(rows[k]["launches"]) for k in sorted(it_kernels)},
}
json.dump(doc, open(os.path.join(ROOT, "profiles", "poisson_traffic_latest.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
