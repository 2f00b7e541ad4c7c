"""dev: a longer fuzz campaign for the SPARSE kernel against TILE (the round-4 changes: word list in LDS incl. its
overflow into memory, tile selection by reach, compact tile list, dirty-cell write-back): random level sizes,
parameters, constraints, iteration counts, stopping rules, LDS capacities; both arithmetic modes; bit-identical state,
iteration counts and activity counters (incl. the tile-visit counter, deterministic since the reach test).
usage: tools/dev_fuzz_sparse.py [seed] [trials]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph  # noqa: E402
import oracle as O  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 100
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = morph.Context(0, capi.MATH_FAST)
rng = np.random.RandomState(seed)
bad = used = 0
for trial in range(trials):
    mode = capi.MATH_EXACT if trial % 4 == 3 else capi.MATH_FAST
    ctx.set_math_mode(mode)
    w, h, kw, cons = F._draw(rng, 700, 300)
    w, h = max(w, 40), max(h, 40)
    if trial % 3:
        kw = dict(bcond=kw["bcond"], eps=kw["eps"], ssim_clamp=kw["ssim_clamp"])
    iters = float(rng.randint(10, 90))
    fixed = int(rng.randint(0, 2))
    cap = int(rng.choice([0, 0, 1, 5, 17, 64]))
    resm = int(rng.choice([0, 0, 2, 3]))      # resident visits: automatic, re-centred at every commit, given up at the first
    extra = float(rng.choice([0, 150, 400]))  # more TILE sweeps before the comparison: down to a few clusters (what goes resident)
    words = F._plant_clusters(rng, w, h) if trial % 2 else None   # ... or a planted mask of a few clusters,
    sweeps = float(rng.choice([6, 12, 25, 80])) if trial % 2 else 80.0   # on a level that is more or less settled
    st = rng.get_state()
    res = []
    for sched in (capi.SWEEP_TILE, capi.SWEEP_SPARSE):
        rng.set_state(st)
        pyr = F._pruned_level(ctx, O, rng, w, h, kw, cons, trial, sweeps)
        if extra:
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, extra, None, 1, None))     # (the TILE schedule is still set)
        if words is not None:
            pyr[1].set_impmask(words)
        ctx.set_tuning(sched, 0, cap if sched == capi.SWEEP_SPARSE else 0)
        ctx.set_sparse_resident(resm)
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, fixed, C.byref(pr)))
        res.append(([pyr[1].field(n).copy() for n in F.STATE], (pr.iters, pr.improving, pr.commits, pr.candidates, pr.evaluations, pr.active_tiles),
                    pr.sched_launches[3]))
    ok = res[0][1] == res[1][1] and all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(res[0][0], res[1][0]))
    used += res[1][2] > 0
    if not ok:
        bad += 1
        print("MISMATCH trial %d mode %d %dx%d iters %g fixed %d cap %d resident %d: counters %s vs %s" % (trial, mode, w, h, iters, fixed, cap, resm, res[0][1], res[1][1]), flush=True)
ctx.set_tuning(0, 0, 0)
ctx.set_sparse_resident(0)
print("fuzz seed %d: %d trials, sparse kernel used in %d, mismatches %d; tile visits served from the resident LDS copy: %d" % (seed, trials, used, bad, ctx.sparse_resident_visits()))
