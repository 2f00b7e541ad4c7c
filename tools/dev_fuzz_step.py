"""dev: STEP == SPLIT (FAST, bitwise) on random level sizes, boundary conditions and constraints"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth
import oracle as O
import test_gpu_parity as T

ctx = morph.Context(0, capi.MATH_FAST)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
names = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    w, h = int(rng.randint(10, 420)), int(rng.randint(10, 160))
    bcond = int(rng.randint(0, 3))
    ncons = int(rng.randint(0, 4))
    iters = float(rng.randint(1, 7))
    cons = synth.make_constraints(w, h, ncons) if ncons and min(w, h) > 40 else ()
    kw = dict(w_tps=float(10 ** rng.uniform(-3, 0)), w_ssim=float(10 ** rng.uniform(0, 3)), w_ui=float(10 ** rng.uniform(3, 6)),
              ssim_clamp=float(rng.choice([0.0, 0.0, 0.3])), eps=float(rng.choice([0.01, 0.01, 0.003, 0.03])))
    res = []
    for sched in (capi.SWEEP_SPLIT, capi.SWEEP_STEP):
        P = T._params(O, bcond=bcond, **kw)
        lo, pyr, P = T._make_level(ctx, O, w, h, cons=cons, P=P, seed=trial)
        ctx.set_tuning(sched, 0, 0)
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, 1, C.byref(pr)))
        res.append(([pyr[1].field(n).copy() for n in names], pr.commits))
    diff = [n for n, a, b in zip(names, res[0][0], res[1][0]) if not np.array_equal(a.view(np.uint32), b.view(np.uint32))]
    if diff or res[0][1] != res[1][1]:
        bad += 1
        print("MISMATCH", w, h, bcond, ncons, iters, diff, res[0][1], res[1][1])
print("fuzz done, mismatches:", bad)
