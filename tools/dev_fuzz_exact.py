"""dev: EXACT sweeps (TILE, SPLIT and STEP schedules) == oracle, bitwise, on random sizes / conditions

usage: dev_fuzz_exact.py [seed] [trials] [tex8]   (tex8: the REF_TEX8 build against the oracle with CUDA's filter weights)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth
import oracle as O
import test_gpu_parity as T

TEX8 = len(sys.argv) > 3 and sys.argv[3] == "tex8"
O.lib().vmo_set_tex_filter(1 if TEX8 else 0)
ctx = morph.Context(0, capi.MATH_REF_TEX8 if TEX8 else capi.MATH_EXACT)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    w, h = int(rng.randint(10, 300)), int(rng.randint(10, 120))
    bcond = int(rng.randint(0, 3))
    ncons = int(rng.randint(0, 4))
    iters = int(rng.randint(1, 4))
    cons = synth.make_constraints(w, h, ncons) if ncons and min(w, h) > 40 else ()
    kw = dict(w_tps=float(10 ** rng.uniform(-3, 0)), w_ssim=float(10 ** rng.uniform(0, 3)), w_ui=float(10 ** rng.uniform(3, 6)),
              ssim_clamp=float(rng.choice([0.0, 0.0, 0.3])), eps=float(rng.choice([0.01, 0.01, 0.003, 0.03])))
    for sched in (capi.SWEEP_TILE, capi.SWEEP_SPLIT, capi.SWEEP_STEP):
        P = T._params(O, bcond=bcond, **kw)
        lo, pyr, P = T._make_level(ctx, O, w, h, cons=cons, P=P, seed=trial)
        for _ in range(iters):
            lo.optimize_iter(P)
        ctx.set_tuning(sched, 0, 0)
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(iters), None, 1, None))
        try:
            T._assert_state_equal(lo, pyr[1])
        except AssertionError as e:
            bad += 1
            print("MISMATCH", w, h, bcond, ncons, iters, sched, str(e)[:120])
print("fuzz done, mismatches:", bad)
