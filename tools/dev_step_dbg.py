"""dev: where the STEP schedule's state departs from SPLIT's (FAST)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth
import oracle as O
import test_gpu_parity as T

ctx = morph.Context(0, capi.MATH_FAST)
names = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")
for (w, h, bcond, ncons) in ((150, 97, capi.BCOND_NONE, 0), (150, 97, capi.BCOND_BORDER, 0), (150, 97, capi.BCOND_NONE, 4), (150, 97, capi.BCOND_BORDER, 4)):
    cons = synth.make_constraints(w, h, ncons) if ncons else ()
    for iters in (1.0, 2.0):
        res = []
        for sched in (capi.SWEEP_SPLIT, capi.SWEEP_STEP):
            P = T._params(O, bcond=bcond)
            lo, pyr, P = T._make_level(ctx, O, w, h, cons=cons, P=P)
            ctx.set_tuning(sched, 0, 0)
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, 1, C.byref(pr)))
            res.append(([pyr[1].field(n).copy() for n in names], pr.commits, pr.candidates))
        msg = []
        for n, a, b in zip(names, res[0][0], res[1][0]):
            d = a.view(np.uint32) != b.view(np.uint32)
            if d.any():
                idx = np.argwhere(d)[0]
                msg.append("%s:%d first %s" % (n, d.sum(), tuple(idx)))
        print(w, h, "bcond", bcond, "cons", ncons, "iters", iters, "commits", res[0][1], res[1][1], "cand", res[0][2], res[1][2], "|", "; ".join(msg) or "identical")
