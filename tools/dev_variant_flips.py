"""dev: how large a perturbation is each arithmetic?  One level of a real 1080p pyramid (the state the solver reaches
it in), 1 / 3 / 10 sweeps from the same start under EXACT (the oracle's arithmetic) and, against it: EXACT with another
commit order, EXACT_FMA, REF_FASTMATH (the reference's own --use_fast_math build restated) and FAST.  Per variant the
fraction of pixels whose v differs from EXACT's by more than 1e-4 / 1e-3 / 1e-2 px, the RMS, and the energy.
usage: tools/dev_variant_flips.py [back: 1 = 120x68, 2 = 240x135, 3 = 480x270]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth  # noqa: E402

back = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = morph.Context(0, capi.MATH_EXACT)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h, frame=3)
VAR = (("exact", capi.MATH_EXACT, 0), ("order1", capi.MATH_EXACT, 1), ("order2", capi.MATH_EXACT, 2), ("fma", capi.MATH_EXACT_FMA, 0),
       ("reffm", capi.MATH_REF_FASTMATH, 0), ("fast", capi.MATH_FAST, 0))
for sweeps in (1, 3, 10, 40):
    res = {}
    for name, mode, order in VAR:
        ctx.set_math_mode(capi.MATH_EXACT)
        ctx.set_commit_order(0)
        p = morph.Pyramid(ctx)
        p.build(i0, i1, 32)
        L = p._L
        nl = p.size() - 1
        el = nl - back
        capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
        for e in range(nl - 1, el, -1):          # coarser levels: always the oracle's arithmetic
            capi.check(L.vm_upsample_v(p._h, e - 1, e))
            capi.check(L.vm_init_level(p._h, e - 1, w, h, None, 0))
            capi.check(L.vm_optimize_level(p._h, e - 1, 500.0, None, 0, None))
        capi.check(L.vm_upsample_v(p._h, el - 1, el))
        ctx.set_math_mode(mode)
        ctx.set_commit_order(order)
        capi.check(L.vm_init_level(p._h, el - 1, w, h, None, 0))
        pr = capi.Progress()
        capi.check(L.vm_optimize_level(p._h, el - 1, float(sweeps), None, 1, C.byref(pr)))
        res[name] = (p[el].v, pr.commits, pr.candidates, pr.evaluations)
        lw, lh = p[el].width, p[el].height
        p.clear()
    ctx.set_commit_order(0)
    ref = res["exact"]
    print("level %dx%d, %d sweep(s): exact commits %d candidates %d evaluations %d" % (lw, lh, sweeps, ref[1], ref[2], ref[3]))
    for name, _, _ in VAR[1:]:
        d = np.sqrt(((res[name][0] - ref[0]) ** 2).sum(-1))
        print("  %-7s > 1e-4: %.4f  > 1e-3: %.4f  > 1e-2: %.4f  > 0.1: %.5f  RMS %.5f  max %.3f   commits %d cand %d evals %d" % (
            name, (d > 1e-4).mean(), (d > 1e-3).mean(), (d > 1e-2).mean(), (d > 0.1).mean(), np.sqrt((d ** 2).mean()), d.max(),
            res[name][1], res[name][2], res[name][3]), flush=True)
