"""dev: the small levels of a 1080p pyramid under the STEP and PASS schedules (ms per iteration,
per-phase time), from the state the solver really reaches them in (coarse solve + coarser levels)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
mode = capi.MATH_FAST if (len(sys.argv) < 2 or sys.argv[1] == "fast") else capi.MATH_EXACT
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx = morph.Context(0, mode)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h)
res = {}
for name, sched in (("step", capi.SWEEP_STEP), ("pass", capi.SWEEP_PASS), ("auto", capi.SWEEP_AUTO)):
    pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
    L = pyr._L
    nl = pyr.size() - 1
    capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))
    for el in (nl - 1, nl - 2):
        capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
        capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
        lv = pyr[el]
        ctx.set_tuning(sched if (name != "auto") else 0, 0, 0)
        for rep in range(2 if el == nl - 1 else 1):
            if rep:
                capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
                capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
            pr = capi.Progress()
            capi.check(L.vm_optimize_level(pyr._h, el - 1, float(iters), None, 1, C.byref(pr)))
        ctx.set_tuning(0, 0, 0)
        print("%-5s %dx%d: %7.2f ms per %d iterations = %6.3f ms/iter, %6.2f us per phase; launches %d (sched ms %s), cand/iter %.0f commits %d" % (
            name, lv.width, lv.height, pr.elapsed_ms, pr.iters, pr.elapsed_ms / pr.iters, pr.elapsed_ms * 1e3 / pr.iters / 16, pr.launches,
            [round(x, 1) for x in pr.sched_ms], pr.candidates / pr.iters, pr.commits), flush=True)
        res[(name, el)] = pyr[el].v
import hashlib
for k in sorted(res):
    print("sha1", k, hashlib.sha1(np.ascontiguousarray(res[k]).tobytes()).hexdigest()[:16])
for el in (nl - 1, nl - 2):
    print("level", el, "pass == step:", np.array_equal(res[("step", el)].view(np.uint32), res[("pass", el)].view(np.uint32)))
