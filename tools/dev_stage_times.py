"""dev: wall time of every stage of one fixed-work 1080p solve next to the sweep kernels' event time"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
L = pyr._L
nl = pyr.size() - 1
fixed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
if len(sys.argv) > 5:
    ctx.set_tuning(int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
only = int(sys.argv[6]) if len(sys.argv) > 6 else 0     # stop after this many levels


def timed(f):
    ctx.sync(); t = time.perf_counter(); f(); ctx.sync(); return (time.perf_counter() - t) * 1e3


for rep in range(3):
    rows = []
    t_all = time.perf_counter()
    rows.append(("coarse_solve", timed(lambda: capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))), 0, 0))
    for el in range(nl - 1, 0, -1):
        rows.append(("upsample %d" % el, timed(lambda: capi.check(L.vm_upsample_v(pyr._h, el - 1, el))), 0, 0))
        rows.append(("init %d" % el, timed(lambda: capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))), 0, 0))
        pr = capi.Progress()
        ms = timed(lambda: capi.check(L.vm_optimize_level(pyr._h, el - 1, iters, None, fixed, C.byref(pr))))
        rows.append(("optimize %dx%d" % (pyr[el].width, pyr[el].height), ms, pr.elapsed_ms, pr.launches))
        if only and nl - el >= only:
            break
    tot = (time.perf_counter() - t_all) * 1e3
    if rep == 2:
        for name, ms, ev, ln in rows:
            print("%-24s wall %9.3f ms   events %9.3f ms   launches %6d   wall-ev %8.3f" % (name, ms, ev, ln, ms - ev if ev else 0))
        print("total wall %.2f ms, sum events %.2f" % (tot, sum(r[2] for r in rows)))
