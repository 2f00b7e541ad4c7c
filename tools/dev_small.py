"""dev: the 120x68 level alone, TILE vs SPLIT schedule"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = int(sys.argv[2]) if len(sys.argv) > 2 else 120, int(sys.argv[3]) if len(sys.argv) > 3 else 68
mode = {"tile": 1, "split": 2, "auto": 0}[sys.argv[1]]
parts = int(sys.argv[4]) if len(sys.argv) > 4 else 0
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
pyr.upload_luma(1, i0, i1)
pyr[1].v = np.zeros((h, w, 2), np.float32)
capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
ctx.set_tuning(mode, 0, parts)
for n in (10, 50, 50):
    pr = capi.Progress()
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(n), None, 1, C.byref(pr)))
    print(sys.argv[1], "%dx%d" % (w, h), "iters", n, "ms/iter %.3f" % (pr.elapsed_ms / n), "launches", pr.launches, "cand/iter", pr.candidates / n)
