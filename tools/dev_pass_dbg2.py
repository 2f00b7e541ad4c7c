"""dev: line-search / evaluation counters of every schedule against the oracle's (EXACT)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_EXACT)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 120, 68
i0, i1 = synth.make_pair(w, h)
v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(3).randn(h, w, 2)).astype(np.float32)
P = O.default_params()
lo = O.Level(w, h); lo.set_images(i0, i1); lo.field("v")[...] = v0; lo.init(0.0)
st = np.zeros(4)
rows = []
for it in range(150):
    lo.optimize_iter(P, st)
    if it % 50 == 49:
        rows.append(st.copy())
print("oracle cumulative stats at 50/100/150:", rows)
for name, sched in (("tile", capi.SWEEP_TILE), ("split", capi.SWEEP_SPLIT), ("step", capi.SWEEP_STEP), ("pass", capi.SWEEP_PASS)):
    pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), (60, 34)])
    pyr.upload_luma(1, i0, i1); pyr[1].v = v0
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
    ctx.set_tuning(sched, 0, 0)
    tot = np.zeros(3)
    out = []
    for it in range(0, 150, 50):
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 50.0, None, 1, C.byref(pr)))
        tot += (pr.commits, pr.candidates, pr.evaluations)
        out.append(tuple(tot))
    print(name, out, "v == oracle:", np.array_equal(pyr[1].v.view(np.uint32), lo.field("v").view(np.uint32)))
