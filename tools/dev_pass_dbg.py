"""dev: STEP vs PASS iteration by iteration on one small level: counters and every state array"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 120, 68
i0, i1 = synth.make_pair(w, h)
v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(3).randn(h, w, 2)).astype(np.float32)
pyrs = {}
for name in ("step", "pass"):
    pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), (60, 34)])
    pyr.upload_luma(1, i0, i1); pyr[1].v = v0
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
    pyrs[name] = pyr
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for it in range(0, int(sys.argv[2]) if len(sys.argv) > 2 else 40, chunk):
    res = {}
    for name, sched in (("step", capi.SWEEP_STEP), ("pass", capi.SWEEP_PASS)):
        ctx.set_tuning(sched, 0, 0)
        pr = capi.Progress()
        capi.check(pyrs[name]._L.vm_optimize_level(pyrs[name]._h, 0, float(chunk), None, 1, C.byref(pr)))
        res[name] = (pr.commits, pr.candidates, pr.evaluations, pr.active_tiles)
    diff = [f for f in STATE if not np.array_equal(pyrs["step"][1].field(f).view(np.uint32), pyrs["pass"][1].field(f).view(np.uint32))]
    print(it, res["step"], res["pass"], "state differs in:", diff)
