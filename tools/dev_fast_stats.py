"""dev: how far does FAST arithmetic drift from the oracle; timing of both modes at 1080p"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from videomorphing_amd import capi, morph, synth

ctx = morph.Context(0)
print(ctx.device_info())
P = O.default_params()
kp = capi.KernParams()
for f, _ in capi.KernParams._fields_: setattr(kp, f, getattr(P, f))
ctx.set_params(kp)

def mk(w, h, mode):
    ctx.set_math_mode(mode)
    i0, i1 = synth.make_pair(w, h)
    rng = np.random.RandomState(0)
    v0 = (0.8 * synth.displacement(w, h) + 0.05 * rng.randn(h, w, 2)).astype(np.float32)
    pyr = morph.Pyramid(ctx)
    pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
    pyr.upload_luma(1, i0, i1); pyr[1].v = v0
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
    return pyr, i0, i1, v0

w, h = 160, 120
pe, i0, i1, v0 = mk(w, h, capi.MATH_EXACT)
pf, _, _, _ = mk(w, h, capi.MATH_FAST)
for n in (1, 5, 20, 60):
    ctx.set_math_mode(capi.MATH_EXACT); capi.check(pe._L.vm_optimize_level(pe._h, 0, float(n), None, 1, None))
    ctx.set_math_mode(capi.MATH_FAST); capi.check(pf._L.vm_optimize_level(pf._h, 0, float(n), None, 1, None))
    a, b = pe[1].v, pf[1].v
    dv = np.sqrt(((a - b) ** 2).sum(-1))
    ea, eb = (1 - pe[1].field("value")).sum(), (1 - pf[1].field("value")).sum()
    print("after +%d iters: dv quantiles 50/90/99/max = %.2e %.2e %.2e %.2e  rms %.3e  frac>1e-3 %.3f >0.05 %.4f  E_ssim %.4f vs %.4f" % (
        n, *np.quantile(dv, [0.5, 0.9, 0.99, 1.0]), np.sqrt((dv**2).mean()), (dv > 1e-3).mean(), (dv > 0.05).mean(), ea, eb))

# timing at 1080p, both modes, 20 fixed-work iterations from a near-solution start
for mode, name in ((capi.MATH_EXACT, "exact"), (capi.MATH_FAST, "fast")):
    p, *_ = mk(1920, 1080, mode)
    pr = capi.Progress()
    capi.check(p._L.vm_optimize_level(p._h, 0, 4.0, None, 1, C.byref(pr)))
    t = []
    for rep in range(3):
        capi.check(p._L.vm_optimize_level(p._h, 0, 10.0, None, 1, C.byref(pr)))
        t.append(pr.elapsed_ms / 10)
    print(name, "1080p ms/iter:", t, "Mpix-iter/s:", 1920 * 1080 / (min(t) * 1e-3) / 1e6, "tiles/cand/commit", pr.active_tiles, pr.candidates, pr.commits)
    mask = p[1].field("impmask")
    print("  active mask bits fraction:", np.unpackbits(mask.view(np.uint8)).sum() / (1920 * 1080.0))
