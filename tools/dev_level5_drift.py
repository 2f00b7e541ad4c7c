"""dev: the 120x68 level of config[1] never converges within its 500 iterations -- the final field of a solve is wherever
that slow descent stands after 500 sweeps.  From the SAME start (coarse solve + upsample + init in EXACT), the level under
EXACT, EXACT order 2, EXACT_FMA, REF_FASTMATH and FAST: at checkpoints the level's energy, the RMS distance of each
variant's field from EXACT's (level pixels), and how far "ahead" of EXACT it is along EXACT's own direction of travel
(projection of v_variant - v_exact onto v_exact - v_start, in units of that travel).
usage: tools/dev_level5_drift.py [frame ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth  # noqa: E402
import oracle as O  # noqa: E402

frames = [int(a) for a in sys.argv[1:]] or [3, 9, 15]
ctx = morph.Context(0, capi.MATH_EXACT)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
VAR = (("exact", capi.MATH_EXACT, 0), ("order2", capi.MATH_EXACT, 2), ("fma", capi.MATH_EXACT_FMA, 0), ("reffm", capi.MATH_REF_FASTMATH, 0),
       ("fast", capi.MATH_FAST, 0))
CHECK = (10, 30, 100, 250, 500)
P = O.default_params()


def energy(i0, i1, v):
    lv = O.Level(v.shape[1], v.shape[0])
    lv.set_images(i0, i1)
    lv.field("v")[...] = v
    lv.init(0.0)
    e = lv.energy(P)
    return float(P.w_ssim * e[0] / (v.shape[0] * v.shape[1]) + P.w_tps * e[1]), float(e[0]), float(e[1])


for f in frames:
    i0, i1 = synth.make_pair(w, h, frame=f)
    out = {}
    for name, mode, order in VAR:
        ctx.set_math_mode(capi.MATH_EXACT)
        ctx.set_commit_order(0)
        p = morph.Pyramid(ctx)
        p.build(i0, i1, 32)
        L = p._L
        nl = p.size() - 1
        el = nl - 1
        capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
        capi.check(L.vm_upsample_v(p._h, el - 1, el))
        capi.check(L.vm_init_level(p._h, el - 1, w, h, None, 0))
        v_start = p[el].v
        l0, l1 = p[el].field("img0"), p[el].field("img1")
        ctx.set_math_mode(mode)
        ctx.set_commit_order(order)
        if name == "fast" and os.environ.get("VM_DRIFT_SCHED"):
            ctx.set_tuning(int(os.environ["VM_DRIFT_SCHED"]), 0, int(os.environ.get("VM_DRIFT_PARTS", "0")))
        if mode != capi.MATH_EXACT:
            capi.check(L.vm_init_level(p._h, el - 1, w, h, None, 0))      # the variant's own SSIM values
        done, snaps = 0, {}
        for k in CHECK:
            pr = capi.Progress()
            capi.check(L.vm_optimize_level(p._h, el - 1, float(k - done), None, 1, C.byref(pr)))
            done = k
            snaps[k] = p[el].v
        out[name] = snaps
        ctx.set_tuning(0, 0, 0)
        p.clear()
    ctx.set_math_mode(capi.MATH_EXACT)
    ctx.set_commit_order(0)
    print("frame %d, level %dx%d" % (f, v_start.shape[1], v_start.shape[0]))
    for k in CHECK:
        ve = out["exact"][k]
        travel = ve - v_start
        t2 = float((travel ** 2).sum())
        e_ex = energy(l0, l1, ve)
        line = "  k=%3d  travel RMS %.4f  E_exact %.6f (ssim %.2f tps %.4f) |" % (k, np.sqrt((travel ** 2).sum(-1).mean()), e_ex[0], e_ex[1], e_ex[2])
        for name, _, _ in VAR[1:]:
            v = out[name][k]
            d = v - ve
            e = energy(l0, l1, v)
            line += "  %s: rms %.5f ahead %+.4f dE %+.5f%%" % (name, np.sqrt((d ** 2).sum(-1).mean()), float((d * travel).sum()) / t2, 100 * (e[0] - e_ex[0]) / e_ex[0])
        print(line, flush=True)
        if os.environ.get("VM_DRIFT_WHERE"):
            for name, _, _ in VAR[1:]:
                dd = np.sqrt(((out[name][k] - ve) ** 2).sum(-1))
                hh, ww = dd.shape
                yy, xx = np.mgrid[0:hh, 0:ww]
                edge = np.minimum(np.minimum(xx, ww - 1 - xx), np.minimum(yy, hh - 1 - yy))
                big = dd > 0.03
                top = np.argsort(dd.ravel())[-5:][::-1]
                print("      %-6s max %.3f  >0.03: %3d px (of them within 4 of the border: %3d)  share of sum sq from >0.03: %.2f  rms of the rest %.5f  top: %s" % (
                    name, dd.max(), int(big.sum()), int((big & (edge < 4)).sum()), float((dd[big] ** 2).sum() / max((dd ** 2).sum(), 1e-30)),
                    float(np.sqrt((dd[~big] ** 2).mean())), [(int(t % ww), int(t // ww), round(float(dd.ravel()[t]), 3)) for t in top]), flush=True)
