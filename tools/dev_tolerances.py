"""dev: the two tolerances VERDICT r2 asked to tighten or justify -- measured.
(a) Poisson extension, device multigrid-PCG at several tolerances vs the oracle's CG at 1e-9 (config[4] test case);
(b) FAST vs oracle SSIM energy on a CONVERGED small solve."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_EXACT)
w, h = 160, 110
rgb0, rgb1 = synth.make_rgb_pair(w, h)
cons = synth.make_constraints(w, h, 8)
prm = morph.Parameters()
prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = 30, 1.0, 32, capi.BCOND_BORDER
for c in cons:
    prm.add_point_pair(*c[:4], weight=float(c[4]))
pyr = morph.Pyramid(ctx); pyr.build_rgb(rgb0, rgb1, prm.start_res)
t = morph.MatchingThread(prm, pyr); t.start(); t.wait()
v = pyr._vector[0]
ex = int(0.1 * max(w, h))
e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
crops = {1: e1[ex:ex + h, ex:ex + w].copy(), 2: e0[ex:ex + h, ex:ex + w].copy()}
refs = {}
for side, ext in ((1, e0), (2, e1)):
    for otol in (1e-9, 1e-12):
        refs[(side, otol)] = O.poisson_extend(ext, w, h, ex, crops[side], v, side, tol=otol, max_it=200000)[0]
    d = np.abs(refs[(side, 1e-9)][..., :3].astype(int) - refs[(side, 1e-12)][..., :3].astype(int))
    print("side %d: oracle CG 1e-9 vs 1e-12: max %d, >0: %.5f" % (side, d.max(), (d > 0).mean()))
fr = morph.Frame(ctx, w, h, ex)
for tol in (1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
    fr.upload(e0, e1, v, None)
    for side in (1, 2):
        it, rr, ms = fr.poisson_extend(side, tol=tol)
        out = fr.download_ext(side)
        for otol in (1e-9, 1e-12):
            d = np.abs(out[..., :3].astype(int) - refs[(side, otol)][..., :3].astype(int))
            print("device tol %g side %d (%d its, resid %.2e) vs oracle %g: max %d, >1: %.6f, >0: %.4f" % (tol, side, it, rr, otol, d.max(), (d > 1).mean(), (d > 0).mean()))
# (b)
def gpu_solve(mode, rev, w2, h2, i0, i1, v0, P):
    ctx.set_math_mode(mode); ctx.set_commit_order(rev)
    kp = capi.KernParams()
    for f, _ in capi.KernParams._fields_:
        setattr(kp, f, getattr(P, f))
    ctx.set_params(kp)
    p2 = morph.Pyramid(ctx); p2.build_levels([(w2, h2), ((w2 + 1) // 2, (h2 + 1) // 2)])
    p2.upload_luma(1, i0, i1); p2[1].v = v0
    capi.check(p2._L.vm_init_level(p2._h, 0, w2, h2, None, 0))
    pr = capi.Progress()
    capi.check(p2._L.vm_optimize_level(p2._h, 0, 2000.0, None, 0, C.byref(pr)))
    ctx.set_commit_order(0)
    return p2[1].v, pr.iters
def total_energy(w2, h2, i0, i1, v, P):
    lg = O.Level(w2, h2); lg.set_images(i0, i1); lg.field("v")[...] = v; lg.init(0.0)
    e = lg.energy(P)
    return P.w_ssim * e[0] / (w2 * h2) + P.w_tps * e[1], e[0]
for (w2, h2) in ((160, 120), (128, 96), (200, 150)):
    i0, i1 = synth.make_pair(w2, h2)
    v0 = (0.8 * synth.displacement(w2, h2) + 0.05 * np.random.RandomState(0).randn(h2, w2, 2)).astype(np.float32)
    P = O.default_params()
    runs = {"exact": gpu_solve(capi.MATH_EXACT, 0, w2, h2, i0, i1, v0, P), "exact_rev": gpu_solve(capi.MATH_EXACT, 1, w2, h2, i0, i1, v0, P),
            "fast": gpu_solve(capi.MATH_FAST, 0, w2, h2, i0, i1, v0, P)}
    E = {k: total_energy(w2, h2, i0, i1, v, P) for k, (v, it) in runs.items()}
    rms = lambda a, b: float(np.sqrt(((runs[a][0] - runs[b][0]) ** 2).sum(-1).mean()))
    print("%dx%d converged (its %s): total energy exact %.6f rev %.6f fast %.6f -> floor |rev-exact| %.3f %%, |fast-exact| %.3f %%; ssim sums %s; rms dv floor %.4f fast %.4f" % (
        w2, h2, {k: it for k, (v, it) in runs.items()}, E["exact"][0], E["exact_rev"][0], E["fast"][0], 100 * abs(E["exact_rev"][0] - E["exact"][0]) / E["exact"][0],
        100 * abs(E["fast"][0] - E["exact"][0]) / E["exact"][0], {k: round(float(e[1]), 3) for k, e in E.items()}, rms("exact", "exact_rev"), rms("exact", "fast")))
ctx.set_math_mode(capi.MATH_FAST)
for (w2, h2) in ():
    i0, i1 = synth.make_pair(w2, h2)
    v0 = (0.8 * synth.displacement(w2, h2) + 0.05 * np.random.RandomState(0).randn(h2, w2, 2)).astype(np.float32)
    P = O.default_params()
    lo = O.Level(w2, h2); lo.set_images(i0, i1); lo.field("v")[...] = v0; lo.init(0.0)
    it_o = lo.optimize(P, 2000)
    kp = capi.KernParams()
    for f, _ in capi.KernParams._fields_:
        setattr(kp, f, getattr(P, f))
    ctx.set_params(kp)
    p2 = morph.Pyramid(ctx); p2.build_levels([(w2, h2), ((w2 + 1) // 2, (h2 + 1) // 2)])
    p2.upload_luma(1, i0, i1); p2[1].v = v0
    capi.check(p2._L.vm_init_level(p2._h, 0, w2, h2, None, 0))
    pr = capi.Progress()
    capi.check(p2._L.vm_optimize_level(p2._h, 0, 2000.0, None, 0, C.byref(pr)))
    eo, eg = (1 - lo.field("value")).sum(), (1 - p2[1].field("value")).sum()
    e0_ = lo.energy(P)
    lg = O.Level(w2, h2); lg.set_images(i0, i1); lg.field("v")[...] = p2[1].v; lg.init(0.0)
    e1_ = lg.energy(P)
    tot = lambda e: P.w_ssim * e[0] / (w2 * h2) + P.w_tps * e[1]
    dv = np.sqrt(((lo.field("v") - p2[1].v) ** 2).sum(-1))
    print("%dx%d converged: oracle %d its (improving stops), FAST %d its; SSIM energy %.4f vs %.4f (%.3f %%), total energy %.6f vs %.6f (%.3f %%), rms dv %.4f px, within 0.25 px %.4f" % (
        w2, h2, it_o, pr.iters, eo, eg, 100 * abs(eo - eg) / eo, tot(e0_), tot(e1_), 100 * abs(tot(e0_) - tot(e1_)) / tot(e0_), np.sqrt((dv ** 2).mean()), (dv < 0.25).mean()))
