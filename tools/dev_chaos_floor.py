"""dev: how far do two equally legal EXACT trajectories drift apart at config[1] (1080p, 6 levels,
500 iterations per level, reference stopping rule), and where does FAST sit relative to that?
EXACT with the commits of a phase folded row-major (the oracle's order) vs reversed; FAST under
the automatic schedule; energies from the oracle's vmo_energy on the host."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth  # noqa: E402
import oracle as O  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
iters = float(sys.argv[3]) if len(sys.argv) > 3 else 500.0
frame = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = morph.Context(0, capi.MATH_EXACT)
i0, i1 = synth.make_pair(w, h, frame=frame)
prm = morph.Parameters()
prm.max_iter, prm.max_iter_drop_factor, prm.start_res = iters, 1.0, 32
ctx.set_params(morph.KernParameters(prm))


def energy(v):
    lv = O.Level(w, h)
    lv.set_images(i0, i1)
    lv.field("v")[...] = v
    lv.init(0.0)
    P = O.default_params()
    e = lv.energy(P)
    return float(P.w_ssim * e[0] / (w * h) + P.w_tps * e[1]), e


out, its = {}, {}
for name, mode, rev in (("exact", capi.MATH_EXACT, 0), ("exact_rev", capi.MATH_EXACT, 1), ("fast", capi.MATH_FAST, 0)):
    ctx.set_math_mode(mode)
    ctx.set_commit_order(rev)
    pyr = morph.Pyramid(ctx)
    pyr.build(i0, i1, 32)
    m = morph.Morph(prm, pyr)
    t = time.time()
    m.calculate_halfway_parametrization()
    out[name] = pyr[1].v
    its[name] = [m.progress[el]["iters"] for el in sorted(m.progress)]
    print(name, "%.2f s" % (time.time() - t), "iters fine->coarse", its[name], flush=True)
ctx.set_commit_order(0)
d = synth.displacement(w, h)
for a, b in (("exact", "exact_rev"), ("exact", "fast"), ("exact_rev", "fast")):
    dv = np.sqrt(((out[a] - out[b]) ** 2).sum(-1))
    print("%s vs %s: RMS dv %.4f px, within 0.25 px %.4f, max %.2f" % (a, b, np.sqrt((dv ** 2).mean()), (dv < 0.25).mean(), dv.max()))
for k in out:
    e, parts = energy(out[k])
    print("%s: energy %.6f (ssim %.4f tps %.4f)  RMS error vs ground truth %.4f" % (k, e, parts[0], parts[1], np.sqrt(((out[k] - d) ** 2).sum(-1).mean())))
