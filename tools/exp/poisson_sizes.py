"""dev: the Poisson extension across canvas sizes (both sides as one batch): iterations, residual, ms -- a convergence check of the
cycle's sweeps-per-level choice away from the 1080p canvas it was tuned on (VM_MGB_NU overrides it per process)
usage: tools/exp/poisson_sizes.py [tol]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
ctx = morph.Context(0, capi.MATH_FAST)
for (w, h) in ((320, 180), (640, 360), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (1080, 1920), (4000, 300)):
    ex = int(0.1 * max(w, h))
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=3)
    v = synth.displacement(w, h).astype(np.float32)
    fr = morph.Frame(ctx, w, h, ex)
    out = []
    for rep in range(3):
        fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v, None)
        (i1, r1), (i2, r2), ms = fr.poisson_extend_both(tol=tol)
        out.append(ms)
    e = fr.download_ext(1)
    print("%5d x %4d ex %3d: iterations %2d / %2d, residual %.1e / %.1e, ms %s, alpha max %d, ring mean %.1f" % (
        w, h, ex, i1, i2, r1, r2, ["%.2f" % m for m in out], e[..., 3].max(), e[:ex, :, :3].mean()), flush=True)
    fr.close()
