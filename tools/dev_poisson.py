"""dev: Poisson extension + render at 1080p (config 5 pieces)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
ex = int(0.1 * max(w, h))
rgb0, rgb1 = synth.make_rgb_pair(w, h)
v = synth.displacement(w, h).astype(np.float32)
fr = morph.Frame(ctx, w, h, ex)
e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
for tol in (1e-3, 1e-4, 1e-5):
    fr.upload(e0, e1, v, None)
    t = time.time()
    r1 = fr.poisson_extend(1, tol=tol, max_it=20000)
    r2 = fr.poisson_extend(2, tol=tol, max_it=20000)
    print("tol %g: side1 it %d res %.2e %.1f ms | side2 it %d res %.2e %.1f ms | wall %.3f s" % (tol, *r1, *r2, time.time() - t))
out = fr.download_ext(1)
print("ext1 stats: outside mean", out[:ex].mean(), "alpha max", out[..., 3].max())
ms = [fr.render_halfway_dev(0.5, 0.1 * k, 1) for k in range(1, 10)]
print("render ms", np.mean(ms))
