import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from videomorphing_amd import capi, morph, synth
import oracle
ctx = morph.Context(0, capi.MATH_FAST)
for (w, h, ex) in ((1, 1698, 1), (2, 3396, 2), (1698, 1, 1), (3, 1700, 1), (5, 1020, 0)):
    rng = np.random.RandomState(w + h)
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=2)
    v = (0.3 * rng.randn(h, w, 2)).astype(np.float32)
    e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
    if ex == 0:
        e0[100:140, 1:3, 3] = 255
        e1[500:520, :, 3] = 255
    fr = morph.Frame(ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    (i1, r1), (i2, r2), ms = fr.poisson_extend_both(tol=1e-6)
    worst = 0
    for side, ext, other in ((1, e0, e1), (2, e1, e0)):
        ref, _, _ = oracle.poisson_extend(ext, w, h, ex, other[ex:ex + h, ex:ex + w].copy(), v, side, tol=1e-9)
        out = fr.download_ext(side)
        worst = max(worst, int(np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int)).max()), int(out[..., 3].max()))
    print(w, h, ex, "iterations", i1, i2, "residual %.1e %.1e" % (r1, r2), "ms %.2f" % ms, "worst colour diff / alpha", worst, flush=True)
    fr.close()
