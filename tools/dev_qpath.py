"""dev: quadratic path convergence at several sizes / inputs"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
for (w, h) in ((480, 270), (960, 540), (1920, 1080)):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    smooth = (synth.displacement(w, h) + np.stack([0.8 * np.sin(xx / 9.0), 0.6 * np.cos(yy / 5.0)], -1)).astype(np.float32)
    rough = (smooth + 0.3 * np.random.RandomState(1).randn(h, w, 2)).astype(np.float32)
    for name, v in (("smooth", smooth), ("rough", rough)):
        fr = morph.Frame(ctx, w, h, 8)
        fr.upload(None, None, v, None)
        for mi in (4, 8, 16, 32, 64):
            try:
                r = fr.quadratic_path(tol=1e-5, max_it=mi)
                print(w, h, name, "converged", r)
                break
            except capi.VmError as e:
                print(w, h, name, "max_it", mi, str(e)[-60:])
        fr.close()
