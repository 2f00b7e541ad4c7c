import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from videomorphing_amd import capi, morph, synth
import fullsize_fixture as FX
ctx = morph.Context(0, capi.MATH_FAST)
for (w, h) in ((480, 270), (960, 540), (1920, 1080)):
    for name, v in (("fixture", FX.field(w, h, 0)), ("rough", (synth.displacement(w, h) + 0.3 * np.random.RandomState(1).randn(h, w, 2)).astype(np.float32))):
        fr = morph.Frame(ctx, w, h, 8)
        fr.upload(None, None, v, None)
        for tol in (1e-4, 3e-5):
            try:
                rs = [fr.quadratic_path(tol=tol, max_it=64) for _ in range(3)]
                print(w, h, name, tol, "converged", rs[-1][0], "%.2e" % rs[-1][1], "ms", ["%.2f" % r[2] for r in rs])
            except capi.VmError as e:
                print(w, h, name, tol, "FAILED", str(e)[-70:])
        fr.close()
