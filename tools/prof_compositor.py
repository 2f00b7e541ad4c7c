"""dev/profile workload: the compositor stages at 1080p -- Poisson extension of both sides
(1e-5), the quadratic path and 9 rendered in-between frames -- for rocprofv3 --kernel-trace."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

ctx = morph.Context(0, capi.MATH_FAST)
w, h = 1920, 1080
ex = int(0.1 * max(w, h))
rgb0, rgb1 = synth.make_rgb_pair(w, h)
v = synth.displacement(w, h).astype(np.float32)
fr = morph.Frame(ctx, w, h, ex)
e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
for rep in range(3):
    fr.upload(e0, e1, v, None)
    r1, r2, _ = fr.poisson_extend_both(tol=1e-5)
    try:
        qp = fr.quadratic_path(tol=1e-4)
    except capi.VmError as e:
        qp = str(e)
    ms = [fr.render_halfway_dev(0.1 * k, 0.1 * k, 1) for k in range(1, 10)]
print("poisson", r1, r2, "qpath", qp, "render ms", float(np.mean(ms)))
