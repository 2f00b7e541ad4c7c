"""dev/profile workload: the Poisson extension of FOUR 1080p frames per batch (eight systems, tol 1e-5: the shape
bench.py's config[4] pipeline runs), three batches, for rocprofv3 --kernel-trace (tools/prof_any.sh / prof_pmc.sh).
usage: tools/prof_poisson4.py [frames per batch] [tol]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
ctx = morph.Context(0, capi.MATH_FAST)
w, h = 1920, 1080
ex = int(0.1 * max(w, h))
inputs = []
for k in range(nf):
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=k)
    inputs.append((morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), (synth.displacement(w, h) + 0.25 * k).astype(np.float32)))
frs = [morph.Frame(ctx, w, h, ex) for _ in range(nf)]
for rep in range(3):
    for f, (e0, e1, v) in zip(frs, inputs):
        f.upload(e0, e1, v, None)
    res, ms = morph.poisson_extend_frames(frs, tol=tol)
print("frames per batch", nf, "tol", tol, "ms per frame", ms / nf, "iterations", [s[0] for r in res for s in r])
