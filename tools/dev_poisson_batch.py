"""dev: the Poisson extension at 1080p (ex = 192), ms per frame (both sides, tol 1e-5): one side at a time, both sides
as one batch, 2 / 4 frames as one batch; iteration counts; byte agreement between the forms.
usage: tools/dev_poisson_batch.py [tol]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
ctx = morph.Context(0, capi.MATH_FAST)
w, h = 1920, 1080
ex = int(0.1 * max(w, h))
nf = 4
inputs = []
for k in range(nf):
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=k)
    v = (synth.displacement(w, h) + 0.25 * k).astype(np.float32)
    inputs.append((morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v))
frs = [morph.Frame(ctx, w, h, ex) for _ in range(nf)]


def load():
    for f, (e0, e1, v) in zip(frs, inputs):
        f.upload(e0, e1, v, None)
    ctx.sync()


def timed(fn):
    load()
    fn()                                    # workspaces
    load()
    t0 = time.perf_counter()
    r = fn()
    ctx.sync()
    return r, (time.perf_counter() - t0) * 1e3


solver = "mgb (batched, ring-only)"
r, ms = timed(lambda: [(f.poisson_extend(1, tol=tol), f.poisson_extend(2, tol=tol)) for f in frs])
print("%s: one side at a time: %.2f ms per frame (wall), device ms per side %s, iterations %s" % (
    solver, ms / nf, [round(a[2], 2) for p in r for a in p][:4], [a[0] for p in r for a in p]))
ref = [(f.download_ext(1), f.download_ext(2)) for f in frs]
if True:
    for nb in (1, 2, 4):
        r, ms = timed(lambda: [morph.poisson_extend_frames(frs[k:k + nb], tol=tol) for k in range(0, nf, nb)])
        its = [s[0] for call in r for fr_ in call[0] for s in fr_]
        dmax = max(int(np.abs(f.download_ext(s).astype(int) - ref[i][s - 1].astype(int)).max()) for i, f in enumerate(frs) for s in (1, 2))
        print("batches of %d frame(s) x 2 sides: %.2f ms per frame (wall), device ms per batch %s, iterations %s, max colour diff vs one-at-a-time %d"
              % (nb, ms / nf, [round(call[1], 2) for call in r], its, dmax))
