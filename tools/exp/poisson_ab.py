"""round-5 experiment: Poisson extension of 1080p frames (ex = 192, tol 1e-5) -- ms per frame for one frame per batch and
four, and a hash of the extended canvases: run once per library (VM_LIB_PATH) to compare two builds bit for bit"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

w, h, ex = 1920, 1080, 192
ctx = morph.Context(0, capi.MATH_FAST)
rgb0, rgb1 = synth.make_rgb_pair(w, h)
e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
v = (synth.displacement(w, h)).astype(np.float32)
frs = [morph.Frame(ctx, w, h, ex) for _ in range(4)]
hh = hashlib.sha256()
for n in (1, 4):
    for rep in range(3):
        for f in frs[:n]:
            f.upload(e0, e1, v, None)
        res, ms = morph.poisson_extend_frames(frs[:n], tol=1e-5)
    print("frames per batch %d: %.3f ms per frame, iterations %s" % (n, ms / n, [r[0][0] for r in res]))
    for f in frs[:n]:
        hh.update(f.download_ext(1).tobytes()); hh.update(f.download_ext(2).tobytes())
qp = frs[0].quadratic_path(tol=1e-4)
hh.update(frs[0].download_qpath().tobytes())
print("quadratic path", qp)
print("sha256:", hh.hexdigest()[:24])
