"""round-5 experiment: k_render_win (v window in LDS) against the plain gather kernel (VM_RENDER=plain: the only value vm_render.hip recognises): ms per 1080p frame at
nine in-between positions, with and without a quadratic path in the frame; the outputs must agree byte for byte (run
twice, once per mode: the hash printed at the end is the comparison)"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

w, h, ex = 1920, 1080, 192
ctx = morph.Context(0, capi.MATH_FAST)
i0, i1 = synth.make_pair(w, h)
prm = morph.Parameters(); prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
morph.Morph(prm, pyr).calculate_halfway_parametrization()
rgb0, rgb1 = synth.make_rgb_pair(w, h)
fr = morph.Frame(ctx, w, h, ex)
fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), None, None)
fr.set_v_from_level(pyr, 1)
hh = hashlib.sha256()
for label, qp in (("no path", None), ("with path", (0.25 * synth.displacement(w, h)).astype(np.float32))):
    if qp is not None:
        fr.upload(None, None, None, qp)
    fr.render_halfway_dev(0.5, 0.5, 1)
    ms = []
    for rep in range(3):
        ms = [fr.render_halfway_dev(0.5, 0.1 * k, 1) for k in range(1, 10)]
    for k in (1, 3, 5, 9):
        hh.update(fr.render_halfway(0.5, 0.1 * k, 1).tobytes())
    print(os.environ.get("VM_RENDER", "win"), label, "ms per frame %.4f" % (sum(ms) / len(ms)), " ".join("%.3f" % m for m in ms))
print("sha256 of 8 rendered frames:", hh.hexdigest()[:24])
