"""dev: which tile groups of a PASS iteration ran and how many candidates they had, against the mask state (build with VM_DEFS=-DVM_PASS_DEBUG)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_EXACT)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 120, 68
i0, i1 = synth.make_pair(w, h)
v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(3).randn(h, w, 2)).astype(np.float32)
pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), (60, 34)])
pyr.upload_luma(1, i0, i1); pyr[1].v = v0
capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
ctx.set_tuning(capi.SWEEP_STEP, 0, 0)
capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(sys.argv[1]) if len(sys.argv) > 1 else 100.0, None, 1, None))
m = pyr[1].field("impmask")
b = np.zeros((h, w), bool)
for y in range(h):
    for x in range(w):
        b[y, x] = (m[y // 5 + 1, x // 5 + 1] >> ((x % 5) + (y % 5) * 5)) & 1
buf = (C.c_uint8 * 2048)()
capi.check(ctx._L.vm_dbg_pass_placement(ctx._h, buf, 2048))
ctx.set_tuning(capi.SWEEP_PASS, 0, 0)
pr = capi.Progress()
capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 1, C.byref(pr)))
print("PASS iteration: candidates", pr.candidates, "commits", pr.commits)
import ctypes
raw = (C.c_uint32 * 2048)()
# read the raw words through the byte interface is lossy: use hip memcpy via a second entry? -> reuse: low byte only
capi.check(ctx._L.vm_dbg_pass_placement(ctx._h, buf, 2048))
lo = np.frombuffer(buf, dtype=np.uint8).reshape(8, 32, 8).astype(int)     # [launch][part][group]: n_cand of the workgroup, 255 = group not live
for l in range(4):
    print("launch (pass) %d: per group (live, candidates):" % l, [("-" if (lo[l, :, g] == 255).all() else ("MIX" if (lo[l, :, g] == 255).any() else int(lo[l, :, g].sum()))) for g in range(8)])
for ty in range(4):
    for tx in range(2):
        ox, oy = tx * 69, ty * 21
        x0, x1 = max(ox - 2, 0), min(ox + 65, w - 1); y0, y1 = max(oy - 2, 0), min(oy + 17, h - 1)
        hits = sum(b[max(y - 2, 0):y + 3, max(x - 2, 0):x + 3].any() for y in range(oy, min(oy + 16, h)) for x in range(ox, min(ox + 64, w)) if ((x - ox) & 1) == 0 and ((y - oy) & 1) == 0)
        print("pass 0 tile", (tx, ty), "bits in tile+-2:", int(b[y0:y1 + 1, x0:x1 + 1].sum()), "phase-0 hits", hits)
