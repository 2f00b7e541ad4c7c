"""dev: is a cycling finest level PERIODIC, bit for bit?  From iteration 200 on, one iteration per call: the full state of the
level after every iteration (v, luma, mean, var, cross, value, tps_b, ui_b, impmask) hashed; the first period P for which
state(i) == state(i - P) holds for the rest of the window.  usage: tools/dev_cycle_period.py [frame ...]"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

frames = [int(a) for a in sys.argv[1:]] or [6, 9, 10]
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")
for f in frames:
    i0, i1 = synth.make_pair(w, h, frame=f)
    p = morph.Pyramid(ctx)
    p.build(i0, i1, 32)
    L = p._L
    nl = p.size() - 1
    capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
    for e in range(nl - 1, 1, -1):
        capi.check(L.vm_upsample_v(p._h, e - 1, e))
        capi.check(L.vm_init_level(p._h, e - 1, w, h, None, 0))
        capi.check(L.vm_optimize_level(p._h, e - 1, 500.0, None, 1, None))
    capi.check(L.vm_upsample_v(p._h, 0, 1))
    capi.check(L.vm_init_level(p._h, 0, w, h, None, 0))
    pr = capi.Progress()
    capi.check(L.vm_optimize_level(p._h, 0, 200.0, None, 1, C.byref(pr)))
    if pr.iters_live < 200:
        print("frame", f, "converged after", pr.iters_live)
        p.clear()
        continue
    hs, commits = [], []
    for k in range(48):
        pr = capi.Progress()
        capi.check(L.vm_optimize_level(p._h, 0, 1.0, None, 1, C.byref(pr)))
        lv = p[1]
        # only a strip around the active pixels can change: hash the bottom 24 rows (and check the rest once)
        hs.append(hashlib.sha1(b"".join(np.ascontiguousarray(lv.field(n)[-24:]).tobytes() for n in STATE[:-1]) +
                               np.ascontiguousarray(lv.field("impmask")).tobytes()).hexdigest())
        commits.append(int(pr.commits))
    period = None
    for P_ in range(1, 17):
        if all(hs[i] == hs[i - P_] for i in range(16, 48)):
            period = P_
            break
    print("frame %d: commits per iteration %s; state period over iterations 216..248: %s" % (f, commits[:12], period))
    p.clear()
