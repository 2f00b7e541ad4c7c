import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
for f in (1, 4, 5, 6, 9, 10, 11, 22, 23):
    i0, i1 = synth.make_pair(w, h, frame=f)
    p = morph.Pyramid(ctx); p.build(i0, i1, 32)
    L = p._L; nl = p.size() - 1
    capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
    for e in range(nl - 1, 1, -1):
        capi.check(L.vm_upsample_v(p._h, e - 1, e)); capi.check(L.vm_init_level(p._h, e - 1, w, h, None, 0))
        capi.check(L.vm_optimize_level(p._h, e - 1, 500.0, None, 1, None))
    capi.check(L.vm_upsample_v(p._h, 0, 1)); capi.check(L.vm_init_level(p._h, 0, w, h, None, 0))
    pr = capi.Progress()
    capi.check(L.vm_optimize_level(p._h, 0, 100.0, None, 1, C.byref(pr)))
    if pr.iters_live < 100:
        print("frame", f, "converged after", pr.iters_live); p.clear(); continue
    out = []
    for k in range(6):
        capi.check(L.vm_optimize_level(p._h, 0, 50.0, None, 1, None))
        m = p[1].field("impmask")
        ys, xs = np.nonzero(m)
        bits = [(5 * (x - 1) + b % 5, 5 * (y - 1) + b // 5) for y, x in zip(ys, xs) for b in range(25) if (int(m[y, x]) >> b) & 1]
        bx = [q[0] for q in bits]; by = [q[1] for q in bits]
        out.append("words %d bits %d bbox x %d..%d y %d..%d (%dx%d px)" % (len(ys), len(bits), min(bx), max(bx), min(by), max(by), max(bx) - min(bx) + 1, max(by) - min(by) + 1))
    print("frame", f, "cycling:", " | ".join(out[::2]), flush=True)
    p.clear()
