"""dev: WHERE does FAST leave the legal family?  config[1] solves (reference stopping rule) in which exactly one
level -- or everything but one -- runs in FAST arithmetic and the rest in EXACT; final-field RMS against the all-EXACT
run, next to the all-FAST run and to two members of the legal family (another commit order, REF_FASTMATH).
usage: tools/dev_fast_bisect.py [frame ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

frames = [int(a) for a in sys.argv[1:]] or [3, 9, 15]
ctx = morph.Context(0, capi.MATH_EXACT)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
SCHED = {"auto": capi.SWEEP_AUTO, "tile": capi.SWEEP_TILE}


def solve(i0, i1, mode_of_level, order=0, sched="auto", upmode=None):
    """mode_of_level(el) -> math mode of python level index el (nl - 1 = coarsest GPU level ... 1 = finest)"""
    p = morph.Pyramid(ctx)
    p.build(i0, i1, 32)
    L = p._L
    nl = p.size() - 1
    ctx.set_math_mode(capi.MATH_EXACT)
    ctx.set_commit_order(order)
    ctx.set_tuning(SCHED[sched], 0, 0)
    capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
    its = []
    for e in range(nl - 1, 0, -1):
        m = mode_of_level(e)
        ctx.set_math_mode(upmode(e) if upmode else m)
        capi.check(L.vm_upsample_v(p._h, e - 1, e))
        capi.check(L.vm_init_level(p._h, e - 1, w, h, None, 0))
        ctx.set_math_mode(m)
        pr = capi.Progress()
        capi.check(L.vm_optimize_level(p._h, e - 1, 500.0, None, 0, C.byref(pr)))
        its.append(pr.iters)
    v = p[1].v
    p.clear()
    ctx.set_math_mode(capi.MATH_EXACT)
    ctx.set_commit_order(0)
    ctx.set_tuning(0, 0, 0)
    return v, its


E, F, RF = capi.MATH_EXACT, capi.MATH_FAST, capi.MATH_REF_FASTMATH
for f in frames:
    i0, i1 = synth.make_pair(w, h, frame=f)
    ref, its = solve(i0, i1, lambda e: E)
    rms = lambda v: float(np.sqrt(((v - ref) ** 2).sum(-1).mean()))
    print("frame %d: all-EXACT iters (coarse->fine) %s" % (f, its), flush=True)
    runs = [("EXACT order 2", lambda e: E, dict(order=2)), ("REF_FASTMATH", lambda e: RF, {}), ("all FAST", lambda e: F, {}),
            ("all FAST, TILE schedule", lambda e: F, dict(sched="tile")),
            ("FAST sweeps, EXACT init/upsample", lambda e: F, dict(upmode=lambda e: E)),
            ("EXACT sweeps, FAST init/upsample", lambda e: E, dict(upmode=lambda e: F))]
    for k in range(5, 0, -1):
        runs.append(("only level %d FAST" % k, (lambda kk: (lambda e: F if e == kk else E))(k), {}))
    for k in range(5, 0, -1):
        runs.append(("all FAST but level %d" % k, (lambda kk: (lambda e: E if e == kk else F))(k), {}))
    for name, fn, kw in runs:
        v, it = solve(i0, i1, fn, **kw)
        print("  %-36s RMS vs all-EXACT %.4f   iters %s" % (name, rms(v), it), flush=True)
