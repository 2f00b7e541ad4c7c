"""dev: how many of the first N frames end with a cycling finest level (all 500 sweeps executed), FAST, config[1];
and the mean solve time.  usage: tools/dev_count_cycling.py [N=40]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
cyc, ms = [], []
for f in range(N):
    i0, i1 = synth.make_pair(w, h, frame=f)
    p = morph.Pyramid(ctx)
    p.build(i0, i1, 32)
    nl = p.size() - 1
    prog = (capi.Progress * (nl - 1))()
    ctx.sync(); t = time.perf_counter()
    capi.check(p._L.vm_solve(p._h, 500.0, 1.0, None, 0, None, 1, prog))
    ctx.sync(); ms.append((time.perf_counter() - t) * 1e3)
    if prog[0].iters_live >= 500:
        cyc.append(f)
    p.clear()
print("frames 0..%d: %d cycling %s; mean %.1f ms, mean of the converging ones %.1f ms" % (
    N - 1, len(cyc), cyc, sum(ms) / len(ms), sum(m for f, m in enumerate(ms) if f not in cyc) / max(1, N - len(cyc))))
