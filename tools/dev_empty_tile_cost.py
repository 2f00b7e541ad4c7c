"""dev: what a TILE launch costs when (nearly) every workgroup finds no set mask bit near its tile -- the early-out path of
k_optimize: n solved 1080p pairs, then 64 more fixed-work iterations of the finest level (and of the 960x540 one) under the
forced TILE schedule; us per launch.  usage: tools/dev_empty_tile_cost.py [pairs ...]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [1, 8, 30]
ctx = morph.Context(0, capi.MATH_FAST)
prm = morph.Parameters()
prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
ctx.set_params(morph.KernParameters(prm))
w, h = 1920, 1080
for n in ns:
    batch = []
    for k in range(n):
        i0, i1 = synth.make_pair(w, h, frame=k % 8)
        p = morph.Pyramid(ctx)
        p.build(i0, i1, 32)
        batch.append(p)
    morph.solve_batch(batch, 500, 1.0, fixed_work=False)
    L = batch[0]._L
    arr = (C.c_void_p * n)(*[p._h for p in batch])
    ctx.set_tuning(capi.SWEEP_TILE, 0, 0)
    for lvl, name in ((0, "1920x1080"), (1, "960x540")):
        try:
            t0 = time.perf_counter()
            prog = (capi.Progress * n)()
            capi.check(L.vm_optimize_level_batch(arr, n, lvl, 64.0, None, 1, prog))
            ctx.sync()
            dt = time.perf_counter() - t0
            print("%2d pairs, %s: %.1f us per TILE launch (64 iterations x 4 passes; line searches per iteration and pair: %.1f)" % (
                n, name, dt * 1e6 / 256, sum(prog[k].candidates for k in range(n)) / 64.0 / n), flush=True)
        except Exception as e:
            print("level", name, "failed:", e)
    ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    for p in batch:
        p.clear()
