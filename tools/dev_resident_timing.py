"""dev: what the resident visits of the sparse kernel buy on cycling frames -- config[1] solves (FAST, automatic schedule,
reference stopping rule) with list-driven visits only (vm_dbg_sparse_resident 1) and with resident visits (0): ms per solve
(best of 3) and the finest level's share.  usage: tools/dev_resident_timing.py [frame ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

frames = [int(a) for a in sys.argv[1:]] or [6, 9, 10, 11, 0]
ctx = morph.Context(0, capi.MATH_FAST)
prm = morph.Parameters()
prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
ctx.set_params(morph.KernParameters(prm))
w, h = 1920, 1080
for f in frames:
    i0, i1 = synth.make_pair(w, h, frame=f)
    row = []
    for mode in (1, 0):
        ctx.set_sparse_resident(mode)
        best, lvl, its = 1e9, 0.0, 0
        for rep in range(3):
            pyr = morph.Pyramid(ctx)
            pyr.build(i0, i1, 32)
            prog = (capi.Progress * 5)()
            ctx.sync()
            t0 = time.perf_counter()
            capi.check(pyr._L.vm_solve(pyr._h, 500.0, 1.0, None, 0, None, 0, prog))
            ctx.sync()
            ms = (time.perf_counter() - t0) * 1e3
            if ms < best:
                best, lvl, its = ms, float(prog[0].elapsed_ms), int(prog[0].iters)
            pyr.clear()
        row.append((best, lvl, its))
    print("frame %2d: list-driven %.1f ms (finest level %.1f ms, %d iterations) | resident %.1f ms (finest level %.1f ms, %d iterations)"
          % (f, row[0][0], row[0][1], row[0][2], row[1][0], row[1][1], row[1][2]), flush=True)
ctx.set_sparse_resident(0)
