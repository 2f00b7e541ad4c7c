"""dev: the 120x68 level of n 1080p pairs in ONE launch per phase / pass (one stream), n = 1 .. 30, under the
STEP, TILE and PASS schedules: us per phase and per pair-phase.  Tells whether the small-batch regime (the 7-8
pairs per GPU of an 8-GPU run of config[2]) is bound by the latency of one line search or by the chip's VALU
throughput (time growing with n).  usage: tools/dev_step_scaling.py [iters] [level: 1 = 120x68, 2 = 240x135]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
frames = [synth.make_pair(w, h, frame=k) for k in range(4)]
for n in (1, 2, 4, 8, 16, 30):
    for name, sched in (("step", capi.SWEEP_STEP), ("tile", capi.SWEEP_TILE), ("pass", capi.SWEEP_PASS)):
        if sched == capi.SWEEP_PASS and n > 8:
            continue
        pyrs = []
        for k in range(n):
            p = morph.Pyramid(ctx)
            p.build(frames[k % 4][0], frames[k % 4][1], 32)
            pyrs.append(p)
        L = pyrs[0]._L
        nl = pyrs[0].size() - 1
        el = nl - back                      # python level index of the level to time
        ctx.set_tuning(0, 0, 0)
        for p in pyrs:                      # the state the solver reaches this level in
            capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
            for e in range(nl - 1, el, -1):
                capi.check(L.vm_upsample_v(p._h, e - 1, e))
                capi.check(L.vm_init_level(p._h, e - 1, w, h, None, 0))
                capi.check(L.vm_optimize_level(p._h, e - 1, 500.0, None, 1, None))
            capi.check(L.vm_upsample_v(p._h, el - 1, el))
            capi.check(L.vm_init_level(p._h, el - 1, w, h, None, 0))
        ctx.set_tuning(sched, 0, 0)
        arr = (C.c_void_p * n)(*[p._h for p in pyrs])
        prog = (capi.Progress * n)()
        capi.check(L.vm_optimize_level_batch(arr, n, el - 1, float(iters), None, 1, prog))
        pr = prog[0]
        lv = pyrs[0][el]
        print("%-4s n=%2d %dx%d: %8.2f ms per %d iterations, %7.2f us per phase, %6.2f us per pair-phase; cand/iter/pair %.0f, launches %d" % (
            name, n, lv.width, lv.height, pr.elapsed_ms, pr.iters, pr.elapsed_ms * 1e3 / pr.iters / 16, pr.elapsed_ms * 1e3 / pr.iters / 16 / n,
            pr.candidates / pr.iters, pr.launches), flush=True)
        for p in pyrs:
            p.clear()
ctx.set_tuning(0, 0, 0)
