"""dev: what does one iteration of a CYCLING finest level cost, and what is in it?  A frame whose 1080p level keeps
exchanging rounding-level moves (FAST): the level's first 100 iterations, then 400 more on their own -- tile visits,
line searches, commits and evaluations per iteration of the cycling regime, time per iteration; with a -DVM_PROF
build also the stage stamps of the last tile visit and of the last pass of the SPARSE kernel.
usage: tools/dev_cycling.py [frame ...]"""
import ctypes as C
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
    capi.check(L.vm_optimize_level(p._h, 0, 100.0, None, 1, C.byref(pr)))
    print("frame %d: first 100 iterations: live %d, %.1f ms" % (f, pr.iters_live, pr.elapsed_ms))
    if pr.iters_live < 100:
        print("   (the level converged: not a cycling frame)")
        p.clear()
        continue
    pr = capi.Progress()
    capi.check(L.vm_optimize_level(p._h, 0, 400.0, None, 1, C.byref(pr)))
    n = max(pr.iters_live, 1)
    print("   next 400: live %d, %.1f ms = %.1f us per iteration; per iteration: tile visits %.2f, line searches %.1f, commits %.2f, evaluations %.0f; "
          "sched ms %s launches %s" % (pr.iters_live, pr.elapsed_ms, pr.elapsed_ms * 1e3 / n, pr.active_tiles / n, pr.candidates / n, pr.commits / n,
                                       pr.evaluations / n, [round(x, 1) for x in pr.sched_ms], list(pr.sched_launches)))
    if hasattr(L, "vm_dbg_prof_read"):
        try:
            buf = np.zeros(512 * 16 * 2, np.uint64)
            L.vm_dbg_prof_read.argtypes = [C.c_void_p, C.c_size_t]
            if L.vm_dbg_prof_read(buf.ctypes.data, buf.nbytes) == 0:
                st = buf[:32].reshape(4, 8).astype(np.int64)
                sf = buf[8192 + 512:8192 + 512 + 16].astype(np.int64)
                us = lambda a, b: (b - a) / 100.0
                print("   last tile visit: entry -> LDS %.2f us; save %.2f; total %.2f" % (us(sf[0], sf[1]), us(sf[2], sf[3]), us(sf[0], sf[3])))
                # stamps of a phase: 0 start, 1 candidates compacted, 6 pixel context + window loaded (lean pieces only), 2 search
                # done (wave 0), 3 barrier, 4 commits counted, 5 gather done + barrier
                for k in range(4):
                    print("     phase %d: total %.2f = candidates %.2f + context %.2f + search %.2f + barrier %.2f + commits %.2f + gather %.2f" % (
                        k, us(st[k][0], st[k][5]), us(st[k][0], st[k][1]), us(st[k][1], st[k][6]) if st[k][6] > st[k][1] else float("nan"),
                        us(max(st[k][6], st[k][1]), st[k][2]), us(st[k][2], st[k][3]), us(st[k][3], st[k][4]), us(st[k][4], st[k][5])))
                print("   last pass of the kernel: tile selection %.2f us, sweeps %.2f, list update %.2f" % (us(sf[4], sf[5]), us(sf[5], sf[6]), us(sf[6], sf[7])))
        except AttributeError:
            pass
    p.clear()
