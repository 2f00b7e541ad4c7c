"""dev: activity of the finest level per frame (tiles / candidates / commits per iteration, schedule times)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
for frame in ([int(a) for a in sys.argv[1:]] or range(4)):
    i0, i1 = synth.make_pair(1920, 1080, frame=frame)
    p = morph.Pyramid(ctx); p.build(i0, i1, 32)
    nl = p.size() - 1
    prog = (capi.Progress * (nl - 1))()
    arr = (C.c_void_p * 1)(p._h)
    capi.check(p._L.vm_solve_batch(arr, 1, 500.0, 1.0, None, 1, prog))
    pr = prog[0]
    print("frame %d finest: iters %d (live %d)  ms %.1f  tiles/iter %.1f  cand/iter %.1f  commits/iter %.2f  evals/cand %.1f  sched_ms %s launches %s" % (
        frame, pr.iters, pr.iters_live, pr.elapsed_ms, pr.active_tiles / max(pr.iters_live, 1), pr.candidates / max(pr.iters_live, 1), pr.commits / max(pr.iters_live, 1),
        pr.evaluations / max(pr.candidates, 1), [round(x, 1) for x in pr.sched_ms], list(pr.sched_launches)))
