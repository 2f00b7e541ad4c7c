"""dev: per-level, per-chunk activity and timing of the bench workload (1080p, fast)"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
mode = capi.MATH_FAST if (len(sys.argv) < 2 or sys.argv[1] == "fast") else capi.MATH_EXACT
ctx = morph.Context(0, mode)
ctx.set_params(morph.KernParameters(morph.Parameters()))
ctx.set_tuning(int(os.environ.get("VM_SCHED", "0")), 0, 0)
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
L = pyr._L
nl = pyr.size() - 1
capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))
tot = 0
for el in range(nl - 1, 0, -1):
    capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
    capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
    lv = pyr[el]
    done = 0
    print("level %dx%d" % (lv.width, lv.height))
    for chunk in (1, 1, 2, 4, 8, 16, 32, 64, 128, 244):
        pr = capi.Progress()
        capi.check(L.vm_optimize_level(pyr._h, el - 1, float(chunk), None, 0, C.byref(pr)))
        done += pr.iters
        tot += pr.elapsed_ms
        print("  iters %3d..%3d: %8.3f ms/iter  tiles/launch %7.1f cand/iter %9.0f commits/iter %8.0f improving %d" % (
            done - pr.iters, done, pr.elapsed_ms / max(pr.iters, 1), pr.active_tiles / max(pr.iters * 4, 1),
            pr.candidates / max(pr.iters, 1), pr.commits / max(pr.iters, 1), pr.improving))
        if not pr.improving or pr.iters < chunk:
            break
print("total kernel ms", tot)
d = synth.displacement(w, h)
v = pyr[1].v
print("rms err vs ground truth d: %.3f px" % np.sqrt(((v - d) ** 2).sum(-1).mean()))
