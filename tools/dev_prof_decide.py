"""dev: stage stamps of the last k_decide launch (needs a build with VM_DEFS=-DVM_PROF)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
ctx.set_tuning(int(os.environ.get("VM_SCHED", "2")), int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
L = pyr._L
nl = pyr.size() - 1
capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))
el = nl - 1
capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
pr = capi.Progress()
capi.check(L.vm_optimize_level(pyr._h, el - 1, 20.0, None, 1, C.byref(pr)))
print("ms/iter %.3f launches %d cand/iter %.0f" % (pr.elapsed_ms / pr.iters, pr.launches, pr.candidates / pr.iters))
buf = np.zeros((512, 16), np.uint64)
L.vm_dbg_prof_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.vm_dbg_prof_read(buf.ctypes.data, buf.nbytes) == 0
ok = buf[:, 0] > 0
b = buf[ok].astype(np.int64)
t0 = b[:, 0].min()
names = ["entry", "mask", "tables", "compact+..ctx", "nb_load", "gradient", "foldover", "golden", "end"]
print("workgroups with stamps:", ok.sum(), " Lf", np.unique(b[:, 9]), " n_mine", np.unique(b[:, 10]))
print("entry skew (us): max %.2f" % ((b[:, 0].max() - t0) / 100.0))
for k in range(1, 9):
    valid = b[:, k] > 0
    prev = k - 1
    d = (b[valid, k] - b[valid, prev]) / 100.0
    print("%-14s  n=%3d  mean %6.2f us  min %6.2f  max %6.2f   | since first entry: mean %6.2f max %6.2f" % (
        names[k], valid.sum(), d.mean(), d.min(), d.max(), ((b[valid, k] - t0) / 100.0).mean(), ((b[valid, k] - t0) / 100.0).max()))
