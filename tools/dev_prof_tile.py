"""dev: stage stamps of active k_optimize tiles in the sparse regime (build with VM_DEFS=-DVM_PROF)"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 1920, 1080
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build(i0, i1, 32)
L = pyr._L
nl = pyr.size() - 1
capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))
for el in range(nl - 1, 0, -1):
    capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
    capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
    pr = capi.Progress()
    capi.check(L.vm_optimize_level(pyr._h, el - 1, 500.0 if el > 1 else float(sys.argv[1] if len(sys.argv) > 1 else 40.0), None, 0, C.byref(pr)))
    print("level", el, "iters", pr.iters, "ms/iter %.3f" % (pr.elapsed_ms / pr.iters))
buf = np.zeros((512, 16), np.uint64)
L.vm_dbg_prof_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.vm_dbg_prof_read(buf.ctypes.data, buf.nbytes) == 0
b = buf[buf[:, 0] > 0].astype(np.int64)
recent = b[b[:, 0] > b[:, 0].max() - 100000]      # stamps of the last millisecond
print("active tiles stamped recently:", len(recent))
tot = (b[:, 14] - b[:, 0]) / 100.0
for lo, hi in ((0, 64), (64, 256), (256, 600), (600, 900), (900, 1025)):
    m = (b[:, 15] >= lo) & (b[:, 15] < hi)
    if m.any():
        dec = sum((b[m, 3 + 3 * p] - b[m, 2 + 3 * p]) for p in range(4)) / 100.0
        com = sum((b[m, 4 + 3 * p] - b[m, 3 + 3 * p]) for p in range(4)) / 100.0
        print("cand %4d..%4d: %3d tiles, total %6.1f us (line searches %6.1f, commits %5.1f, load %4.1f)" % (lo, hi, m.sum(), tot[m].mean(), dec.mean(), com.mean(), ((b[m, 1] - b[m, 0]) / 100.0).mean()))
for r in recent[:6]:
    t = (r[:15] - r[0]) / 100.0
    ph = ["  ph%d: compact %.2f decide %.2f commit %.2f" % (p, t[2 + 3 * p] - (t[1] if p == 0 else t[1 + 3 * p]), t[3 + 3 * p] - t[2 + 3 * p], t[4 + 3 * p] - t[3 + 3 * p]) for p in range(4)]
    print("cand %d  load %.2f |%s | save %.2f  total %.2f us" % (r[15], t[1], "".join(ph), t[14] - t[13], t[14]))
