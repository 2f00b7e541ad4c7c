"""dev: stage stamps of the last k_pass launch on the 120x68 level of a 1080p pyramid
(needs a build with VM_DEFS=-DVM_PROF: wave 0 of every workgroup stamps its phases)"""
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
el = nl - 1
capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
ctx.set_tuning(capi.SWEEP_PASS, 0, 0)
pr = capi.Progress()
capi.check(L.vm_optimize_level(pyr._h, el - 1, 40.0, None, 1, C.byref(pr)))
print("ms/iter %.3f launches %d cand/iter %.0f  us per pass launch %.2f" % (pr.elapsed_ms / pr.iters, pr.launches, pr.candidates / pr.iters, pr.elapsed_ms * 1e3 / pr.iters / 4))
buf = np.zeros(512 * 16 * 2, np.uint64)
L.vm_dbg_prof_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.vm_dbg_prof_read(buf.ctypes.data, buf.nbytes) == 0
st = buf[:8192].reshape(256, 4, 8).astype(np.int64)
se = buf[8192:8192 + 512].reshape(256, 2).astype(np.int64)
ok = se[:, 0] > 0
t0 = se[ok, 0].min()
print("workgroups:", ok.sum(), " launch span first start -> last end: %.2f us; start skew %.2f us" % ((se[ok, 1].max() - t0) / 100.0, (se[ok, 0].max() - t0) / 100.0))
names = ["phase start", "loads landed", "fold + owned stores", "mask test", "line search", "record stores + vmcnt + wg barrier", "tile barrier", "wg barrier 2"]
sf = buf[8192 + 512:8192 + 512 + 2048].reshape(256, 4, 2).astype(np.int64)
def pct(d):
    return "mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.max())
for ph in range(4):
    a = st[ok, ph]
    own = (np.where((a[:, 4] > a[:, 3]) & (a[:, 4] < a[:, 5]), a[:, 4], a[:, 3]) - a[:, 0]) / 100.0
    print("phase %d: wave 0 own work (start -> line search done): %s" % (ph, pct(own)))
    grp = np.arange(256)[ok] % 8
    arr = a[:, 5]
    last = np.array([arr[grp == g].max() for g in range(8)])
    print("         workgroup arrival (after its slowest wave) -> group's last arrival: %s" % pct((last[grp] - arr) / 100.0))
    print("         last arrival -> released (poll saw it): %s" % pct((a[:, 6] - last[grp]) / 100.0))
    print("phase %d: start at %.2f us (mean over workgroups), end %.2f" % (ph, (a[:, 0].mean() - t0) / 100.0, (a[:, 7].mean() - t0) / 100.0))
    # a stamp older than the phase's start is left over from an earlier launch (wave 0 had no
    # candidate: no line-search stamp): such a stage took no time in this workgroup
    a = a.copy()
    fresh = a >= a[:, :1]
    for k in range(1, 8):
        a[:, k] = np.where(fresh[:, k], a[:, k], a[:, k - 1])
    for k in range(1, 8):
        v = fresh[:, k]
        if v.sum() == 0:
            continue
        d = (a[v, k] - a[v, k - 1]) / 100.0
        print("   %-20s n=%3d mean %6.2f min %6.2f max %6.2f" % (names[k], v.sum(), d.mean(), d.min(), d.max()))
print("tail after the last barrier: %.2f us" % ((se[ok, 1] - st[ok, 3, 7]).mean() / 100.0))
print("entry -> phase 0 start: %.2f us" % ((st[ok, 0, 0] - se[ok, 0]).mean() / 100.0))
sx = buf[8192 + 512:8192 + 512 + 2048].reshape(256, 8).astype(np.int64)
for k, nm in enumerate(["view struct loaded", "tables requested, geometry", "early-out test done (1 workgroup barrier)"]):
    print("   entry -> %-45s %.2f us (mean), max %.2f" % (nm, ((sx[ok, k] - se[ok, 0]) / 100.0).mean(), ((sx[ok, k] - se[ok, 0]) / 100.0).max()))
