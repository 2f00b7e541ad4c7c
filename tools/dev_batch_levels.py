"""dev: per-level time of a batched fixed-work solve (8 / 32 pairs) under each schedule policy"""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = morph.Context(0, capi.MATH_EXACT if os.environ.get("VM_DEV_EXACT") else capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 0
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx.set_tuning(sched, threads, parts)
w, h = 1920, 1080
pyrs = []
for k in range(nb):
    i0, i1 = synth.make_pair(w, h, frame=k % 4)
    p = morph.Pyramid(ctx); p.build(i0, i1, 32); pyrs.append(p)
L = pyrs[0]._L
nl = pyrs[0].size() - 1
arr = (C.c_void_p * nb)(*[p._h for p in pyrs])
for rep in range(2):
    prog = (capi.Progress * (nb * (nl - 1)))()
    ctx.sync(); t = time.perf_counter()
    capi.check(L.vm_solve_batch(arr, nb, 500.0, 1.0, None, 1, prog))
    ctx.sync(); dt = time.perf_counter() - t
print("batch %d sched %d parts %d: %.1f ms per batch, %.1f ms per pair" % (nb, sched, parts, dt * 1e3, dt * 1e3 / nb))
for el in range(nl - 2, -1, -1):
    pr = prog[el]
    print("  level %4dx%-4d  %8.1f ms  launches %6d  cand/iter/pair %8.0f" % (pyrs[0][el + 1].width, pyrs[0][el + 1].height, pr.elapsed_ms, pr.launches, pr.candidates / pr.iters))
