"""dev: does the 60-pair job slow down with the allocation history of the process?  (bench.py's scale_reference took 970 ms
after the other extras and 838 before them.)  One process: the job on fresh contexts (A); the same contexts and pyramids again
(A'); churn -- 48 pyramids built, solved in batches of 8 on a fourth context, freed, twice; the job on the OLD contexts and
pyramids (A''); the job on NEW contexts and pyramids (B).  ms per job.  usage: tools/dev_alloc_history.py"""
import ctypes as C
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from videomorphing_amd import capi, morph, synth  # noqa: E402

w, h, nlev = 1920, 1080, 6
prm = morph.Parameters()
kp = morph.KernParameters(prm)
L = capi.load()
imgs = [synth.make_pair(w, h, frame=k) for k in range(8)]


def solve_group(ps, fixed=1):
    arr = (C.c_void_p * len(ps))(*[p._h for p in ps])
    prog = (capi.Progress * (len(ps) * (nlev - 1)))()
    capi.check(L.vm_solve_batch(arr, len(ps), 500.0, 1.0, None, fixed, prog))
    return [prog]


def make_job():
    ctxs = [morph.Context(0, capi.MATH_FAST) for _ in range(3)]
    for c in ctxs:
        c.set_params(kp)

    def pyramid(c, im):
        p = morph.Pyramid(c)
        p.build(im[0], im[1], 32, nlevels=nlev)
        return p
    pyrs, B, nctx, _ = bench.config2_setup(list(range(60)), ctxs, lambda ids: imgs, pyramid, 32)
    return ctxs, pyrs, B


def run(job, reps=2):
    ctxs, pyrs, B = job
    out = []
    for _ in range(reps):
        for c in ctxs:
            c.sync()
        t0 = time.perf_counter()
        bench.config2_step(pyrs, ctxs, B, solve_group)
        for c in ctxs:
            c.sync()
        out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out


A = make_job()
print("A   fresh contexts, warm-up + 2:", run(A, 3), flush=True)
cx = morph.Context(0, capi.MATH_FAST)
cx.set_params(kp)
for rnd in range(2):
    churn = []
    for k in range(48):
        p = morph.Pyramid(cx)
        p.build(imgs[k % 8][0], imgs[k % 8][1], 32, nlevels=nlev)
        churn.append(p)
    for g in range(0, 48, 8):
        solve_group(churn[g:g + 8])
    cx.sync()
    for p in churn:
        p.clear()
    del churn
print("A'' old contexts and pyramids after the churn:", run(A, 2), flush=True)
Bj = make_job()
print("B   new contexts and pyramids after the churn, warm-up + 2:", run(Bj, 3), flush=True)
print("A''' the old ones once more:", run(A, 1), flush=True)
