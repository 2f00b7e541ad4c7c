"""dev: do solves on several contexts (streams, host threads) of one process overlap?  N threads, one context and one 1080p
pair each, every thread solves its pair `reps` times; wall time against N x the single-stream time.
usage: tools/dev_streams.py [sched] [N ...]     sched: auto | tile | step"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth  # noqa: E402

sched = {"auto": capi.SWEEP_AUTO, "tile": capi.SWEEP_TILE, "step": capi.SWEEP_STEP}[sys.argv[1] if len(sys.argv) > 1 else "auto"]
ns = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4, 8]
w, h, nlev, reps = 1920, 1080, 6, 3
prm = morph.Parameters()
kp = morph.KernParameters(prm)
L = capi.load()
i0, i1 = synth.make_pair(w, h, frame=0)
base = None
for n in ns:
    ctxs, pyrs = [], []
    for k in range(n):
        c = morph.Context(0, capi.MATH_FAST)
        c.set_params(kp)
        c.set_tuning(sched, 0, 0)
        p = morph.Pyramid(c)
        p.build(i0, i1, 32, nlevels=nlev)
        ctxs.append(c)
        pyrs.append(p)

    def work(k, r):
        for _ in range(r):
            prog = (capi.Progress * (nlev - 1))()
            capi.check(L.vm_solve(pyrs[k]._h, 500.0, 1.0, None, 0, None, 1, prog))
        ctxs[k].sync()
    for k in range(n):
        work(k, 1)                        # warm-up
    ths = [threading.Thread(target=work, args=(k, reps)) for k in range(n)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = (time.perf_counter() - t0) * 1e3 / reps
    if base is None:
        base = dt / n
    print("%d stream(s): %.1f ms per round of %d solves = %.2f x one solve alone" % (n, dt, n, dt / base), flush=True)
    for p in pyrs:
        p.clear()
    for c in ctxs:
        c.close()
