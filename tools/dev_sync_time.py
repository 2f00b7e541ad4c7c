"""dev: which part of the sync oracle is slow on the GPU box?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
w, h, d = 70, 37, 11
cons = [(10, 10, 1, 14, 12, 2), (40, 20, 3, 38, 22, 3)]
def T(msg, f, n=3):
    t = time.perf_counter()
    for _ in range(n): r = f()
    print("%-20s %.4f s per call" % (msg, (time.perf_counter() - t) / n), flush=True); return r
ui, bx, by, bz = T("sync_ui", lambda: oracle.sync_ui(w, h, d, 140, 74, cons, 100.0))
T("sync_diag", lambda: oracle.sync_diag(ui, 0.001))
p = np.random.default_rng(0).standard_normal((d, h, w)).astype(np.float32)
T("sync_apply (literal)", lambda: oracle.sync_apply(ui, 0.001, p))
T("sync_dot", lambda: oracle.sync_dot(p, p))
for it in (0, 1, 5, 25):
    x, y, z = (np.zeros((d, h, w), np.float32) for _ in range(3))
    T("solve_level iters=%d" % it, lambda: oracle.sync_solve_level(x, y, z, 140, 74, cons, 100.0, 0.001, float(it)), 1)
