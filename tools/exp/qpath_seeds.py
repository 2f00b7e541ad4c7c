import os, sys
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth
import oracle
ctx = morph.Context(0, capi.MATH_FAST)
for seed, w, h in ((8368, 192, 6), (8132, 322, 32), (7212, 192, 7), (7116, 192, 8)):
    # the fuzz tool's recipe for the quadratic path's field (tools/dev_fuzz_poisson.py), rng state replayed
    rng = np.random.RandomState(seed)
    int(rng.choice([rng.randint(8, 40), rng.randint(40, 140), rng.randint(120, 330), 64, 128, 192, 63, 65, 127, 129])); int(rng.choice([rng.randint(6, 30), rng.randint(30, 100), rng.randint(90, 200), 16, 32, 48, 15, 17, 31, 33]))
    ex = int(rng.choice([1, 2, 3, rng.randint(2, 12), rng.randint(8, 40), 16, 64]))
    v = (rng.uniform(0.2, 1.2) * synth.displacement(w, h) + rng.uniform(0, 0.4) * rng.randn(h, w, 2)).astype(np.float32)
    for e in range(2):
        for _ in range(rng.randint(0, 7)):
            rw, rh = rng.randint(1, max(2, w // 4)), rng.randint(1, max(2, h // 4))
            rng.randint(0, w - rw + 1), rng.randint(0, h - rh + 1)
        for _ in range(rng.randint(0, 12)):
            rng.randint(0, h), rng.randint(0, w)
    vs = (0.5 * synth.displacement(w, h) + 0.05 * rng.randn(h, w, 2)).astype(np.float32)
    uo, _, _ = oracle.quadratic_path(vs, tol=1e-10)
    fr = morph.Frame(ctx, w, h, 2)
    fr.upload(None, None, vs, None)
    for tol in (1e-4, 3e-5, 1e-5):
        try:
            r = fr.quadratic_path(tol=tol)
            print(seed, w, h, "tol", tol, "iterations", r[0], "residual %.2e" % r[1], "max |u - oracle| %.2e" % float(np.abs(fr.download_qpath() - uo).max()), flush=True)
        except capi.VmError as e:
            print(seed, w, h, "tol", tol, "FAILED", str(e)[-80:], flush=True)
    fr.close()
