"""dev: the dense first sweeps of the 1080p / 960x540 levels for a batch of pairs (ms per iteration)"""
import sys, os, ctypes as C, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx.set_tuning(capi.SWEEP_TILE, threads, 0)
frames = [synth.make_pair(w, h, frame=k) for k in range(min(npairs, 4))]
v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(1).randn(h, w, 2)).astype(np.float32)
batch = []
for k in range(npairs):
    pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
    pyr.upload_luma(1, *frames[k % len(frames)]); pyr[1].v = v0
    batch.append(pyr)
arr = (C.c_void_p * npairs)(*[p._h for p in batch])
for rep in range(2):
    for p in batch:
        capi.check(p._L.vm_init_level(p._h, 0, w, h, None, 0))
    prog = (capi.Progress * npairs)()
    capi.check(batch[0]._L.vm_optimize_level_batch(arr, npairs, 0, 2.0, None, 1, prog))
print("T=%d " % threads, end="")
print("pairs %d %dx%d: dense sweeps %.3f ms per iteration (%.1f us per pass launch), sched ms %s, evals %.3g, commits %d" % (
    npairs, w, h, prog[0].elapsed_ms / 2, prog[0].elapsed_ms * 1e3 / 8, [round(x, 2) for x in prog[0].sched_ms], sum(p.evaluations for p in prog), sum(p.commits for p in prog)))
import hashlib
print("v checksum", float(np.abs(batch[-1][1].v).sum()), "sha1", hashlib.sha1(np.ascontiguousarray(batch[-1][1].v).tobytes()).hexdigest()[:16], hashlib.sha1(np.ascontiguousarray(batch[0][1].v).tobytes()).hexdigest()[:16])
