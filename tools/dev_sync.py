"""dev: time the synchronisation stage on a 1080p x 60-frame pair (per level: iterations, ms, us per
iteration, achieved algorithmic GB/s at 124 B per voxel-iteration) and the stage-1 renderer"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph
max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 500
d = int(sys.argv[2]) if len(sys.argv) > 2 else 60
render = int(sys.argv[3]) if len(sys.argv) > 3 else 1
only = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # > 0: solve just this level, from zero (profiling)
w0, h0 = 1920, 1080
ctx = morph.Context(0, capi.MATH_FAST)
P = morph.Parameters()
P.w_ui, P.w_tps, P.max_iter = 100.0, 0.001, max_iter
rng = np.random.default_rng(23)
for k in range(24):
    lx, ly, lz = int(rng.integers(2, w0 - 2)), int(rng.integers(2, h0 - 2)), int(rng.integers(0, d))
    P.lp.append([morph.Conp(lx, ly, lz)])
    P.rp.append([morph.Conp(int(np.clip(lx + rng.integers(-40, 41), 0, w0 - 1)), int(np.clip(ly + rng.integers(-40, 41), 0, h0 - 1)),
                            int(np.clip(lz + rng.integers(-3, 4), 0, d - 1)))])
    P.cnt.append([morph.Connect((k, 0), (k, 0))])
levels = morph.sync_level_table(w0, h0, d, 16)
print("levels", levels)
pyr = morph.SyncPyramid(ctx)
pyr.build_levels(levels)
for rep in range(2):
    th = morph.SyncThread(P, pyr)
    ctx.sync(); t = time.perf_counter()
    if only:
        th._max_iter = float(max_iter)
        th.load_identity(only)
        th.optimize_level(only)
    else:
        th.run()
    ctx.sync(); dt = time.perf_counter() - t
print("sync solve: %.1f ms wall (incl. result delivery of %d frames)" % (dt * 1e3, d))
tot = 0
for el in sorted(th.progress, reverse=True):
    pr = th.progress[el]
    w, h, dd = levels[el]
    n = w * h * dd
    us = pr["elapsed_ms"] * 1e3 / max(pr["iters"], 1)
    tot += pr["elapsed_ms"]
    print("  level %d %4dx%-4dx%-3d %8d voxels  iters %5d  %8.2f ms  %7.2f us/iter  %7.1f GB/s algorithmic  resid %s" %
          (el, w, h, dd, n, pr["iters"], pr["elapsed_ms"], us, n * 124.0 / (us * 1e-6) / 1e9, pr["resid"]))
print("  levels total %.1f ms, %.3f G voxel-iters/s" % (tot, sum(p["voxel_iters"] for p in th.progress.values()) / tot / 1e6))
if render:
    dr = min(d, 8)
    frame = np.zeros((h0, w0, 4), np.uint8)
    frame[..., 0] = (np.arange(w0) % 256)[None, :]
    frame[..., 1] = (np.arange(h0) % 256)[:, None]
    flow = np.zeros((h0, w0, 2), np.float32)
    flow[..., 0] = 1.5
    for s in range(2):
        for t_ in range(d):
            pyr.upload_frame(s, t_, frame)
            pyr.upload_flow(s, t_, flow)
    for fa in (0.0, 1.0, 0.5):
        ms = [pyr.render_resample_dev(fa, f) for f in range(dr)]
        print("render_resample fa=%.1f: %.3f ms per frame (first %.3f)" % (fa, float(np.median(ms[1:])), ms[0]))
