"""dev: does the config[4] compositor's speed depend on WHICH hardware queue its second lane's stream lands on?  k dummy contexts
(streams) are created and kept before the job is set up; the compositor of 30 frames is timed three times.
usage: tools/exp/queue_probe.py [kmax]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from videomorphing_amd import capi, morph, synth
w, h, ex = 1920, 1080, 192
blk = capi.ParamBlock()
blk.kp = morph.KernParameters(morph.Parameters())
blk.max_iter, blk.max_iter_drop_factor, blk.start_res, blk.math_mode = 500.0, 1.0, 32, capi.MATH_FAST
cons = synth.make_constraints(w, h, 8)
ids = list(range(8))
imgs = [synth.make_pair(w, h, frame=f) for f in ids]
rgb0, rgb1 = synth.make_rgb_pair(w, h)
nlev = synth.num_levels(w, h, 32)
main = morph.Context(0, capi.MATH_FAST)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    dummies = [morph.Context(0, capi.MATH_FAST) for _ in range(k)]
    job = bench.Config4Job(np, capi, morph, synth, 0, blk, cons, w, h, nlev, ids, imgs, rgb0, rgb1, ex, 1e-5, lane0=main)
    try:
        g = job.pyramids()
        job.warm_up(g)
        r = [job.compositor(g) for _ in range(3)]
        best = min(r, key=lambda t: t[0])
        print("dummy streams %2d: lane gain %.2f after %d rejected streams, do-nothing probe says overlap %s: compositor ms per frame %.2f (each run %s), split %s" % (k, job.lane_gain or 0, job.lane_retries, job.lane_ctx[0].runs_beside(job.lane_ctx[1]), best[0] * 1e3 / 8, [round(t[0] * 1e3 / 8, 2) for t in r], [round(x * 1e3 / 8, 2) for x in best[1]]), flush=True)
    finally:
        job.close()
        for d in dummies:
            d.close()
