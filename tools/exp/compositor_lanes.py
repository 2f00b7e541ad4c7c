"""round-5 experiment: the compositor (canvas upload from page-locked memory, v upscale, Poisson extension of both sides
in 4-frame batches, 9 renders per frame) of 48 1080p frames on L contexts (streams, one host thread each) side by side:
does a second lane hide the first one's PCIe uploads and latency-bound coarse-grid launches?

usage: compositor_lanes.py [frames per batch = 4] [lanes, comma separated = 1,2,3,4]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth

w, h, ex, nframes = 1920, 1080, 192, 48
per_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx0 = morph.Context(0, capi.MATH_FAST)
i0, i1 = synth.make_pair(w, h)
prm = morph.Parameters(); prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
pyr = morph.Pyramid(ctx0); pyr.build(i0, i1, 32)
morph.Morph(prm, pyr).calculate_halfway_parametrization()
rgb0, rgb1 = synth.make_rgb_pair(w, h)
e0, e1 = morph.pin_host(morph.make_extended(rgb0, ex)), morph.pin_host(morph.make_extended(rgb1, ex))

def lane_work(ctx, frs, nbatches):
    for _ in range(nbatches):
        for f in frs:
            f.upload(e0, e1, None, None)
            f.set_v_from_level(pyr, 1)
        morph.poisson_extend_frames(frs, tol=1e-5)
        for f in frs:
            for k in range(1, 10):
                f.render_halfway_dev(0.1 * k, 0.1 * k, 1)
    ctx.sync()

for L in ((1, 2, 3, 4) if len(sys.argv) < 3 else [int(a) for a in sys.argv[2].split(',')]):
    ctxs = [morph.Context(0, capi.MATH_FAST) for _ in range(L)]
    frs = [[morph.Frame(c, w, h, ex) for _ in range(per_batch)] for c in ctxs]
    for c, fr in zip(ctxs, frs):
        lane_work(c, fr, 1)                 # workspaces
    nb = nframes // per_batch // L
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=L) as ex_:
        list(ex_.map(lambda a: lane_work(a[0], a[1], nb), zip(ctxs, frs)))
    dt = time.perf_counter() - t0
    print("frames per batch %d, lanes %d: %.2f ms per frame (%d frames)" % (per_batch, L, dt * 1e3 / (nb * per_batch * L), nb * per_batch * L), flush=True)
    for fr in frs:
        for f in fr:
            f.close()
    for c in ctxs:
        c.close()
