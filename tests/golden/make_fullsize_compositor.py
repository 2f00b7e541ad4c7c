"""Generates the full-size (1920x1080, ex = 192) fixtures of everything BASELINE config[4] adds to config[1]'s
solver: the ORACLE's outputs, computed once in the build container (CPU only), for inputs both hosts can regenerate
(tests/fullsize_fixture.py).

  render   -> render_1080p_hashes.json     kernel_render_halfway_image (render.cu:16-60) on 2304x1464 canvases:
              SHA-256 of the oracle's RGB8 frame for t in {0, 0.3, 0.5, 1} x color_from in {0, 1, 2} x
              {no quadratic path, path}, plus one frame with color_fa != geo_fa; byte-exact on the HIP side
  poisson  -> poisson_1080p_ring_f<k>.npz  CPoissonExt::prepare + poissonExtend (PoissonExt.cpp:49-362), both sides of
              frames 0 and 7: the oracle's double-precision CG at 1e-9 on the 2304x1464 canvas (1.30 M unknowns per
              side); the whole ring's RGB, delta-coded along rows and deflated.  The HIP solver is held to
              max |colour difference| <= 1 against it at every tolerance bench.py times
  qpath    -> qpath_1080p_lattice.npz      CQuadraticPath::optimize (QuadraticPath.cpp:24-223): the oracle's CG at
              1e-10 on the fixture field; u on a stride-6 lattice + four full rows and columns (float32)

usage: python tests/golden/make_fullsize_compositor.py [render] [poisson] [qpath] [--frames 0,7]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import synth  # noqa: E402  (host-side input generator only)
import oracle  # noqa: E402
import fullsize_fixture as FX  # noqa: E402

W, H, EX = 1920, 1080, 192
RENDER_CASES = [(t, t, cf, p) for t in (0.0, 0.3, 0.5, 1.0) for cf in (0, 1, 2) for p in (0, 1)] + [(0.7, 0.25, 1, 1)]
LATTICE = 6
LINES = (0, 377, 702, -1)


def make_extended(rgb, ex):
    """Pyramid::build's canvas, pyramid.cu:186-200: (255, 255, 255, 255) with the frame pasted at (ex, ex), alpha 0"""
    h, w = rgb.shape[:2]
    out = np.full((h + 2 * ex, w + 2 * ex, 4), 255, np.uint8)
    out[ex:ex + h, ex:ex + w, :3] = rgb
    out[ex:ex + h, ex:ex + w, 3] = 0
    return out


def render(frame=0):
    rgb0, rgb1 = synth.make_rgb_pair(W, H, frame=frame)
    v, u = FX.field(W, H, frame), FX.path(W, H, frame)
    e0, e1 = FX.padded(rgb0, EX), FX.padded(rgb1, EX)
    f0, f1 = e0.astype(np.float32), e1.astype(np.float32)
    doc = {"size": [W, H], "ex": EX, "frame": frame, "inputs": FX.sha(e0, e1, v, u), "cases": []}
    zero = np.zeros_like(u)
    for color_fa, geo_fa, cf, p in RENDER_CASES:
        img = oracle.render_halfway(W, H, EX, color_fa, geo_fa, cf, f0, f1, v, u if p else zero)
        doc["cases"].append({"color_fa": color_fa, "geo_fa": geo_fa, "color_from": cf, "path": p, "sha256": FX.sha(img),
                             "mean": round(float(img.mean()), 4)})
        print("render", doc["cases"][-1], flush=True)
    json.dump(doc, open(os.path.join(HERE, "render_1080p_hashes.json"), "w"), indent=1, sort_keys=True)


def poisson(frames):
    for frame in frames:
        rgb0, rgb1 = synth.make_rgb_pair(W, H, frame=frame)
        v = FX.field(W, H, frame)
        e = [make_extended(rgb0, EX), make_extended(rgb1, EX)]
        out = {"inputs": np.frombuffer(FX.sha(e[0], e[1], v).encode(), np.uint8), "tol": np.float64(1e-9)}
        for side in (1, 2):
            t0 = time.time()
            other = e[2 - side][EX:EX + H, EX:EX + W].copy()           # the crop taken before any extension, PoissonExt.cpp:26-27
            ref, it, rr = oracle.poisson_extend(e[side - 1], W, H, EX, other, v, side, tol=1e-9, max_it=200000)
            assert rr <= 1e-9 and ref[..., 3].max() == 0
            for name, band in zip(("top", "bottom", "left", "right"), FX.ring_bands(ref, W, H, EX)):
                out["s%d_%s" % (side, name)] = FX.delta_encode(band, 1)
            out["s%d_iters" % side] = np.int64(it)
            out["s%d_rel_res" % side] = np.float64(rr)
            out["s%d_sha256" % side] = np.frombuffer(FX.sha(ref).encode(), np.uint8)
            print("poisson frame %d side %d: %d CG iterations (3 channels), rel. residual %.2e, %.0f s" % (frame, side, it, rr, time.time() - t0), flush=True)
        np.savez_compressed(os.path.join(HERE, "poisson_1080p_ring_f%d.npz" % frame), **out)


def qpath(frame=0):
    v = FX.field(W, H, frame)
    t0 = time.time()
    u, it, rr = oracle.quadratic_path(v, tol=1e-10)
    print("qpath: %d CG iterations, rel. residual %.2e, %.0f s, max |u| %.3f" % (it, rr, time.time() - t0, np.abs(u).max()), flush=True)
    np.savez_compressed(os.path.join(HERE, "qpath_1080p_lattice.npz"), inputs=np.frombuffer(FX.sha(v).encode(), np.uint8),
                        lattice=u[::LATTICE, ::LATTICE].copy(), rows=u[list(LINES)].copy(), cols=u[:, list(LINES)].copy(),
                        stride=np.int64(LATTICE), lines=np.asarray(LINES, np.int64), iters=np.int64(it), rel_res=np.float64(rr),
                        abs_max=np.float64(np.abs(u).max()), abs_mean=np.float64(np.abs(u).mean()), frame=np.int64(frame))


if __name__ == "__main__":
    args = sys.argv[1:]
    frames = [int(x) for x in args[args.index("--frames") + 1].split(",")] if "--frames" in args else [0, 7]
    what = [a for a in args if a in ("render", "poisson", "qpath")] or ["render", "poisson", "qpath"]
    oracle.lib().vmo_set_threads(len(os.sched_getaffinity(0)))
    if "render" in what:
        render()
    if "qpath" in what:
        qpath()
    if "poisson" in what:
        poisson(frames)
