"""Known-answer tests of the oracle's temporal coherence path (oracle/vm_oracle_temporal.c,
the flag == true energy term, the flow pyramid).  upsample.cu / pyramid.cu's temporal halves
have no reference-run outputs (CUDA + OpenCV), so these closed forms are what pins them; the
flow SCALING is pinned by outputs of the reference's own resample library
(tests/golden/flow_ref.npz, generator tests/golden/make_flow_golden.py).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from videomorphing_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _splat(O, v, f0, f1, ssim=None):
    h, w = v.shape[:2]
    acc = np.zeros((h, w, 3), dtype=np.int64)
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (v, f0, f1)]
    s = np.ascontiguousarray(ssim, dtype=np.float32) if ssim is not None else None
    O.lib().vmo_temp_splat(w, h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                           s.ctypes.data if s is not None else None, acc.ctypes.data)
    return acc


def _norm(O, acc):
    h, w = acc.shape[:2]
    v, wt = np.zeros((h, w, 2), np.float32), np.zeros((h, w), np.float32)
    O.lib().vmo_temp_normalise(w, h, acc.ctypes.data, v.ctypes.data, wt.ctypes.data)
    return v, wt


def test_splat_with_zero_flow_reproduces_the_field(oracle):
    rng = np.random.RandomState(1)
    h, w = 12, 17
    v = rng.randn(h, w, 2).astype(np.float32) * 3
    z = np.zeros((h, w, 2), np.float32)
    ref, wt = _norm(oracle, _splat(oracle, v, z, z))
    # p_ref = p: weight 1 on itself, 0 on the others; values to the fixed-point quantum 2^-32
    assert np.abs(ref - v).max() <= 2.0 ** -32 and np.all(wt == 1.0)


def test_splat_translates_by_an_integer_flow_and_leaves_holes(oracle):
    rng = np.random.RandomState(2)
    h, w = 10, 14
    v = rng.randn(h, w, 2).astype(np.float32)
    f = np.broadcast_to(np.float32([2, 1]), (h, w, 2)).copy()
    ref, wt = _norm(oracle, _splat(oracle, v, f, f))
    # p_ref = p + 0.5 (f0 + f1) = p + (2, 1); v_ref = v + 0.5 (f1 - f0) = v
    # (the bilinear taps of the constant flow sum their four weights to 1 +- 1 ulp)
    assert np.abs(ref[1:, 2:] - v[:-1, :-2]).max() <= 5e-7 and np.abs(wt[1:, 2:] - 1.0).max() <= 5e-7
    assert np.all(wt[0, :] == 0) and np.all(wt[:, :2] == 0) and np.all(ref[0] == 0)


def test_splat_half_pixel_flow_and_different_flows_per_video(oracle):
    h, w = 8, 12
    v = np.broadcast_to(np.float32([1.5, -0.5]), (h, w, 2)).copy()
    f0 = np.broadcast_to(np.float32([0.5, 0.0]), (h, w, 2)).copy()
    f1 = np.broadcast_to(np.float32([1.5, 0.0]), (h, w, 2)).copy()
    ref, wt = _norm(oracle, _splat(oracle, v, f0, f1))
    # p_ref = p + (1, 0) exactly; v_ref = v + 0.5 (f1 - f0) = v + (0.5, 0)
    assert np.allclose(ref[:, 1:], [2.0, -0.5]) and np.all(wt[:, 1:] == 1.0) and np.all(wt[:, 0] == 0)
    f0[...] = (0.25, 0.0)
    f1[...] = (0.75, 0.0)                                     # p_ref = p + 0.5: two targets, weight 1/2 each
    ref, wt = _norm(oracle, _splat(oracle, v, f0, f1))
    assert np.allclose(wt[:, 1:-1], 1.0) and np.allclose(wt[:, 0], 0.5)
    assert np.allclose(ref[:, 1:], [1.75, -0.5])


def test_splat_is_weighted_by_the_neighbour_pages_ssim_and_order_independent(oracle):
    rng = np.random.RandomState(3)
    h, w = 9, 11
    va, vb = rng.randn(h, w, 2).astype(np.float32), rng.randn(h, w, 2).astype(np.float32)
    fa = (rng.rand(h, w, 2).astype(np.float32) - 0.5) * 3
    fb = (rng.rand(h, w, 2).astype(np.float32) - 0.5) * 3
    s = rng.rand(h, w).astype(np.float32)
    z = np.zeros((h, w, 2), np.float32)
    ref, wt = _norm(oracle, _splat(oracle, va, z, z, s))
    assert np.abs(wt - s).max() <= 2.0 ** -32 and np.allclose(ref, va, atol=1e-6)      # sum v s / sum s
    # two sources into one accumulator: the order of accumulation does not matter (fixed point)
    acc1 = _splat(oracle, va, fa, fb) + _splat(oracle, vb, fb, fa)
    acc2 = _splat(oracle, vb, fb, fa) + _splat(oracle, va, fa, fb)
    assert np.array_equal(acc1, acc2)
    # and agrees with a float64 accumulation of the same contributions to float rounding
    ref, wt = _norm(oracle, acc1)
    assert np.isfinite(ref).all() and wt.max() < 16


def test_temporal_fill_smooths_and_fills_rows_as_written(oracle):
    """upsample.cu:303-337 for one in-between page.  fill_zeros_x divides the UNWEIGHTED sum of
    the nearest valid neighbours by the sum of inverse distances (:129-148): a hole at distance
    k from a single valid neighbour gets k times its value -- replicated literally."""
    h, w = 6, 16
    vv = np.broadcast_to(np.float32([1.0, 2.0]), (h, w, 2)).copy()
    z = np.zeros((h, w, 2), np.float32)
    out = np.zeros((h, w, 2), np.float32)
    arrs = [vv, z, z, vv, z, z]
    oracle.lib().vmo_temporal_fill(w, h, *[a.ctypes.data for a in arrs], out.ctypes.data)
    assert np.array_equal(out, vv)                          # both neighbours land on every pixel
    f = np.broadcast_to(np.float32([3.0, 0.0]), (h, w, 2)).copy()      # everything moves 3 px right
    oracle.lib().vmo_temporal_fill(w, h, vv.ctypes.data, f.ctypes.data, f.ctypes.data,
                                   vv.ctypes.data, f.ctypes.data, f.ctypes.data, out.ctypes.data)
    assert np.allclose(out[:, 3:], [1.0, 2.0])
    for k, x in ((1, 2), (2, 1), (3, 0)):                  # holes x < 3: one valid neighbour at distance k
        assert np.allclose(out[:, x], [1.0 * k, 2.0 * k]), (x, out[0, x])


def test_temporal_energy_term(oracle):
    """energy_change with flag == true adds w_temp (|v + d - ref|_1 - |v - ref|_1) mask factor_d
    inv_wh (morph.cu:752-759); flag == false adds +0"""
    w, h = 24, 20
    i0, i1 = synth.make_pair(w, h)
    lv = oracle.Level(w, h)
    lv.set_images(i0, i1)
    rng = np.random.RandomState(5)
    lv.field("v")[...] = rng.randn(h, w, 2).astype(np.float32) * 0.3
    lv.init(0.0)
    P = oracle.default_params()
    lv.field("temp_ref")[...] = rng.randn(h, w, 2).astype(np.float32)
    lv.field("temp_mask")[...] = rng.rand(h, w).astype(np.float32)
    px, py, dx, dy = 9, 7, 0.37, -0.21
    e0 = oracle.lib().vmo_dbg_energy_change(lv._p, C.byref(P), px, py, dx, dy)
    lv.set_temporal(1, 4.0)
    e1 = oracle.lib().vmo_dbg_energy_change(lv._p, C.byref(P), px, py, dx, dy)
    v, r, m = lv.field("v")[py, px], lv.field("temp_ref")[py, px], lv.field("temp_mask")[py, px]
    vt = (abs(v[0] + dx - r[0]) - abs(v[0] - r[0])) + (abs(v[1] + dy - r[1]) - abs(v[1] - r[1]))
    want = P.w_temp * vt * m * 4.0 / (w * h)
    assert abs((e1 - e0) - want) < 1e-5 * max(1.0, abs(want)) and abs(want) > 1e-4
    lv.set_temporal(0, 4.0)
    assert oracle.lib().vmo_dbg_energy_change(lv._p, C.byref(P), px, py, dx, dy) == e0


def test_flow_scale_matches_the_reference_library(oracle):
    """pinned by reference-run outputs: tests/golden/flow_ref.npz comes from include/resample
    itself (image::load(-50, 50) -> scale -> image::store, pyramid.cu:284-321 call order)"""
    G = np.load(os.path.join(HERE, "golden", "flow_ref.npz"))
    for name in sorted(k[:-3] for k in G.files if k.endswith("_in")):
        fin, want = G[name + "_in"], G[name + "_out"]
        got = oracle.flow_scale(fin, want.shape[1], want.shape[0])
        d = np.abs(got - want)
        assert d.max() <= 5e-5, (name, d.max())            # flows span [-50, 50]: 1 ulp of powf x 100
        assert (d == 0).mean() > 0.3, (name, (d == 0).mean())


def test_flow_scale_of_a_constant_flow(oracle):
    f = np.broadcast_to(np.float32([4.0, -2.0]), (24, 32, 2)).copy()
    same = oracle.flow_scale(f, 32, 24)
    assert np.abs(same - f).max() < 2e-3                    # same size: identity up to the sRGB round trip
    half = oracle.flow_scale(f, 16, 12)
    assert np.abs(half - [2.0, -1.0]).max() < 2e-3          # half the pixels, half the displacement


def test_flow_concat_of_constant_flows_adds_them(oracle):
    a = np.broadcast_to(np.float32([1.5, 0.5]), (10, 12, 2)).copy()
    b = np.broadcast_to(np.float32([0.25, -1.0]), (10, 12, 2)).copy()
    assert np.allclose(oracle.flow_concat(a, b), [1.75, -0.5])
    y, x = np.mgrid[0:10, 0:12].astype(np.float32)
    ramp = np.stack([0.1 * x, 0.0 * y], -1).astype(np.float32)          # f_next(q) = (0.1 q.x, 0)
    got = oracle.flow_concat(a, ramp)                                    # f(p) + f_next(p + f(p))
    want = 1.5 + 0.1 * np.clip(x + 1.5, 0, 11)
    assert np.allclose(got[..., 0], want, atol=1e-5)


def test_video_geometry_integer_form_equals_the_float_form():
    import itertools
    import oracle as O
    for (w, h, d, sr) in itertools.product([64, 80, 127, 128, 256, 1920], [48, 64, 200, 1080], [1, 2, 5, 8, 16, 17, 33, 64], [4, 8, 32]):
        if min(w, h) < sr:
            continue
        assert O.video_geometry(w, h, d, sr) == synth.video_levels(w, h, d, sr), (w, h, d, sr)
    lv, ft = synth.video_levels(80, 64, 16, 8)
    assert lv == [(80, 64, 16), (40, 32, 16), (20, 16, 16), (10, 8, 9)] and ft == [1, 1, 1, 2]
    assert synth.page_frames(lv, ft)[3] == [0, 2, 4, 6, 8, 10, 12, 14, 15]
    assert O.factor_d_table(16, [l[2] for l in lv]) == [2.0, 2.0, 2.0, 2.0, 1.0]


def test_video_solve_ties_pages_to_their_neighbours(oracle):
    """a whole small video solve on the oracle: the frames of a static video (zero flow) carry
    independent noise, so pages solved on their own (w_temp = 0) disagree; the temporal term
    pulls every page towards its solved neighbour"""
    w, h, d = 40, 32, 3
    i0, i1 = synth.make_pair(w, h, amp=0.8)
    levels = [(w, h, d), (20, 16, d)]
    z = np.zeros((h, w, 2), np.float32)
    spread = {}
    for wt in (0.0, 40.0):
        vid = oracle.Video(levels)
        for t in range(d):
            rng = np.random.RandomState(100 + t)
            vid.set_images(0, t, i0 + 4 * rng.randn(h, w).astype(np.float32), i1 + 4 * rng.randn(h, w).astype(np.float32))
        vid.set_flows(0, {k: [z] * d for k in ("f0", "f1", "b0", "b1")})
        P = oracle.default_params(w_temp=wt)
        vid.solve(P, 15)
        vs = [vid.pages[0][t].field("v").copy() for t in range(d)]
        assert np.abs(vs[1]).max() > 0.05
        spread[wt] = 0.5 * (np.abs(vs[0] - vs[1]).mean() + np.abs(vs[2] - vs[1]).mean())
        if wt > 0:
            assert vid.pages[0][0].field("temp_mask").min() > 0        # every pixel got a reference
            assert vid.iters[0].keys() == {0, 1, 2}
    assert spread[0.0] > 0 and spread[40.0] < 0.8 * spread[0.0], spread
