"""Known-answer tests that pin the CPU oracle (CPU only, no GPU).

The reference holds no tests or golden vectors for this path;
what pins the oracle is listed in oracle/vm_oracle.h: the table rows recorded from
a run of the reference's stencils.cpp (tests/golden/stencil_rows.json), the
agreement of the reference's two statements of the thin-plate operator, and the
analytic KATs of SURVEY.md 8(c) below.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from videomorphing_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tables(oracle):
    L = oracle.lib()
    tps = np.zeros((5, 5, 5, 5), np.float32)
    tps2 = np.zeros((5, 5, 5, 5), np.float32)
    io = np.zeros((5, 5, 5, 5), np.int32)
    imp = np.zeros((5, 5, 3, 3), np.uint32)
    L.vmo_tps_stencil(tps.ctypes.data)
    L.vmo_tps_rows_from_dense(tps2.ctypes.data)
    L.vmo_io_stencil(io.ctypes.data)
    L.vmo_improvmask_stencil(imp.ctypes.data)
    return tps, tps2, io, imp


def test_stencil_rows_recorded_from_reference(tables):
    tps, _, io, imp = tables
    g = json.load(open(os.path.join(HERE, "golden", "stencil_rows.json")))
    assert tps[2, 2].ravel().tolist() == g["tps_interior_row"]
    corner = np.zeros((5, 5), np.float32)
    for k, v in g["tps_corner_00_nonzero"].items():
        i, j = [int(t) for t in k.strip("()").split(",")]
        corner[i, j] = v
    assert np.array_equal(tps[0, 0], corner)
    assert io[0, 0].ravel().tolist() == g["iomask_corner_00"]
    assert int(imp[0, 0, 0, 0]) == int(g["improvmask_00_block_00"], 16)


def test_tps_two_reference_statements_agree(tables):
    """stencils.cpp:156-261 vs the dense rows of morph.cu:439-469, all 25 border classes"""
    tps, tps2, _, _ = tables
    assert np.array_equal(tps, tps2)
    # symmetry of the Hessian: row p at q equals row q at p (checked on the 5x5 image)
    for y in range(5):
        for x in range(5):
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    qx, qy = x + dx, y + dy
                    if 0 <= qx < 5 and 0 <= qy < 5:
                        assert tps[y, x, dy + 2, dx + 2] == tps[qy, qx, 2 - dy, 2 - dx]
    # rows annihilate constants and (in the interior) affine fields
    assert np.allclose(tps.sum(axis=(2, 3)), 0)


def test_io_and_improvmask_tables(tables):
    _, _, io, imp = tables
    for By in range(5):
        for Bx in range(5):
            # a 9x9 image realises the classes at positions {0,1,4,7,8}
            pos = [0, 1, 4, 7, 8]
            y, x = pos[By], pos[Bx]
            for i in range(5):
                for j in range(5):
                    inside = 0 <= y + i - 2 < 9 and 0 <= x + j - 2 < 9
                    assert io[By, Bx, i, j] == int(inside)
    # every pixel's window is covered exactly once by the union of the block masks
    for oy in range(5):
        for ox in range(5):
            bits = sum(bin(int(m)).count("1") for m in imp[oy, ox].ravel())
            assert bits == 25
            assert imp[oy, ox, 1, 1] & (1 << (ox + 5 * oy))


def test_calc_border_matches_branchfree_original(oracle):
    """the #if 1 arithmetic of morph.cu:45-52 restated literally vs the oracle's branches"""
    def isignbit(i):
        return (i & 0xFFFFFFFF) >> 31

    def orig(p, dim):
        s = isignbit(p - 2)
        aux = p - (dim - 2)
        return p * s + (0 if s else 1) * (2 + (0 if isignbit(aux) else 1) * (1 + aux))
    L = oracle.lib()
    for dim in (5, 6, 7, 34, 1080):
        for p in range(dim):
            assert L.vmo_calc_border(p, dim) == orig(p, dim)


def test_ssim_closed_forms(oracle):
    L = oracle.lib()
    n = 25.0
    assert L.vmo_ssim(100.0, 100.0, 500.0, 500.0, 450.0, 1.0, 0.0) == 0.0      # n <= 1
    assert L.vmo_ssim(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0) == 0.0
    # equal constant windows: zero variance -> c = s = 1
    a = 120.0
    assert L.vmo_ssim(n * a, n * a, n * a * a, n * a * a, n * a * a, n, 0.0) == 1.0
    # identical textured windows: value 1 up to rounding
    rng = np.random.RandomState(0)
    x = rng.uniform(16, 240, 25).astype(np.float32)
    val = L.vmo_ssim(float(x.sum()), float(x.sum()), float((x * x).sum()), float((x * x).sum()),
                     float((x * x).sum()), n, 0.0)
    assert abs(val - 1.0) < 1e-5
    # against the formula in double
    y = rng.uniform(16, 240, 25).astype(np.float32)
    mx, my = x.mean(dtype=np.float64), y.mean(dtype=np.float64)
    vx = max(0.0, (x.astype(np.float64) ** 2).mean() - mx * mx)
    vy = max(0.0, (y.astype(np.float64) ** 2).mean() - my * my)
    cov = (x.astype(np.float64) * y).mean() - mx * my
    c2, c3 = 58.5225, 29.26125
    ref = min(1.0, (2 * np.sqrt(vx * vy) + c2) / (vx + vy + c2) * (abs(cov) + c3) / (np.sqrt(vx * vy) + c3))
    val = L.vmo_ssim(float(x.sum()), float(y.sum()), float((x * x).sum()), float((y * y).sum()),
                     float((x * y).sum()), n, 0.0)
    assert abs(val - ref) < 2e-4
    assert L.vmo_ssim(float(x.sum()), float(y.sum()), float((x * x).sum()), float((y * y).sum()),
                      float((x * y).sum()), n, 0.9) >= np.float32(0.9)           # clamp


def test_tex2d_semantics(oracle):
    L = oracle.lib()
    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    t = lambda x, y: L.vmo_tex2d(img.ctypes.data, 4, 3, x, y)
    assert t(0.5, 0.5) == 0.0 and t(3.5, 2.5) == 11.0           # texel centres
    assert t(1.0, 0.5) == 0.5 and t(0.5, 1.0) == 2.0             # midpoints
    assert t(-5.0, -5.0) == 0.0 and t(99.0, 99.0) == 11.0        # clamp addressing
    assert t(1.25, 1.75) == pytest.approx(0.75 + 4 * 1.25)


def _level(oracle, w, h, i0, i1, v=None, clamp=0.0):
    lv = oracle.Level(w, h)
    lv.set_images(i0, i1)
    if v is not None:
        lv.field("v")[...] = v
    lv.init(clamp)
    return lv


def test_init_level_affine_field_has_zero_tps_gradient(oracle):
    """KAT 4: the bending-energy gradient of an affine field vanishes in the interior"""
    w, h = 40, 30
    i0, i1 = synth.make_pair(w, h)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    v = np.stack([0.02 * x - 0.01 * y + 0.3, 0.015 * y + 0.01 * x - 0.2], -1).astype(np.float32)
    lv = _level(oracle, w, h, i0, i1, v)
    b = lv.field("tps_b")
    # every dropped-at-the-border operator still annihilates affine fields
    assert np.abs(b).max() < 1e-4
    lq = _level(oracle, w, h, i0, i1, np.stack([0.01 * x * x, 0 * x], -1).astype(np.float32))
    # v.x = a x^2: dxx = 2a everywhere, so the gradient lives only where operators are dropped
    bq = lq.field("tps_b")
    assert np.abs(bq[4:-4, 4:-4]).max() < 1e-3 and np.abs(bq[:, :2]).max() > 0.01
    assert np.array_equal(lv.field("counter")[0, :3], [9, 12, 15])
    assert lv.field("counter")[5, 5] == 25 and lv.field("tps_axy")[5, 5] == 20.0
    m = lv.field("impmask")
    assert m[0].max() == 0 and m[:, 0].max() == 0 and m[1, 1] == (1 << 25) - 1


def test_identical_images_do_not_move(oracle):
    """KAT 1: identical images, v0 = 0: SSIM sits at 1, nothing beyond rounding flukes moves
    and the level converges at once"""
    w, h = 69, 42
    i0, _ = synth.make_pair(w, h)
    lv = _level(oracle, w, h, i0, i0)
    assert np.abs(lv.field("value") - 1).max() < 1e-5
    P = oracle.default_params()
    it = lv.optimize(P, 50)
    assert it <= 3
    assert np.abs(lv.field("v")).max() < 0.05


def test_known_shift_is_recovered(oracle):
    """KAT 2: img1 = img0 shifted by 2s: the greedy descent moves most of the way to
    v = (s, 0) before it stops improving (it under-shoots on smooth texture: the
    contrast/structure SSIM is flat there), with no motion across the shift"""
    w, h, s = 96, 64, 0.75
    base = max(w, h) / 4.0
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    f = lambda n: (16 + 224 * n / 2.0).astype(np.float32)
    i0 = f(synth.value_noise(x, y, base, octaves=5))
    i1 = f(synth.value_noise(x - 2 * s, y, base, octaves=5))
    pyr = synth.build_pyramid(i0, i1, 3)
    P = oracle.default_params()
    lv = oracle.solve(pyr, P, 300, 1.0, threads=4)
    v = lv.field("v")[8:-8, 8:-8]
    assert 0.6 * s < np.median(v[..., 0]) < 1.1 * s and abs(np.median(v[..., 1])) < 0.1
    assert np.sqrt(((v - [s, 0]) ** 2).sum(-1).mean()) < 0.4 * s     # v = 0 scores s


def test_energy_change_is_zero_at_zero_step_and_even_in_tps(oracle):
    w, h = 32, 24
    i0, i1 = synth.make_pair(w, h)
    lv = _level(oracle, w, h, i0, i1)            # v = 0: tps_b = 0
    P = oracle.default_params(w_ssim=0.0, w_ui=0.0)
    L = oracle.lib()
    e = lambda dx, dy: L.vmo_dbg_energy_change(lv._p, C.byref(P), 10, 10, dx, dy)
    assert e(0.0, 0.0) == 0.0
    # pure thin-plate term: w_tps * axy * |d|^2 with axy = 40/2
    assert e(0.5, 0.0) == pytest.approx(0.05 * 20.0 * 0.25, rel=1e-6)
    assert e(0.3, -0.4) == pytest.approx(e(-0.3, 0.4), rel=1e-6)


def test_foldover_bound_on_identity_field(oracle):
    """KAT 5: v = 0 everywhere: the neighbour ring is the unit square, so a step along
    an axis may go 1 - eps, along the diagonal 1 - eps (ring corner at distance sqrt 2
    is reached at t = 1 per axis)"""
    w, h = 20, 16
    i0, i1 = synth.make_pair(w, h)
    lv = _level(oracle, w, h, i0, i1)
    P = oracle.default_params()
    L = oracle.lib()
    f = lambda gx, gy: L.vmo_dbg_foldover(lv._p, C.byref(P), 8, 8, gx, gy)
    assert f(1.0, 0.0) == pytest.approx(1.0 - 0.01, abs=1e-6)
    assert f(0.0, -1.0) == pytest.approx(1.0 - 0.01, abs=1e-6)
    r = np.float32(np.sqrt(0.5))
    assert f(r, r) == pytest.approx(np.sqrt(2.0) - 0.01, abs=1e-5)
    # a neighbour already displaced towards the pixel shortens the admissible step
    lv.field("v")[8, 9] = (-0.5, 0.0)
    assert f(1.0, 0.0) < 0.99


def test_upsample_affine(oracle):
    """KAT 8: upsampling an affine field reproduces affine x scale away from the clamp"""
    sw, sh, dw, dh = 20, 15, 40, 30
    src, dst = oracle.Level(sw, sh), oracle.Level(dw, dh)
    y, x = np.mgrid[0:sh, 0:sw].astype(np.float32)
    src.field("v")[...] = np.stack([0.1 * x + 0.05 * y + 1, -0.2 * y + 0.3], -1)
    dst.upsample_from(src)
    Y, X = np.mgrid[0:dh, 0:dw].astype(np.float32)
    xs, ys = (X + 0.5) * 0.5 - 0.5, (Y + 0.5) * 0.5 - 0.5
    ref = np.stack([(0.1 * xs + 0.05 * ys + 1) * 2, (-0.2 * ys + 0.3) * 2], -1)
    assert np.abs(dst.field("v") - ref)[1:-1, 1:-1].max() < 1e-5


def test_coarse_solve_against_dense_system(oracle, tables):
    """Morph::cpu_optimize_level: the banded solve equals the dense solve of the matrix
    built independently from the stencil table + UI/bcond diagonals (morph.cu:439-562)"""
    tps = tables[0].astype(np.float64)
    w, h, w0, h0 = 14, 9, 140, 90
    cons = np.array([[20, 20, 28, 24, 1.0], [100, 60, 96, 58, 0.7], [70, 30, 70, 36, 1.0]], np.float32)
    for bcond in (2, 1):
        P = oracle.default_params(bcond=bcond)
        lv = oracle.Level(w, h)
        assert lv.coarse_solve(w0, h0, P, cons) == 0
        n = w * h
        A = np.zeros((n, n))
        B = np.zeros((n, 2))
        cls = lambda p, d: p if p < 2 else (3 if p == d - 2 else (4 if p == d - 1 else 2))
        for y in range(h):
            for x in range(w):
                for i in range(5):
                    for j in range(5):
                        qx, qy = x + j - 2, y + i - 2
                        if 0 <= qx < w and 0 <= qy < h:
                            A[y * w + x, qy * w + qx] += P.w_tps * tps[cls(y, h), cls(x, w), i, j]
        inv_wh = 1.0 / (w * h)
        for lx, ly, rx, ry, wt in cons:
            x0, y0 = (lx + 0.5) / w0 * w - 0.5, (ly + 0.5) / h0 * h - 0.5
            x1, y1 = (rx + 0.5) / w0 * w - 0.5, (ry + 0.5) / h0 * h - 0.5
            cx, cy, vx, vy = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
            for yy in range(int(np.floor(cy)), int(np.ceil(cy)) + 1):
                for xx in range(int(np.floor(cx)), int(np.ceil(cx)) + 1):
                    if 0 <= xx < w and 0 <= yy < h:
                        bw = (1 - abs(yy - cy)) * (1 - abs(xx - cx)) * wt
                        A[yy * w + xx, yy * w + xx] += 2 * bw * P.w_ui * inv_wh
                        B[yy * w + xx] += 2 * bw * P.w_ui * inv_wh * np.array([vx, vy])
        border = [(x, y) for y in range(h) for x in range(w) if x in (0, w - 1) or y in (0, h - 1)]
        corners = [(0, 0), (0, h - 1), (w - 1, 0), (w - 1, h - 1)]
        for (x, y) in (border if bcond == 2 else corners):
            A[y * w + x, y * w + x] += P.w_ui * inv_wh
        ref = np.linalg.solve(A, B).reshape(h, w, 2)
        assert np.abs(ref).max() > 0.05
        assert np.abs(lv.field("v") - ref).max() < 2e-5
    lv.field("v")[...] = 7
    assert lv.coarse_solve(w0, h0, oracle.default_params(), ()) == 0
    assert not lv.field("v").any()                      # B = 0 => v = 0 (morph.cu:565-570)


def test_render_identity_and_blend(oracle):
    """KAT 9: v = 0, u = 0 reproduces the crop; color_from 1 is the rounded blend"""
    from videomorphing_amd import morph
    w, h, ex = 40, 30, 4
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
    z = np.zeros((h, w, 2), np.float32)
    f0, f1 = e0.astype(np.float32), e1.astype(np.float32)
    assert np.array_equal(oracle.render_halfway(w, h, ex, 0.3, 0.5, 0, f0, f1, z, z), rgb0)
    assert np.array_equal(oracle.render_halfway(w, h, ex, 0.3, 0.5, 2, f0, f1, z, z), rgb1)
    out = oracle.render_halfway(w, h, ex, 0.25, 0.5, 1, f0, f1, z, z)
    ref = (rgb0.astype(np.float32) * np.float32(0.75) + rgb1.astype(np.float32) * np.float32(0.25)
           + 0.5).astype(np.uint8)
    assert np.array_equal(out, ref)
    # a constant field v = (2, 1), t = 0.5: p = q, samples ext0 at q - v, ext1 at q + v
    v = np.zeros((h, w, 2), np.float32) + (2, 1)
    out = oracle.render_halfway(w, h, ex, 0.0, 0.5, 0, f0, f1, v, z)
    assert np.array_equal(out[4:-4, 4:-4], rgb0[3:-5, 2:-6])


def test_upscale_result(oracle):
    v = np.random.RandomState(1).randn(12, 16, 2).astype(np.float32)
    assert np.array_equal(oracle.upscale_result(v, 16, 12), v)
    up = oracle.upscale_result(np.ones((12, 16, 2), np.float32), 64, 36)
    assert np.allclose(up[..., 0], 4.0) and np.allclose(up[..., 1], 3.0)


def test_poisson_against_dense_solve(oracle):
    """the CG oracle solves the system PoissonExt.cpp:214-312 assembles: compare with a
    dense solve of the same matrix on a toy canvas"""
    from videomorphing_amd import morph
    w, h, ex = 18, 12, 3
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    e0 = morph.make_extended(rgb0, ex)
    other = morph.make_extended(rgb1, ex)[ex:ex + h, ex:ex + w]
    v = (0.6 * synth.displacement(w, h, amp=1.5)).astype(np.float32)
    filled, typ, n = oracle.poisson_prepare(e0, w, h, ex, other, v, 1)
    cw, ch = w + 2 * ex, h + 2 * ex
    assert n == (typ > 0).sum() and (typ == 2).sum() == cw * ch - w * h
    assert (typ == 1).sum() == 2 * (w + h) - 4
    assert filled[..., 3].max() == 0
    out, it, rr = oracle.poisson_extend(e0, w, h, ex, other, v, 1, tol=1e-10)
    assert rr <= 1e-10
    idx = -np.ones((ch, cw), int)
    idx[typ > 0] = np.arange(n)
    A = np.zeros((n, n))
    B = np.zeros((n, 3))
    f = filled.astype(np.float64)
    marker = lambda p: tuple(p) == (255, 0, 255, 0)

    def g(ya, xa, yb, xb):
        if typ[ya, xa] > 1 and typ[yb, xb] > 1 and not marker(filled[ya, xa]) and not marker(filled[yb, xb]):
            return f[ya, xa, :3] - f[yb, xb, :3]
        return np.zeros(3)
    for y in range(ch):
        for x in range(cw):
            if typ[y, x] == 0:
                continue
            i = idx[y, x]
            if typ[y, x] == 1:
                A[i, i] += 1
                B[i] += f[y, x, :3]
            for (yy, xx, sgn, ga) in ((y - 1, x, 1, (y, x, y - 1, x)), (y, x - 1, 1, (y, x, y, x - 1)),
                                      (y, x + 1, -1, (y, x + 1, y, x)), (y + 1, x, -1, (y + 1, x, y, x))):
                if 0 <= yy < ch and 0 <= xx < cw and typ[yy, xx] > 0:
                    A[i, i] += 1
                    A[i, idx[yy, xx]] -= 1
                    B[i] += sgn * g(*ga)
    X = np.linalg.solve(A, B)
    ref = np.clip(X, 0, 255).astype(np.float32).astype(np.int32)
    got = out[typ > 0][:, :3].astype(np.int32)
    d = np.abs(ref - got)
    assert d.max() <= 1 and (d > 0).mean() < 0.01       # truncation of x.9999 vs (x+1).0
    assert np.array_equal(out[typ == 0], e0[typ == 0])


def test_quadratic_path_closed_forms(oracle):
    """QuadraticPath.cpp:24-223: (1) v = 0 gives J_opt = I, a zero right-hand side and u = 0;
    (2) a uniform stretch v = (a x, 0) gives J_opt = diag(sqrt((1-a)(1+a)), 1) everywhere, a
    right-hand side that lives on the first and last column only, and the linear zero-mean
    u = ((sqrt(1 - a^2) - 1)(x - mean x), 0); (3) the solution has zero mean and satisfies the
    5-point Neumann system it was assembled for"""
    h, w = 24, 40
    u, it, rr = oracle.quadratic_path(np.zeros((h, w, 2), np.float32))
    assert it == 0 and np.all(u == 0)
    a = 0.3
    v = np.zeros((h, w, 2), np.float32)
    v[..., 0] = a * np.arange(w, dtype=np.float32)[None, :]
    u, it, rr, jo = oracle.quadratic_path(v, want_jopt=True)
    s = np.sqrt((1 - a) * (1 + a))
    assert np.allclose(jo[..., 0], s, atol=1e-6) and np.allclose(jo[..., 3], 1, atol=1e-6)
    assert np.allclose(jo[..., 1], 0, atol=1e-6) and np.allclose(jo[..., 2], 0, atol=1e-6)
    x = np.arange(w) - (w - 1) / 2.0
    assert rr <= 1e-10 and np.abs(u[..., 0] - (s - 1) * x[None, :]).max() < 2e-5 and np.abs(u[..., 1]).max() < 2e-5
    # a smooth field: residual of the assembled system, zero mean
    rng = np.random.RandomState(3)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    v = np.stack([1.5 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 1.2 * np.cos(xx / 9.0 + yy / 11.0)], -1).astype(np.float32)
    u, it, rr, jo = oracle.quadratic_path(v, want_jopt=True)
    assert abs(u[..., 0].mean()) < 1e-5 and abs(u[..., 1].mean()) < 1e-5 and np.abs(u).max() > 0.01
    for c, (jx, jy, one_x, one_y) in enumerate(((0, 1, 1.0, 0.0), (2, 3, 0.0, 1.0))):
        fx, fy = jo[..., jx] - one_x, jo[..., jy] - one_y      # the field whose divergence drives channel c
        B = np.zeros((h, w))
        B[1:, :] += fy[1:, :]; B[:, 1:] += fx[:, 1:]
        B[:, :-1] -= fx[:, 1:]; B[:-1, :] -= fy[1:, :]
        U = u[..., c].astype(np.float64)
        A = np.zeros((h, w))
        A[1:, :] += U[1:, :] - U[:-1, :]; A[:-1, :] += U[:-1, :] - U[1:, :]
        A[:, 1:] += U[:, 1:] - U[:, :-1]; A[:, :-1] += U[:, :-1] - U[:, 1:]
        assert np.abs(A - (B - B.mean())).max() < 1e-4 * max(np.abs(B).max(), 1e-3) + 2e-6


def test_commit_orders_are_legal_reorderings(oracle):
    """vmo_set_commit_order: row-major / reversed / column-major / column-major reversed apply the SAME
    commits of a phase in another sequence (the reference leaves it to float atomics,
    morph.cu:951-1015): the window sums then differ in their last bits only -- after one sweep (16 phases)
    the fields agree to < 0.05 px -- and from there the
    trajectories separate, which is what the chaos-floor tests measure.  Order 0 is the default."""
    L = oracle.lib()
    P = oracle.default_params()
    w, h = 96, 40
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.8 * synth.displacement(w, h)).astype(np.float32)

    def run(order, sweeps):
        if order is not None:
            L.vmo_set_commit_order(order)
        lv = oracle.Level(w, h)
        lv.set_images(i0, i1)
        lv.field("v")[...] = v0
        lv.init(0.0)
        for _ in range(sweeps):
            lv.optimize_iter(P)
        return lv.field("v").copy(), lv.field("mean").copy()
    try:
        base = run(None, 1)
        one = [run(o, 1) for o in range(4)]
        assert np.array_equal(base[0].view(np.uint32), one[0][0].view(np.uint32))
        assert np.array_equal(base[1].view(np.uint32), one[0][1].view(np.uint32))
        for a in range(4):
            for b in range(a + 1, 4):
                assert not np.array_equal(one[a][1].view(np.uint32), one[b][1].view(np.uint32)), (a, b)
                # measured: max |dv| 0.005 - 0.008 px after the sweep's 16 phases
                assert np.abs(one[a][0] - one[b][0]).max() < 0.05, (a, b)
    finally:
        L.vmo_set_commit_order(0)


def test_tex8_filter_switch(oracle):
    """vmo_set_tex_filter: the diagnostic restatement of CUDA's linear filter -- bilinear weights in 9-bit
    fixed point, 8 fractional bits (CUDA C Programming Guide, Texture Fetching / Linear Filtering; the
    reference samples through it at morph.cu:212-213, 680-681, 960-961).  Mode 1 rounds the fraction to the
    nearest 1/256 (ties up; 1.0 reachable), mode 2 truncates, mode 0 (default) keeps exact float weights.
    A ramp image makes the filter's output the quantised coordinate itself."""
    L = oracle.lib()
    w, h = 8, 4
    img = np.tile(np.arange(w, dtype=np.float32), (h, 1)).copy()         # T[i, j] = i
    t = lambda x, y=1.5: L.vmo_tex2d(img.ctypes.data, w, h, x, y)
    q = 1.0 / 256
    try:
        assert L.vmo_get_tex_filter() == 0
        assert abs(t(2.5 + 0.3 * q) - (2.0 + 0.3 * q)) < 1e-6              # exact float weights: no staircase
        L.vmo_set_tex_filter(1)
        assert t(2.5) == 2.0 and t(2.5 + q) == 2.0 + q                   # representable fractions are untouched
        assert t(2.5 + 0.3 * q) == 2.0 and t(2.5 + 0.7 * q) == 2.0 + q   # nearest
        assert t(2.5 + 255.6 * q) == 3.0                                 # the fraction reaches 1.0: the next texel exactly
        assert t(2.5 + 0.01) == 2.0 + 3 * q and t(2.5 - 0.01) == 2.0 - 3 * q     # eps = 0.01 px = 2.56 quanta -> 3
        assert t(2.5 + 0.005) == 2.0 + q                                 # 1.28 quanta -> 1
        # both axes: weights quantised independently, products exact
        img2 = (np.arange(h, dtype=np.float32)[:, None] * 16 + np.arange(w, dtype=np.float32)[None, :]).copy()
        t2 = lambda x, y: L.vmo_tex2d(img2.ctypes.data, w, h, x, y)
        assert t2(2.5 + 0.3 * q, 1.5 + 1.6 * q) == 16 * (1 + 2 * q) + 2.0
        L.vmo_set_tex_filter(2)
        assert t(2.5 + 0.7 * q) == 2.0 and t(2.5 + 1.2 * q) == 2.0 + q   # truncation
        assert t(2.5 + 255.6 * q) == 2.0 + 255 * q
        # a short solve under either rule stays close to the exact-weight one and differs from it
        P = oracle.default_params()
        ww, hh = 96, 40
        i0, i1 = synth.make_pair(ww, hh)
        v0 = (0.8 * synth.displacement(ww, hh)).astype(np.float32)

        def run(mode):
            L.vmo_set_tex_filter(mode)
            lv = oracle.Level(ww, hh)
            lv.set_images(i0, i1)
            lv.field("v")[...] = v0
            lv.init(0.0)
            for _ in range(3):
                lv.optimize_iter(P)
            return lv.field("v").copy()
        a, b, c = run(0), run(1), run(2)
        assert not np.array_equal(a, b) and not np.array_equal(b, c)
        assert np.sqrt(((a - b) ** 2).sum(-1).mean()) < 0.1 and np.sqrt(((a - c) ** 2).sum(-1).mean()) < 0.1
    finally:
        L.vmo_set_tex_filter(0)
