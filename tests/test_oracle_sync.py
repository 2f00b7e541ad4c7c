"""Known-answer tests of the oracle's synchronisation stage (oracle/vm_oracle_sync.c): the
reference holds no test for it (parity unpinned), so the restatement is held against the
mathematics the reference's code implements -- A = 2 w_tps (sum D_aa^T D_aa + 2 sum D_ab^T D_ab)
+ diag(UI) built independently here from difference operators -- and against closed forms."""
import itertools
import math

import numpy as np
import pytest

import oracle


def _second_diff(n):
    """(n-2) x n operator rows (1, -2, 1)"""
    D = np.zeros((max(n - 2, 0), n))
    for i in range(n - 2):
        D[i, i:i + 3] = (1, -2, 1)
    return D


def _first_diff(n):
    D = np.zeros((max(n - 1, 0), n))
    for i in range(n - 1):
        D[i, i:i + 2] = (-1, 1)
    return D


def _dense_matrix(w, h, d, w_tps, ui):
    """the matrix CSyncThread::genMatrix assembles, from its definition: second differences along
    each axis (weight 1) and mixed first differences per plane (weight 2), all times 2 w_tps,
    dropping every operator that would leave the grid; index = (z * h + y) * w + x"""
    I = {"x": np.eye(w), "y": np.eye(h), "z": np.eye(d)}
    D2 = {"x": _second_diff(w), "y": _second_diff(h), "z": _second_diff(d)}
    D1 = {"x": _first_diff(w), "y": _first_diff(h), "z": _first_diff(d)}

    def kron3(mz, my, mx):
        return np.kron(mz, np.kron(my, mx))

    ops = [(1.0, kron3(I["z"], I["y"], D2["x"])), (1.0, kron3(I["z"], D2["y"], I["x"])), (1.0, kron3(D2["z"], I["y"], I["x"])),
           (2.0, kron3(I["z"], D1["y"], D1["x"])), (2.0, kron3(D1["z"], D1["y"], I["x"])), (2.0, kron3(D1["z"], I["y"], D1["x"]))]
    A = np.zeros((w * h * d, w * h * d))
    for wt, D in ops:
        A += wt * D.T @ D
    return 2 * w_tps * A + np.diag(ui.ravel().astype(np.float64))


def _rows_as_matrix(w, h, d, w_tps, ui):
    A = np.zeros((w * h * d, w * h * d))
    for z, y, x in itertools.product(range(d), range(h), range(w)):
        data = oracle.sync_row(x, y, z, w, h, d, w_tps, float(ui[z, y, x]))
        for i, j, k in zip(*np.nonzero(data)):
            A[(z * h + y) * w + x, ((z + i - 2) * h + (y + j - 2)) * w + (x + k - 2)] = data[i, j, k]
    return A


@pytest.mark.parametrize("dims", [(7, 6, 8), (6, 6, 6), (5, 4, 3), (9, 3, 2), (4, 4, 1)])
def test_rows_equal_the_operator_definition(dims):
    w, h, d = dims
    rng = np.random.default_rng(1)
    ui = (rng.random((d, h, w)) < 0.1) * rng.random((d, h, w)).astype(np.float32) * 50
    A_ref = _dense_matrix(w, h, d, 0.001, ui)
    A = _rows_as_matrix(w, h, d, 0.001, ui.astype(np.float32))
    assert np.allclose(A, A_ref, rtol=2e-6, atol=1e-9)
    assert np.allclose(A, A.T, rtol=1e-6, atol=1e-9)


def test_interior_row_values():
    w_tps = np.float32(0.001)
    r = oracle.sync_row(5, 5, 5, 12, 12, 12, float(w_tps))
    assert r[2, 2, 2] == pytest.approx(84 * 0.001, rel=1e-6)
    assert r[2, 2, 1] == pytest.approx(-24 * 0.001, rel=1e-6) and r[1, 2, 2] == r[2, 2, 1] == r[2, 3, 2]
    assert r[2, 2, 0] == pytest.approx(2 * 0.001, rel=1e-6) and r[2, 1, 1] == pytest.approx(4 * 0.001, rel=1e-6)
    assert np.count_nonzero(r) == 25
    assert abs(float(r.astype(np.float64).sum())) < 1e-7
    # the three nz-per-page counts the reference sizes its CSR arrays with (SyncThread.cpp:303-307)
    w, h, d = 9, 7, 6
    nnz = [sum(np.count_nonzero(oracle.sync_row(x, y, z, w, h, d, 0.001)) for y in range(h) for x in range(w)) for z in range(d)]
    assert nnz[0] == nnz[-1] == 19 * w * h - 12 * (w + h) + 4
    assert nnz[1] == nnz[-2] == 24 * w * h - 14 * (w + h) + 4
    assert nnz[2] == nnz[3] == 25 * w * h - 14 * (w + h) + 4


def test_states_cover_every_position():
    for n in range(1, 12):
        for p in range(n):
            s = oracle.lib().vmo_sync_state(p, n)
            assert 0 <= s < 5
    w, h, d = 9, 5, 4
    tab = np.zeros((5, 5, 5, 25), np.float32)
    oracle.lib().vmo_sync_table(w, h, d, 0.001, tab.ctypes.data)
    L = oracle.lib()
    taps = [(-2, 0, 0), (-1, -1, 0), (-1, 0, -1), (-1, 0, 0), (-1, 0, 1), (-1, 1, 0), (0, -2, 0), (0, -1, -1), (0, -1, 0),
            (0, -1, 1), (0, 0, -2), (0, 0, -1), (0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 1, -1), (0, 1, 0), (0, 1, 1), (0, 2, 0),
            (1, -1, 0), (1, 0, -1), (1, 0, 0), (1, 0, 1), (1, 1, 0), (2, 0, 0)]
    for z, y, x in itertools.product(range(d), range(h), range(w)):
        r = oracle.sync_row(x, y, z, w, h, d, 0.001)
        row = tab[L.vmo_sync_state(z, d), L.vmo_sync_state(y, h), L.vmo_sync_state(x, w)]
        assert np.array_equal(np.array([r[2 + a, 2 + b, 2 + c] for a, b, c in taps], np.float32), row), (x, y, z)


def test_affine_fields_are_in_the_null_space_and_apply_matches_the_matrix():
    w, h, d = 11, 9, 7
    zz, yy, xx = np.meshgrid(np.arange(d), np.arange(h), np.arange(w), indexing="ij")
    ui = np.zeros((d, h, w), np.float32)
    for p in (np.ones_like(xx), xx, yy, zz, 2 * xx - 3 * yy + zz + 5):
        out = oracle.sync_apply(ui, 0.001, p.astype(np.float32))
        assert np.abs(out).max() < 2e-5
    rng = np.random.default_rng(2)
    p = rng.standard_normal((d, h, w)).astype(np.float32)
    ui[3, 4, 5] = 7.0
    A = _dense_matrix(w, h, d, 0.001, ui)
    assert np.allclose(oracle.sync_apply(ui, 0.001, p).ravel(), A @ p.ravel().astype(np.float64), rtol=1e-4, atol=1e-5)


def test_blocked_dot_is_the_exact_sum_to_double_rounding():
    rng = np.random.default_rng(3)
    for shape in [(3, 5, 7), (9, 17, 70), (20, 33, 41)]:
        a = rng.standard_normal(shape).astype(np.float32)
        b = rng.standard_normal(shape).astype(np.float32)
        exact = math.fsum((a * b).astype(np.float64).ravel())  # float products, exactly summed
        got = oracle.sync_dot(a, b)
        assert got == pytest.approx(exact, rel=2e-7)
        assert got == np.float32(exact) or abs(got - exact) <= abs(exact) * 6e-8


def test_ui_terms():
    # one constraint whose midpoint sits exactly on a voxel: weight 1 there, nothing elsewhere
    w, h, d, w0, h0 = 8, 6, 5, 16, 12
    diag, bx, by, bz = oracle.sync_ui(w, h, d, w0, h0, [(4, 4, 1, 12, 8, 3)], 100.0)
    # level coords: (2, 2, 1) and (6, 4, 3): midpoint (4, 3, 2), half difference (2, 1, 1)
    assert np.count_nonzero(diag) == 1 and diag[2, 3, 4] == 100.0
    assert (bx[2, 3, 4], by[2, 3, 4], bz[2, 3, 4]) == (200.0, 100.0, 100.0)
    # a midpoint between voxels spreads trilinearly; the frame axis is NOT scaled
    diag, bx, by, bz = oracle.sync_ui(w, h, d, w0, h0, [(4, 4, 1, 13, 9, 2)], 100.0)
    assert np.count_nonzero(diag) == 8 and diag.sum() == pytest.approx(100.0, rel=1e-6)
    assert bz.sum() == pytest.approx(50.0, rel=1e-6)
    # two constraints on the same voxel accumulate in order
    diag2, *_ = oracle.sync_ui(w, h, d, w0, h0, [(4, 4, 1, 12, 8, 3), (4, 4, 1, 12, 8, 3)], 100.0)
    assert diag2[2, 3, 4] == 200.0


def _cons_for(w0, h0, d):
    return [(w0 // 4, h0 // 4, 0, w0 // 4 + 4, h0 // 4 + 2, 1), (3 * w0 // 4, h0 // 4, d - 1, 3 * w0 // 4 - 4, h0 // 4, d - 2),
            (w0 // 2, 3 * h0 // 4, d // 2, w0 // 2 + 2, 3 * h0 // 4 - 2, d // 2), (w0 // 4, h0 // 2, d - 1, w0 // 4, h0 // 2 + 4, d - 1),
            (3 * w0 // 4, 3 * h0 // 4, 1, 3 * w0 // 4 + 2, 3 * h0 // 4 + 2, 3)]


def test_cg_reaches_the_dense_solution_and_keeps_the_inherited_field():
    w, h, d = 10, 8, 6
    w0, h0 = 20, 16
    cons = _cons_for(w0, h0, d)
    w_ui, w_tps = 100.0, 0.001
    ui, bx, by, bz = oracle.sync_ui(w, h, d, w0, h0, cons, w_ui)
    A = _dense_matrix(w, h, d, w_tps, ui)
    sol = [np.linalg.solve(A, b.ravel().astype(np.float64)).reshape(d, h, w) for b in (bx, by, bz)]
    x, y, z = (np.zeros((d, h, w), np.float32) for _ in range(3))
    k, res = oracle.sync_solve_level(x, y, z, w0, h0, cons, w_ui, w_tps, 3000.0)
    assert k == 3001
    for got, want in zip((x, y, z), sol):
        assert np.abs(got - want).max() < 2e-2 * max(1.0, np.abs(want).max())
    # the quirk: r starts as b although the field is not zero -> the level ADDS A^-1 b
    x2 = np.full((d, h, w), 3.0, np.float32)
    y2, z2 = np.zeros_like(x2), np.zeros_like(x2)
    oracle.sync_solve_level(x2, y2, z2, w0, h0, cons, w_ui, w_tps, 3000.0)
    assert np.abs((x2 - 3.0) - x).max() < 1e-3 * max(1.0, np.abs(x).max())
    # iteration count: floor(max_iter) + 1 passes of the loop
    x3, y3, z3 = (np.zeros((d, h, w), np.float32) for _ in range(3))
    assert oracle.sync_solve_level(x3, y3, z3, w0, h0, cons, w_ui, w_tps, 2.5)[0] == 3
    # without constraints b = 0: nothing moves
    x4, y4, z4 = (np.zeros((d, h, w), np.float32) for _ in range(3))
    oracle.sync_solve_level(x4, y4, z4, w0, h0, [], w_ui, w_tps, 10.0)
    assert not x4.any() and not y4.any() and not z4.any()


def test_level_geometry():
    # 1080p x 60 frames, start_res 32 / 2 (UI/MdiEditor.cpp:1837): worked by hand from pyramid.cu:143-163
    assert oracle.sync_levels(1920, 1080, 60, 16) == [(1920, 1080, 60), (344, 193, 60), (172, 193, 60), (86, 97, 60), (43, 49, 60), (22, 25, 60)]
    # below 4 M voxels nothing is decimated
    assert oracle.sync_levels(256, 128, 10, 32) == [(256, 128, 10), (256, 128, 10), (128, 128, 10), (64, 64, 10), (32, 32, 10)]
    assert oracle.sync_levels(64, 64, 4, 64)[1:] == [(64, 64, 4)]


def test_upsample_and_result_delivery():
    src = np.full((5, 7), 2.5, np.float32)
    assert np.array_equal(oracle.sync_upsample(src, 14, 10, 2.0), np.full((10, 14), 5.0, np.float32))
    # an affine field stays affine away from the clamped border
    yy, xx = np.mgrid[0:6, 0:8].astype(np.float32)
    up = oracle.sync_upsample(xx, 16, 12, 1.0)
    want = (np.arange(16, dtype=np.float32) + 0.5) / 2 - 0.5
    assert np.allclose(up[:, 1:-1], np.broadcast_to(want, (12, 16))[:, 1:-1], atol=1e-5)
    # result delivery: same size = identity times the ratios; constants stay constants
    X, Y, Z = (np.random.default_rng(4).standard_normal((6, 8)).astype(np.float32) for _ in range(3))
    r = oracle.sync_result(X, Y, Z, 8, 6)
    assert np.array_equal(r[..., 0], X) and np.array_equal(r[..., 2], Z) and not r[..., 3].any()
    r = oracle.sync_result(np.full((6, 8), 1.0, np.float32), np.full((6, 8), 2.0, np.float32), np.full((6, 8), 3.0, np.float32), 24, 12)
    assert np.allclose(r[..., 0], 3.0) and np.allclose(r[..., 1], 4.0) and np.allclose(r[..., 2], 3.0)
    # linear interpolation of a ramp, OpenCV's half-pixel convention
    r = oracle.sync_result(xx, xx, xx, 16, 6)
    assert np.allclose(r[0, 1:-1, 2], want[1:-1], atol=1e-5) and r[0, 0, 2] == 0.0 and r[0, -1, 2] == 7.0


def _video(d, h, w, seed):
    rng = np.random.default_rng(seed)
    v = rng.integers(0, 256, (d, h, w, 4), dtype=np.uint8)
    v[..., 3] = 0
    return v


def test_render_resample_known_answers():
    d, h, w = 5, 12, 16
    v0, v1 = _video(d, h, w, 5), _video(d, h, w, 6)
    zero_flow = np.zeros((d, h, w, 2), np.float32)
    vec = np.zeros((h, w, 4), np.float32)
    # no displacement: fa = 0 gives video0's frame, fa = 1 video1's
    assert np.array_equal(oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 0.0, 2), v0[2, ..., :3])
    assert np.array_equal(oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 1.0, 3), v1[3, ..., :3])
    # a blend truncates after each side
    got = oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 0.5, 1)
    a = ((v0[1, ..., :3].astype(np.float64) + 0.5) * 0.5).astype(np.uint8)
    want = (a + (v1[1, ..., :3].astype(np.float64) + 0.5) * 0.5).astype(np.uint8)
    assert np.array_equal(got, want)
    # a time shift of half a frame with zero flow: the mean of two frames of video0
    vec[..., 2] = 0.5
    got = oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 0.0, 2)
    want = (v0[1, ..., :3].astype(np.float32) * 0.5 + v0[2, ..., :3].astype(np.float32) * 0.5 + 0.5).astype(np.uint8)
    assert np.array_equal(got, want)
    # ... and video1 is shifted the other way
    got = oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 1.0, 2)
    want = (v1[2, ..., :3].astype(np.float32) * 0.5 + v1[3, ..., :3].astype(np.float32) * 0.5 + 0.5).astype(np.uint8)
    assert np.array_equal(got, want)
    # shifts beyond the ends clamp to the first / last frame
    vec[..., 2] = 9.0
    assert np.array_equal(oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 0.0, 2), v0[0, ..., :3])
    assert np.array_equal(oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 1.0, 2), v1[d - 1, ..., :3])
    # the spatial part of the field only says WHERE the time shift is read (at p = q + v for
    # video0, q - v for video1); colours are always fetched at the pixel itself
    vec[...] = 0
    vec[..., 0] = 2.0
    vec[:, 8:, 2] = 1.0
    got = oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 0.0, 2)
    assert np.array_equal(got[:, :6], v0[2, :, :6, :3]) and np.array_equal(got[:, 6:], v0[1, :, 6:, :3])
    got = oracle.render_resample(vec, v0, v1, zero_flow, zero_flow, 1.0, 2)
    assert np.array_equal(got[:, :10], v1[2, :, :10, :3]) and np.array_equal(got[:, 10:], v1[3, :, 10:, :3])
