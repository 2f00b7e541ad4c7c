"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.

EXACT arithmetic mode is required to be BIT-IDENTICAL to the oracle (both sides
use only IEEE +,-,*,/,sqrt in the same order; the commit order the reference
leaves to atomics is fixed row-major on both sides).  FAST mode (fused
multiply-add, v_rcp/v_sqrt) is held to the float tolerances of SURVEY.md
section 8(d), written next to each assertion.
"""
import ctypes as C
import os

import numpy as np
import pytest

from videomorphing_amd import capi, morph, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STATE = ["luma", "mean", "var", "cross", "value", "tps_b", "ui_axy", "ui_b", "counter", "tps_axy"]


def _params(O, **kw):
    return O.default_params(**kw)


def _kp(P):
    k = capi.KernParams()
    for f, _ in capi.KernParams._fields_:
        setattr(k, f, getattr(P, f))
    return k


def _make_level(gpu_ctx, O, w, h, v0=None, cons=(), seed=0, frame=0, P=None):
    """One level on both sides with identical inputs, initialised."""
    P = P or _params(O)
    i0, i1 = synth.make_pair(w, h, frame=frame)
    if v0 is None:
        rng = np.random.RandomState(seed)
        v0 = (0.8 * synth.displacement(w, h) + 0.05 * rng.randn(h, w, 2)).astype(np.float32)
    lo = O.Level(w, h)
    lo.set_images(i0, i1)
    lo.field("v")[...] = v0
    lo.init(P.ssim_clamp)
    lo.splat(w, h, cons)
    gpu_ctx.set_params(_kp(P))
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_levels([(w, h), (max((w + 1) // 2, 5), max((h + 1) // 2, 5))])
    pyr.upload_luma(1, i0, i1)
    pyr[1].v = v0
    ca, n = morph._cons_array(cons)
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, ca, n))
    return lo, pyr, P


def _assert_state_equal(lo, lg, fields=STATE + ["v"], exact=True, tol=0.0):
    for f in fields:
        a, b = lo.field(f), (lg.v if f == "v" else lg.field(f))
        if exact:
            bad = np.flatnonzero(a.view(np.uint32).ravel() != b.view(np.uint32).ravel())
            assert bad.size == 0, "%s: %d of %d words differ, first at %d: %r vs %r" % (
                f, bad.size, a.size, bad[0], a.ravel()[bad[0]], b.ravel()[bad[0]])
        else:
            assert np.allclose(a, b, rtol=tol, atol=tol), "%s: max abs diff %g" % (f, np.abs(a - b).max())
    assert np.array_equal(lo.field("impmask"), lg.field("impmask")), "improving mask differs"


@pytest.mark.parametrize("w,h", [(96, 64), (150, 97), (69, 21), (33, 7), (5, 5)])
def test_init_level_exact(gpu_ctx, oracle, w, h):
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    cons = [(10, 10, 14, 12, 1.0), (w - 8.0, h - 3.0, w - 10.0, h - 4.0, 0.5)] if w > 40 else []
    lo, pyr, _ = _make_level(gpu_ctx, oracle, w, h, cons=cons)
    _assert_state_equal(lo, pyr[1])


@pytest.mark.parametrize("w,h,iters", [(96, 64, 3), (150, 97, 2), (200, 40, 2), (64, 16, 2)])
def test_sweep_exact(gpu_ctx, oracle, w, h, iters):
    """bit-tight sweeps: identical accept/reject set, identical state"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    lo, pyr, P = _make_level(gpu_ctx, oracle, w, h)
    for it in range(iters):
        imp_o = lo.optimize_iter(P)
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, C.byref(pr)))
        assert pr.iters == 1 and pr.improving == imp_o
        _assert_state_equal(lo, pyr[1])


@pytest.mark.parametrize("w,h,parts,threads", [(96, 64, 8, 1024), (150, 97, 3, 256), (69, 21, 1, 512), (200, 40, 5, 1024)])
def test_sweep_exact_split_schedule(gpu_ctx, oracle, w, h, parts, threads):
    """the SPLIT schedule (decide/commit kernels, a tile's candidates spread over `parts`
    workgroups), the STEP schedule (one launch per phase) and the PASS schedule (one launch per
    pass, tile-local barriers; parts == 1: its write-through hand-off) give the same bits as the
    oracle, and mixing schedules between calls is seamless (all work on the same state in HBM)"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    cons = synth.make_constraints(w, h, 4)
    P = _params(oracle, bcond=capi.BCOND_BORDER)
    lo, pyr, P = _make_level(gpu_ctx, oracle, w, h, cons=cons, P=P)
    try:
        for it, mode in enumerate([capi.SWEEP_SPLIT, capi.SWEEP_STEP, capi.SWEEP_PASS, capi.SWEEP_TILE, capi.SWEEP_PASS, capi.SWEEP_STEP,
                                   capi.SWEEP_SPLIT, capi.SWEEP_PASS]):
            gpu_ctx.set_tuning(mode, threads, parts)
            imp_o = lo.optimize_iter(P)
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, C.byref(pr)))
            assert pr.iters == 1 and pr.improving == imp_o and pr.commits > 0
            _assert_state_equal(lo, pyr[1])
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)


def _state_bits(lvl):
    return [lvl.field(n).view(np.uint32).copy() for n in ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")]


@pytest.mark.parametrize("w,h,bcond,ncons", [(120, 68, capi.BCOND_NONE, 0), (150, 97, capi.BCOND_BORDER, 4), (69, 21, capi.BCOND_CORNER, 0), (333, 47, capi.BCOND_NONE, 3)])
def test_fast_step_schedule_is_bit_identical_to_split(gpu_ctx, oracle, w, h, bcond, ncons):
    """FAST: the STEP schedule (one launch per phase, the commit folded into the next phase's
    launch from a second copy of the sums) must give the bits of the two-kernel SPLIT schedule
    -- same lean line search, the fold defined by image coordinates -- call after call, with a
    TILE call in between (all three share the state in HBM)"""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    cons = synth.make_constraints(w, h, ncons) if ncons else ()
    out = []
    try:
        for sched in (capi.SWEEP_SPLIT, capi.SWEEP_STEP, capi.SWEEP_PASS):
            P = _params(oracle, bcond=bcond)
            lo, pyr, P = _make_level(gpu_ctx, oracle, w, h, cons=cons, P=P)
            trace = []
            for mode, iters in ((sched, 3.0), (capi.SWEEP_TILE, 1.0), (sched, 5.0), (sched, 1.0)):
                gpu_ctx.set_tuning(mode, 0, 0)
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, 1, C.byref(pr)))
                trace.append((_state_bits(pyr[1]), pr.commits, pr.candidates))
            out.append(trace)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    for other in out[1:]:
        for (sa, ca, na), (sb, cb, nb) in zip(out[0], other):
            assert ca == cb and ca > 0 and na == nb
            for a, b in zip(sa, sb):
                assert np.array_equal(a, b)


def test_fast_split_matches_fast_tile_statistically(gpu_ctx, oracle):
    """FAST: the two schedules differ only in lane fan-out (tree-sum order); after 20
    sweeps their fields agree to RMS <= 0.01 px"""
    w, h = 120, 68
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    res = []
    try:
        for mode in (capi.SWEEP_TILE, capi.SWEEP_SPLIT):
            gpu_ctx.set_tuning(mode, 0, 0)
            lo, pyr, P = _make_level(gpu_ctx, oracle, w, h)
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 20.0, None, 0, None))
            res.append(pyr[1].v)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    dv = np.sqrt(((res[0] - res[1]) ** 2).sum(-1))
    assert np.sqrt((dv ** 2).mean()) <= 0.01 and dv.max() < 0.1, (np.sqrt((dv ** 2).mean()), dv.max())


@pytest.mark.parametrize("bcond", [capi.BCOND_NONE, capi.BCOND_CORNER, capi.BCOND_BORDER])
def test_sweep_exact_constraints_bcond(gpu_ctx, oracle, bcond):
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 120, 75
    cons = synth.make_constraints(w, h, 6)
    P = _params(oracle, bcond=bcond)
    lo, pyr, P = _make_level(gpu_ctx, oracle, w, h, cons=cons, P=P)
    for it in range(3):
        lo.optimize_iter(P)
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, None))
    _assert_state_equal(lo, pyr[1])


def test_identical_images_no_motion(gpu_ctx, oracle):
    """KAT: identical images, v0 = 0 -> (almost) nothing moves: the SSIM values sit
    at 1 up to rounding, so only rounding-level flukes can be accepted; GPU and
    oracle agree bit for bit on which"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 96, 64
    i0, _ = synth.make_pair(w, h)
    P = _params(oracle)
    lo = oracle.Level(w, h)
    lo.set_images(i0, i0)
    lo.init(0.0)
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_levels([(w, h), (48, 32)])
    pyr.upload_luma(1, i0, i0)
    pyr[1].v = np.zeros((h, w, 2), np.float32)
    gpu_ctx.set_params(_kp(P))
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
    it_o = lo.optimize(P, 20)
    pr = capi.Progress()
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 20.0, None, 0, C.byref(pr)))
    assert pr.iters == it_o
    _assert_state_equal(lo, pyr[1])
    assert np.abs(pyr[1].v).max() < 0.05
    assert pyr[1].field("value").min() > 0.999


def test_full_solve_exact_256(gpu_ctx, oracle):
    """BASELINE config 1 geometry (256^2, 3 levels), fewer iterations: the whole
    coarse-to-fine solve is bit-identical to the oracle, including iteration counts."""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w = h = 256
    i0, i1 = synth.make_pair(w, h)
    pyr_imgs = synth.build_pyramid(i0, i1, 3)
    P = _params(oracle)
    per = []
    lo = oracle.solve(pyr_imgs, P, 30, 1.0, threads=8, per_level=per)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 30, 1.0, 64
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 64)
    assert pyr.size() == 4
    m = morph.Morph(prm, pyr)
    assert m.calculate_halfway_parametrization() is True
    assert [m.progress[el]["iters"] for el in (2, 1)] == [p[1] for p in per]
    a, b = lo.field("v"), pyr[1].v
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "max |dv| = %g" % np.abs(a - b).max()
    # and the solve is heading for the synthetic displacement (30 sweeps per level only)
    d = synth.displacement(w, h)
    assert np.sqrt(((b - d) ** 2).sum(-1).mean()) < np.sqrt((d ** 2).sum(-1).mean())


@pytest.mark.parametrize("sched", [capi.SWEEP_AUTO, capi.SWEEP_SPARSE])
def test_fast_mode_tolerance(gpu_ctx, oracle, sched):
    """FAST arithmetic (fused multiply-add, v_rcp/v_sqrt, lane fan-out) against the
    oracle -- under the automatic schedule and under TILE + SPARSE -- tolerances stated in pixels
    of the level:
      - after 1 sweep: |dv| <= eps (0.01 px, the line-search resolution) on >= 99 % of
        the pixels and <= 0.02 px everywhere;
      - after 61 sweeps: RMS dv <= 0.02 px, >= 99 % of pixels within 0.05 px, none
        beyond 0.25 px, SSIM energy sum(1 - value) within 2 %.
    (SURVEY.md 8(d) allows RMS 0.05 px / 0.5 % energy on full solves; the energy is
    compared mid-descent here, where it still falls by 1 % per sweep.)"""
    w, h = 160, 120
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_tuning(sched, 0, 0)
    lo, pyr, P = _make_level(gpu_ctx, oracle, w, h)
    lo.optimize_iter(P)
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, None))
    dv = np.abs(lo.field("v") - pyr[1].v).max(-1)
    assert (dv <= 0.01).mean() >= 0.99 and dv.max() <= 0.02, ((dv <= 0.01).mean(), dv.max())
    lo.optimize(P, 60)
    pr = capi.Progress()
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 60.0, None, 0, C.byref(pr)))
    a, b = lo.field("v"), pyr[1].v
    dv = np.sqrt(((a - b) ** 2).sum(-1))
    assert np.sqrt((dv ** 2).mean()) <= 0.02, np.sqrt((dv ** 2).mean())
    assert (dv < 0.05).mean() >= 0.99 and dv.max() < 0.25
    eo = (1 - lo.field("value")).sum()
    eg = (1 - pyr[1].field("value")).sum()
    assert abs(eo - eg) <= 0.02 * eo, (eo, eg)
    gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_fast_full_solve_statistics(gpu_ctx, oracle):
    """a complete FAST solve (256^2, 3 levels, 60 sweeps per level) against the oracle's:
    RMS dv <= 0.05 px, >= 99 % of pixels within 0.25 px"""
    w = h = 256
    i0, i1 = synth.make_pair(w, h)
    lo = oracle.solve(synth.build_pyramid(i0, i1, 3), _params(oracle), 60, 1.0, threads=8)
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor = 60, 1.0
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 64)
    morph.Morph(prm, pyr).calculate_halfway_parametrization()
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    dv = np.sqrt(((lo.field("v") - pyr[1].v) ** 2).sum(-1))
    assert np.sqrt((dv ** 2).mean()) <= 0.05, np.sqrt((dv ** 2).mean())
    assert (dv < 0.25).mean() >= 0.99


def test_batched_solve_equals_individual_solves(gpu_ctx):
    """vm_solve_batch: several frame pairs relaxed by the same launches (grid.z = pair)
    end bit-identical to the pairs solved one at a time, each with its own iteration
    counts (reference semantics per pair)"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 150, 100
    # pair 0 is an identical image pair (amp 0): no move lowers its energy, it stops after one
    # sweep per level while the others run on
    frames = [synth.make_pair(w, h, frame=k, amp=0.45 * k) for k in range(3)]
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 25, 1.0, 32
    single, iters = [], []
    for i0, i1 in frames:
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        m = morph.Morph(prm, pyr)
        m.calculate_halfway_parametrization()
        single.append(pyr[1].v)
        iters.append([m.progress[el]["iters"] for el in sorted(m.progress)])
    batch = []
    for i0, i1 in frames:
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        batch.append(pyr)
    prog = morph.solve_batch(batch, 25, 1.0)
    for k in range(3):
        assert [p["iters"] for p in prog[k]] == iters[k]
        assert np.array_equal(single[k].view(np.uint32), batch[k][1].v.view(np.uint32))
    # the three displacement amplitudes make the pairs stop at different sweeps: the batch keeps
    # one convergence flag per pair
    assert len({tuple(i) for i in iters}) > 1, iters


@pytest.mark.parametrize("sched", [capi.SWEEP_STEP, capi.SWEEP_TILE, capi.SWEEP_SPARSE, capi.SWEEP_PASS])
def test_fast_batched_solve_equals_individual_solves(gpu_ctx, sched):
    """FAST, a fixed schedule: the batch dimension (grid.z = pair; STEP: per-pair ping-pong copies
    and record sets, TILE: graph replays with the device iteration counter) changes no bit"""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 150, 100
    frames = [synth.make_pair(w, h, frame=k, amp=0.4 + 0.5 * k) for k in range(3)]
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 40, 1.0, 32
    try:
        gpu_ctx.set_tuning(sched, 0, 0)
        single, iters = [], []
        for i0, i1 in frames:
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            m = morph.Morph(prm, pyr)
            m.calculate_halfway_parametrization()
            single.append(pyr[1].v)
            iters.append([m.progress[el]["iters"] for el in sorted(m.progress)])
        batch = []
        for i0, i1 in frames:
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            batch.append(pyr)
        prog = morph.solve_batch(batch, 40, 1.0)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    for k in range(3):
        assert [p["iters"] for p in prog[k]] == iters[k]
        assert np.array_equal(single[k].view(np.uint32), batch[k][1].v.view(np.uint32))


def test_upsample_exact(gpu_ctx, oracle):
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    for (dw, dh) in [(97, 75), (128, 64)]:
        sw, sh = (dw + 1) // 2, (dh + 1) // 2
        rng = np.random.RandomState(3)
        v = rng.randn(sh, sw, 2).astype(np.float32)
        src, dst = oracle.Level(sw, sh), oracle.Level(dw, dh)
        src.field("v")[...] = v
        dst.upsample_from(src)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(dw, dh), (sw, sh)])
        pyr[2].v = v
        capi.check(pyr._L.vm_upsample_v(pyr._h, 0, 1))
        assert np.array_equal(dst.field("v").view(np.uint32), pyr[1].v.view(np.uint32))


def test_coarse_solve_matches_oracle(gpu_ctx, oracle):
    """host banded solve vs the oracle's: same linear system, two independent
    implementations -> agree to 1e-4 px; zero constraints -> exactly zero"""
    w, h, w0, h0 = 60, 34, 1920, 1080
    cons = synth.make_constraints(w0, h0, 8)
    for bcond in (capi.BCOND_BORDER, capi.BCOND_CORNER):
        P = _params(oracle, bcond=bcond)
        lo = oracle.Level(w, h)
        assert lo.coarse_solve(w0, h0, P, cons) == 0
        gpu_ctx.set_params(_kp(P))
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(120, 68), (w, h)])
        ca, n = morph._cons_array(cons)
        capi.check(pyr._L.vm_coarse_solve(pyr._h, 1, w0, h0, ca, n))
        a, b = lo.field("v"), pyr[2].v
        assert np.abs(a).max() > 0.1
        assert np.abs(a - b).max() <= 1e-4, np.abs(a - b).max()
    capi.check(pyr._L.vm_coarse_solve(pyr._h, 1, w0, h0, None, 0))
    assert not pyr[2].v.any()


def test_upscale_result_exact(gpu_ctx, oracle):
    rng = np.random.RandomState(5)
    for (w, h, w0, h0) in [(120, 68, 1920, 1080), (64, 48, 64, 48), (50, 40, 101, 77)]:
        v = rng.randn(h, w, 2).astype(np.float32)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
        pyr[1].v = v
        out = np.zeros((h0, w0, 2), np.float32)
        capi.check(pyr._L.vm_upscale_result(pyr._h, 0, w0, h0, out.ctypes.data, 0))
        ref = oracle.upscale_result(v, w0, h0)
        assert np.array_equal(ref.view(np.uint32), out.view(np.uint32)), np.abs(ref - out).max()


def _frame_inputs(w, h, ex, seed=0):
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
    rng = np.random.RandomState(seed)
    v = (synth.displacement(w, h) + 0.1 * rng.randn(h, w, 2)).astype(np.float32)
    return e0, e1, v


@pytest.mark.parametrize("color_from,geo,col", [(0, 0.0, 0.0), (1, 0.5, 0.5), (2, 1.0, 1.0), (1, 0.3, 0.7)])
def test_render_exact(gpu_ctx, oracle, color_from, geo, col):
    """render_halfway: byte-identical frames given identical inputs"""
    w, h, ex = 150, 90, 15
    e0, e1, v = _frame_inputs(w, h, ex)
    u = (0.3 * np.random.RandomState(9).randn(h, w, 2)).astype(np.float32)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, u)
    out = fr.render_halfway(col, geo, color_from)
    ref = oracle.render_halfway(w, h, ex, col, geo, color_from, e0.astype(np.float32),
                                e1.astype(np.float32), v, u)
    assert np.array_equal(out, ref), "differing bytes: %d" % (out != ref).sum()


@pytest.mark.parametrize("kind", ["rough", "large", "shear", "outside", "nan"])
@pytest.mark.parametrize("with_path", [False, True])
def test_render_window_hard_cases(gpu_ctx, oracle, kind, with_path):
    """the renderer serves its 21 taps of v (and u) from an LDS window placed where a tile's pixels land
    (k_render_win, vm_render.hip) and falls back to global gathers per tap outside it -- byte-identical to the oracle
    (render.cu:16-60) also where the window does not help: a rough field (every tap somewhere else), a warp larger than
    any margin, a shear (the tile's pixels land far from its centre's), taps leaving the image on every side (the
    clamped staging against tap2's clamps), a frame size that is no multiple of the tile, and non-finite values in
    the field (the window is placed nowhere useful; no out-of-bounds access, same bytes as the oracle)"""
    w, h, ex = 203, 77, 9
    e0, e1, _ = _frame_inputs(w, h, ex)
    rng = np.random.RandomState(41)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    if kind == "rough":
        v = (14.0 * rng.randn(h, w, 2)).astype(np.float32)
    elif kind == "large":
        v = np.stack([55.0 + 3.0 * np.sin(yy / 7.0), -38.0 + 2.0 * np.cos(xx / 9.0)], -1).astype(np.float32)
    elif kind == "shear":
        v = np.stack([0.9 * (yy - h / 2), 0.7 * (xx - w / 2)], -1).astype(np.float32)
    elif kind == "outside":
        v = np.stack([0.6 * (xx - w / 2) + 30.0 * np.sign(xx - w / 2), 0.8 * (yy - h / 2) + 20.0 * np.sign(yy - h / 2)], -1).astype(np.float32)
    else:
        v = (3.0 * rng.randn(h, w, 2)).astype(np.float32)
        v[::13, ::11, 0] = np.nan
        v[5::17, 3::7, 1] = np.inf
        v[h // 2, w // 2] = (-np.inf, np.nan)      # a tile centre or close to one
    u = (2.5 * rng.randn(h, w, 2)).astype(np.float32) if with_path else None
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, u)
    uo = u if with_path else np.zeros((h, w, 2), np.float32)
    for geo in (0.0, 0.2, 0.5, 0.9, 1.0):
        out = fr.render_halfway(0.3, geo, 1)
        ref = oracle.render_halfway(w, h, ex, 0.3, geo, 1, e0.astype(np.float32), e1.astype(np.float32), v, uo)
        assert np.array_equal(out, ref), (kind, with_path, geo, int((out != ref).sum()))
    fr.close()


def test_render_plain_kernel_agrees(gpu_ctx):
    """the plain gather kernel (VM_RENDER=plain -- also the path of fields of 4 GiB and more, which the window
    kernel's 32-bit texel offsets do not reach) renders the same bytes as the window kernel: a child process with
    the switch set hashes four frames of a 640x360 pair with a rough path, this process hashes its own"""
    import hashlib, subprocess, sys, textwrap
    prog = textwrap.dedent("""
        import hashlib, sys
        import numpy as np
        sys.path.insert(0, %r)
        from videomorphing_amd import capi, morph, synth
        w, h, ex = 640, 360, 24
        rgb0, rgb1 = synth.make_rgb_pair(w, h)
        rng = np.random.RandomState(3)
        v = (4.0 * synth.displacement(w, h) + 0.5 * rng.randn(h, w, 2)).astype(np.float32)
        u = (1.5 * rng.randn(h, w, 2)).astype(np.float32)
        fr = morph.Frame(morph.Context(0, capi.MATH_FAST), w, h, ex)
        hh = hashlib.sha256()
        for path in (None, u):
            fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v, path)
            for geo in (0.15, 0.8):
                hh.update(fr.render_halfway(0.4, geo, 1).tobytes())
        print("HASH", hh.hexdigest())
    """) % ROOT
    def run(mode):
        env = dict(os.environ)
        env.pop("VM_RENDER", None)
        if mode:
            env["VM_RENDER"] = mode
        r = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("HASH")][0]
    assert run("plain") == run(None)


def test_render_identity(gpu_ctx):
    """KAT 9: v = 0, u = 0 reproduces the original crop / the exact blend"""
    w, h, ex = 64, 48, 6
    e0, e1, _ = _frame_inputs(w, h, ex)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, np.zeros((h, w, 2), np.float32), None)
    assert np.array_equal(fr.render_halfway(0.0, 0.5, 0), e0[ex:ex + h, ex:ex + w, :3])
    assert np.array_equal(fr.render_halfway(0.0, 0.5, 2), e1[ex:ex + h, ex:ex + w, :3])


@pytest.mark.parametrize("geo", [0.0, 0.35, 1.0])
def test_render_without_a_quadratic_path_equals_a_zero_one(gpu_ctx, oracle, geo):
    """a frame whose quadratic path was never uploaded or computed (the reference app's state: the stage is commented
    out, UI/MdiEditor.cpp:1898-1903) is rendered without the 21 taps of u -- a zero path stays zero through the
    fixed-point steps (render.cu:16-60), so the bytes equal those of an explicit zero path and the oracle's; uploading
    a path, computing one, and dropping it again switch the form back and forth"""
    w, h, ex = 150, 90, 15
    e0, e1, v = _frame_inputs(w, h, ex)
    zero = np.zeros((h, w, 2), np.float32)
    u = (0.3 * np.random.RandomState(9).randn(h, w, 2)).astype(np.float32)
    ref0 = oracle.render_halfway(w, h, ex, 0.4, geo, 1, e0.astype(np.float32), e1.astype(np.float32), v, zero)
    refu = oracle.render_halfway(w, h, ex, 0.4, geo, 1, e0.astype(np.float32), e1.astype(np.float32), v, u)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    assert np.array_equal(fr.render_halfway(0.4, geo, 1), ref0)          # no path: the short form
    fr.upload(e0, e1, v, zero)
    assert np.array_equal(fr.render_halfway(0.4, geo, 1), ref0)          # an explicit zero path: the long form, same bytes
    fr.upload(e0, e1, v, u)
    assert np.array_equal(fr.render_halfway(0.4, geo, 1), refu)
    fr.upload(e0, e1, v, None)                                           # dropped again: the buffer is cleared
    assert np.array_equal(fr.render_halfway(0.4, geo, 1), ref0) and not fr.download_qpath().any()
    fr.close()


def test_poisson_extend(gpu_ctx, oracle):
    """classification + fill are byte-identical to the oracle; the solved colours
    agree within +-1 level (both solvers truncate a float solution to integers) on
    >= 99.9 % of the extended pixels, relative residual <= 1e-5"""
    w, h, ex = 96, 64, 10
    e0, e1, v = _frame_inputs(w, h, ex)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    for side, ext, other in ((1, e0, e1), (2, e1, e0)):
        crop = other[ex:ex + h, ex:ex + w]
        ref, it_o, rr_o = oracle.poisson_extend(ext, w, h, ex, crop, v, side, tol=1e-9)
        it, rr, ms = fr.poisson_extend(side, tol=1e-5, max_it=5000)
        out = fr.download_ext(side)
        assert rr <= 1e-5
        assert np.array_equal(out[..., 3], ref[..., 3])
        d = np.abs(out[..., :3].astype(np.int32) - ref[..., :3].astype(np.int32))
        assert (d <= 1).mean() >= 0.999 and d.max() <= 2, (d.max(), (d > 1).mean())
        # the interior is untouched
        assert np.array_equal(out[ex + 1:ex + h - 1, ex + 1:ex + w - 1], ext[ex + 1:ex + h - 1, ex + 1:ex + w - 1])


@pytest.mark.parametrize("w,h", [(160, 110), (333, 47), (64, 33)])
def test_quadratic_path(gpu_ctx, oracle, w, h):
    """CQuadraticPath (QuadraticPath.cpp:24-223): the device's multigrid-PCG solution of the
    Neumann system agrees with the oracle's double-precision CG to 2e-3 px (both zero-mean),
    and the renderer consumes it from the frame"""
    ex = 8
    e0, e1, _ = _frame_inputs(w, h, ex)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    v = (synth.displacement(w, h) + np.stack([0.8 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 0.6 * np.cos(xx / 11.0 + yy / 5.0)], -1)).astype(np.float32)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    it, rr, ms = fr.quadratic_path(tol=1e-6)
    u = fr.download_qpath()
    ref, it_o, rr_o = oracle.quadratic_path(v)
    assert rr <= 1e-6 and it <= 60 and rr_o <= 1e-10
    assert np.abs(ref).max() > 0.05
    assert np.abs(u - ref).max() <= 2e-3, np.abs(u - ref).max()
    assert abs(float(u[..., 0].mean())) < 1e-4 and abs(float(u[..., 1].mean())) < 1e-4
    out = fr.render_halfway(0.5, 0.5, 1)
    assert np.array_equal(out, oracle.render_halfway(w, h, ex, 0.5, 0.5, 1, e0.astype(np.float32), e1.astype(np.float32), v, u))
    fr.upload(None, None, v, None)      # a new upload without a path resets it to zero
    assert np.all(fr.download_qpath() == 0)


def test_quadratic_path_at_precision_floor(gpu_ctx, oracle):
    """a smooth field on a larger frame: |u| >> |rhs|, float32 PCG bottoms out near 1e-5 and
    would drift if continued -- the solver returns its best iterate: an unattainable tolerance
    is reported loudly, 1e-4 succeeds and stays within 0.02 px of the oracle"""
    w, h, ex = 480, 270, 8
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    v = (synth.displacement(w, h) + np.stack([0.8 * np.sin(xx / 9.0), 0.6 * np.cos(yy / 5.0)], -1)).astype(np.float32)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(None, None, v, None)
    with pytest.raises(capi.VmError):
        fr.quadratic_path(tol=1e-9, max_it=200)
    it, rr, ms = fr.quadratic_path(tol=1e-4, max_it=200)
    assert rr <= 1e-4 and it <= 40
    ref, it_o, rr_o = oracle.quadratic_path(v, tol=1e-9)
    u = fr.download_qpath()
    assert np.abs(u - ref).max() <= 0.02, np.abs(u - ref).max()
    # a folded field: columns of I - grad v and I + grad v anti-parallel -> 0/0 in the blend
    # (QuadraticPath.cpp:96-101); the reference propagates the NaN, here it is an error
    vf = np.zeros((h, w, 2), np.float32)
    vf[..., 0] = 2.0 * np.arange(w, dtype=np.float32)[None, :]      # dv_x/dx = 2: J0 col = (-1, 0), J1 col = (3, 0)
    fr.upload(None, None, vf, None)
    with pytest.raises(capi.VmError, match="not finite"):
        fr.quadratic_path(tol=1e-4)


def test_errors_are_loud(gpu_ctx, vmlib):
    """error behaviour: bad calls return codes + messages, never crash"""
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_levels([(64, 48), (32, 24)])
    with pytest.raises(capi.VmError) as e:
        capi.check(vmlib.vm_optimize_level(pyr._h, 0, 10.0, None, 0, None))
    assert e.value.code == capi.VM_E_STATE
    with pytest.raises(capi.VmError):
        capi.check(vmlib.vm_init_level(pyr._h, 1, 64, 48, None, 0))  # coarsest level
    with pytest.raises(capi.VmError):
        capi.check(vmlib.vm_level_dims(pyr._h, 7, None, None, None))
    h = C.c_void_p()
    assert vmlib.vm_ctx_create(99, C.byref(h)) == capi.VM_E_INVALID


def test_cancellation(gpu_ctx, oracle):
    """run_flag == 0 stops the solve (Morph m_cb, morph.cu:156,1390)"""
    w, h = 128, 96
    i0, i1 = synth.make_pair(w, h)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor = 400, 1.0
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 32)
    flag = C.c_int(0)
    m = morph.Morph(prm, pyr, flag)
    assert m.calculate_halfway_parametrization() is True
    assert m.progress == {}  # no level ran


def test_max_iter_is_validated_not_looped_on(gpu_ctx):
    """a non-finite or absurd max_iter is a caller error (VM_E_INVALID), not a 2^31-step
    loop or an unbounded allocation; 0 and negatives mean one sweep (do { } while, morph.cu:1378-1390)"""
    w, h = 64, 48
    i0, i1 = synth.make_pair(w, h)
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 16)
    L = pyr._L
    capi.check(L.vm_coarse_solve(pyr._h, pyr.size() - 2, w, h, None, 0))
    capi.check(L.vm_upsample_v(pyr._h, 0, 1))
    capi.check(L.vm_init_level(pyr._h, 0, w, h, None, 0))
    pr = capi.Progress()
    for bad in (float("inf"), float("nan"), 3e9):
        rc = L.vm_optimize_level(pyr._h, 0, bad, None, 0, C.byref(pr))
        assert rc == capi.VM_E_INVALID, (bad, rc)
        assert b"max_iter" in L.vm_last_error()
    for one in (0.0, -5.0, 0.25, 1.0):
        capi.check(L.vm_init_level(pyr._h, 0, w, h, None, 0))
        capi.check(L.vm_optimize_level(pyr._h, 0, one, None, 1, C.byref(pr)))
        assert pr.iters == 1
    capi.check(L.vm_init_level(pyr._h, 0, w, h, None, 0))
    capi.check(L.vm_optimize_level(pyr._h, 0, 2.5, None, 1, C.byref(pr)))
    assert pr.iters == 3 and pr.evaluations >= 6 * pr.candidates > 0     # >= 4 gradient + 2 bracket evaluations each


def test_context_driven_from_fresh_threads(gpu_ctx):
    """The API is driven from worker threads (MatchingThread, thread pools); HIP's current
    device is per thread, so every entry point selects the context's device itself.  On a
    multi-GPU box this runs on the LAST device from threads whose current device is 0."""
    import threading
    import torch
    dev = torch.cuda.device_count() - 1
    ctx = morph.Context(dev, capi.MATH_EXACT)
    ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 96, 64
    i0, i1 = synth.make_pair(w, h)
    out, err = {}, []

    def work(tag):
        try:
            pyr = morph.Pyramid(ctx)           # allocates on the context's device from this thread
            pyr.build(i0, i1, 16)
            prm = morph.Parameters()
            prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 6, 1.0, 16
            morph.Morph(prm, pyr).calculate_halfway_parametrization()
            out[tag] = pyr[1].v
        except Exception as e:                  # noqa
            err.append(e)
    for tag in ("a", "b"):
        t = threading.Thread(target=work, args=(tag,))
        t.start()
        t.join()
    assert not err, err
    work("main")
    assert np.array_equal(out["a"], out["main"]) and np.array_equal(out["b"], out["main"])
    assert np.abs(out["main"]).max() > 0
    ctx.close()


@pytest.mark.parametrize("sched", [capi.SWEEP_TILE, capi.SWEEP_SPLIT, capi.SWEEP_STEP, capi.SWEEP_PASS])
def test_reversed_commit_order_matches_the_oracle_switch(gpu_ctx, oracle, sched):
    """vm_set_commit_order: the commits of a phase folded in reversed row-major, column-major and
    reversed column-major order -- other orders the reference's atomics may produce -- equal the
    oracle run with the same switch bit for bit, and every one differs from the row-major
    trajectory and from the others (so the switch measures something)"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 138, 84
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.8 * synth.displacement(w, h)).astype(np.float32)
    P = oracle.default_params()
    kp = capi.KernParams()
    for f, _ in capi.KernParams._fields_:
        setattr(kp, f, getattr(P, f))
    gpu_ctx.set_params(kp)
    res = {}
    try:
        gpu_ctx.set_tuning(sched, 0, 0)
        for rev in (0, 1, 2, 3):
            oracle.lib().vmo_set_commit_order(rev)
            gpu_ctx.set_commit_order(rev)
            lo = oracle.Level(w, h)
            lo.set_images(i0, i1)
            lo.field("v")[...] = v0
            lo.init(0.0)
            for _ in range(4):
                lo.optimize_iter(P)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), (69, 42)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 4.0, None, 1, C.byref(pr)))
            for f in ("v", "mean", "var", "cross", "tps_b", "value"):
                assert np.array_equal(lo.field(f).view(np.uint32), pyr[1].field(f).view(np.uint32)), (rev, f)
            res[rev] = pyr[1].field("mean")
    finally:
        oracle.lib().vmo_set_commit_order(0)
        gpu_ctx.set_commit_order(0)
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    for a in range(4):
        for b in range(a + 1, 4):
            assert not np.array_equal(res[a], res[b]), (a, b)
    with pytest.raises(capi.VmError):
        gpu_ctx.set_commit_order(4)


def test_fast_energy_on_converged_solves_sits_at_the_chaos_floor(gpu_ctx, oracle):
    """SURVEY 8(d) asks for the total energy of the GPU solve within 0.5 % of the CPU path's.  Measured on
    CONVERGED small levels (the reference's stopping rule, <= 2000 sweeps; three sizes): two equally legal
    EXACT runs -- the commits of a phase folded row-major (= the oracle, bit for bit) vs reversed, an order
    the reference leaves to float atomics -- end 0.14 - 0.72 % apart in total energy (r03:
    profiles/r03_notes.md), FAST ends 0.01 - 0.83 % from EXACT.  So the bound is stated against that
    floor: mean |E_FAST - E_EXACT| / E_EXACT <= max(0.5 %, 1.5 x the mean floor), no size beyond 1.2 %, and
    the fields themselves within RMS 0.02 px (the floor's own RMS is 0.006 - 0.01 px)."""
    P = _params(oracle)
    dev_fast, dev_floor, rms_fast = [], [], []
    try:
        for (w, h) in ((160, 120), (128, 96), (200, 150)):
            i0, i1 = synth.make_pair(w, h)
            v0 = (0.8 * synth.displacement(w, h) + 0.05 * np.random.RandomState(0).randn(h, w, 2)).astype(np.float32)
            out = {}
            for name, mode, rev in (("exact", capi.MATH_EXACT, 0), ("rev", capi.MATH_EXACT, 1), ("fast", capi.MATH_FAST, 0)):
                gpu_ctx.set_math_mode(mode)
                gpu_ctx.set_commit_order(rev)
                gpu_ctx.set_params(_kp(P))
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
                pyr.upload_luma(1, i0, i1)
                pyr[1].v = v0
                capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 2000.0, None, 0, C.byref(pr)))
                assert pr.improving == 0 and pr.iters < 2000, (name, pr.iters)           # converged
                lv = oracle.Level(w, h)
                lv.set_images(i0, i1)
                lv.field("v")[...] = pyr[1].v
                lv.init(0.0)
                e = lv.energy(P)
                out[name] = (pyr[1].v, float(P.w_ssim * e[0] / (w * h) + P.w_tps * e[1]))
            E = {k: v[1] for k, v in out.items()}
            dev_floor.append(abs(E["rev"] - E["exact"]) / E["exact"])
            dev_fast.append(abs(E["fast"] - E["exact"]) / E["exact"])
            rms_fast.append(float(np.sqrt(((out["fast"][0] - out["exact"][0]) ** 2).sum(-1).mean())))
    finally:
        gpu_ctx.set_commit_order(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    msg = dict(fast=dev_fast, floor=dev_floor, rms=rms_fast)
    assert np.mean(dev_fast) <= max(0.005, 1.5 * np.mean(dev_floor)), msg
    assert max(dev_fast) <= 0.012 and max(rms_fast) <= 0.02, msg


@pytest.mark.parametrize("variant", [capi.MATH_EXACT_FMA, capi.MATH_REF_FASTMATH])
@pytest.mark.parametrize("sched", [capi.SWEEP_TILE, capi.SWEEP_STEP, capi.SWEEP_PASS, capi.SWEEP_AUTO])
def test_legal_arithmetic_variants_are_rounding_level_variants_of_exact(gpu_ctx, oracle, sched, variant):
    """VM_MATH_EXACT_FMA (the EXACT source with fused multiply-adds: nvcc's default --fmad=true) and
    VM_MATH_REF_FASTMATH (the same as the reference's project compiles it, --use_fast_math: contraction +
    approximate division and square root) -- diagnostic builds of the sweep kernels.  Every schedule runs
    under them; after three sweeps from the same start they agree with EXACT to rounding level -- >= 97 % of
    the pixels within 0.01 px, the same activity to 3 % -- and are not identical to it (so they do perturb
    the trajectory)."""
    w, h = 138, 84
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.8 * synth.displacement(w, h)).astype(np.float32)
    P = _params(oracle)
    out = {}
    try:
        gpu_ctx.set_tuning(sched, 0, 0)
        for mode in (capi.MATH_EXACT, variant):
            gpu_ctx.set_math_mode(mode)
            gpu_ctx.set_params(_kp(P))
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), (69, 42)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 3.0, None, 1, C.byref(pr)))
            out[mode] = (pyr[1].v, pr.commits, pr.candidates)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    a, b = out[capi.MATH_EXACT], out[variant]
    d = np.sqrt(((a[0] - b[0]) ** 2).sum(-1))
    assert (d < 0.01).mean() >= 0.97, (d < 0.01).mean()
    assert abs(a[1] - b[1]) <= 0.03 * a[1] and abs(a[2] - b[2]) <= 0.03 * a[2], (a[1:], b[1:])
    assert not np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.abs(b[0] - v0).max() > 0


_TEX8 = [(capi.MATH_REF_TEX8, 1), (capi.MATH_REF_TEX8_TRUNC, 2)]


@pytest.mark.parametrize("mode,rule", _TEX8)
@pytest.mark.parametrize("sched", [capi.SWEEP_TILE, capi.SWEEP_STEP, capi.SWEEP_PASS])
def test_tex8_build_equals_the_oracle_with_the_same_filter(gpu_ctx, oracle, sched, mode, rule):
    """VM_MATH_REF_TEX8 / _TRUNC -- the EXACT source with CUDA's 8-bit bilinear filter weights in every
    texture fetch (the reference: morph.cu:316-322, taps at :212-213, 680-681, 960-961) -- against the oracle
    with vmo_set_tex_filter(rule): init_level and four sweeps bit-identical under every schedule, and
    different from the exact-weight run (the switch does something)."""
    w, h = 138, 84
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.8 * synth.displacement(w, h)).astype(np.float32)
    P = oracle.default_params()
    gpu_ctx.set_params(_kp(P))
    try:
        oracle.lib().vmo_set_tex_filter(rule)
        gpu_ctx.set_math_mode(mode)
        gpu_ctx.set_tuning(sched, 0, 0)
        lo = oracle.Level(w, h)
        lo.set_images(i0, i1)
        lo.field("v")[...] = v0
        lo.init(0.0)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(w, h), (69, 42)])
        pyr.upload_luma(1, i0, i1)
        pyr[1].v = v0
        capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
        for f in ("luma", "mean", "var", "cross", "value", "tps_b"):
            assert np.array_equal(lo.field(f).view(np.uint32), pyr[1].field(f).view(np.uint32)), ("init", f)
        luma_q = pyr[1].field("luma").copy()
        for _ in range(4):
            lo.optimize_iter(P)
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 4.0, None, 1, C.byref(pr)))
        assert pr.commits > 1000
        for f in ("v", "luma", "mean", "var", "cross", "tps_b", "value"):
            assert np.array_equal(lo.field(f).view(np.uint32), pyr[1].field(f).view(np.uint32)), ("sweeps", f)
        # the exact-weight init differs: the lumas are filtered with other weights
        oracle.lib().vmo_set_tex_filter(0)
        lx = oracle.Level(w, h)
        lx.set_images(i0, i1)
        lx.field("v")[...] = v0
        lx.init(0.0)
        d = np.abs(lx.field("luma") - luma_q)
        assert 0 < d.max() < 1.0 and (d > 0).mean() > 0.5, (d.max(), (d > 0).mean())
    finally:
        oracle.lib().vmo_set_tex_filter(0)
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


@pytest.mark.parametrize("mode,rule", _TEX8)
def test_tex8_full_solve_equals_the_oracle_with_the_same_filter(gpu_ctx, oracle, mode, rule):
    """a whole coarse-to-fine solve (150x97, 3 levels: the inter-level upsample samples the coarser field at
    fractions that 8 bits do not hold -- include/util/imgop_upsample.cu:17-31 fetches through a linear-filtered
    texture too) in the TEX8 build = the oracle with the same filter, bit for bit incl. iteration counts"""
    w, h = 150, 97
    i0, i1 = synth.make_pair(w, h)
    P = _params(oracle)
    try:
        oracle.lib().vmo_set_tex_filter(rule)
        per = []
        lo = oracle.solve(synth.build_pyramid(i0, i1, 3), P, 40, 1.0, threads=8, per_level=per)
        oracle.lib().vmo_set_tex_filter(0)
        lx = oracle.solve(synth.build_pyramid(i0, i1, 3), P, 40, 1.0, threads=8)
        gpu_ctx.set_math_mode(mode)
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 40, 1.0, 32
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        assert pyr.size() == 4
        m = morph.Morph(prm, pyr)
        assert m.calculate_halfway_parametrization() is True
        assert [m.progress[el]["iters"] for el in (2, 1)] == [p[1] for p in per]
        a, b = lo.field("v"), pyr[1].v
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "max |dv| = %g" % np.abs(a - b).max()
        assert not np.array_equal(lx.field("v"), b)
    finally:
        oracle.lib().vmo_set_tex_filter(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_batched_solve_with_per_pair_constraints_equals_individual_solves(gpu_ctx):
    """vm_solve_batch_cons: config[4]'s solve -- every pair of the batch with its OWN user point constraints
    (morph.cu:345-388 splat, :471-505 coarse-level terms) and BCOND_BORDER -- ends bit-identical to the pairs
    solved one at a time by vm_solve with the same constraints"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 150, 100
    frames = [synth.make_pair(w, h, frame=k, amp=0.3 + 0.3 * k) for k in range(3)]
    cons = [synth.make_constraints(w, h, 4 + 2 * k, amp=0.3 + 0.3 * k) for k in range(3)]
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = 20, 1.0, 32, capi.BCOND_BORDER
    gpu_ctx.set_params(morph.KernParameters(prm))
    try:
        single = []
        for (i0, i1), c in zip(frames, cons):
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            ca, n = morph._cons_array(c)
            nl = pyr.size() - 2
            prog = (capi.Progress * nl)()
            capi.check(pyr._L.vm_solve(pyr._h, 20.0, 1.0, ca, n, None, 0, prog))
            single.append((pyr[1].v, [prog[k].iters for k in range(nl)]))
        batch = []
        for i0, i1 in frames:
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            batch.append(pyr)
        prog = morph.solve_batch(batch, 20, 1.0, constraints=cons)
        for k in range(3):
            assert [p["iters"] for p in prog[k]] == single[k][1]
            assert np.array_equal(single[k][0].view(np.uint32), batch[k][1].v.view(np.uint32)), k
        # the constraints did something: the same pairs without them end elsewhere
        free = []
        for i0, i1 in frames:
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            free.append(pyr)
        morph.solve_batch(free, 20, 1.0)
        assert not np.array_equal(free[2][1].v, batch[2][1].v)
        # one shared constraint set for all pairs
        shared = []
        for i0, i1 in frames:
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(i0, i1, 32)
            shared.append(pyr)
        morph.solve_batch(shared, 20, 1.0, constraints=cons[0])
        assert np.array_equal(shared[0][1].v.view(np.uint32), single[0][0].view(np.uint32))
    finally:
        gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
