"""End-to-end pipeline of BASELINE config[4] (point constraints + BCOND_BORDER solve ->
result upscale -> Poisson-extended boundary -> warp/blend render), at a size the oracle
finishes in seconds, stage by stage against the oracle; plus the same pipeline at 1080p
checked through properties."""
import numpy as np
import pytest

from videomorphing_amd import capi, morph, synth

pytestmark = pytest.mark.gpu


def _run_gpu(gpu_ctx, w, h, rgb0, rgb1, cons, max_iter, mode):
    gpu_ctx.set_math_mode(mode)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = max_iter, 1.0, 32, capi.BCOND_BORDER
    for c in cons:
        prm.add_point_pair(*c[:4], weight=float(c[4]))
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_rgb(rgb0, rgb1, prm.start_res)
    t = morph.MatchingThread(prm, pyr)
    t.start()
    t.wait()
    ex = int(0.1 * max(w, h))                       # pyramid.cu:194
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), None, None)
    fr.set_v_from_level(pyr, 1)
    return pyr, fr, ex, t


def test_config4_pipeline_against_oracle(gpu_ctx, oracle):
    w, h = 160, 110
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    cons = synth.make_constraints(w, h, 8)
    pyr, fr, ex, t = _run_gpu(gpu_ctx, w, h, rgb0, rgb1, cons, 30, capi.MATH_EXACT)
    v_gpu = pyr._vector[0]
    # --- solve: same luma pyramid (device builder vs oracle builder agree to 2e-3), so the
    # oracle is run on the DEVICE's lumas to compare the solver bit for bit
    nl = pyr.size() - 1
    imgs = [(pyr[el].field("img0"), pyr[el].field("img1")) for el in range(1, nl)]
    imgs.append((np.zeros((pyr[nl].height, pyr[nl].width), np.float32),) * 2)
    P = oracle.default_params(bcond=capi.BCOND_BORDER)
    lo = oracle.solve(imgs, P, 30, 1.0, cons=cons, threads=8)
    v_ref = oracle.upscale_result(lo.field("v"), w, h)
    assert np.array_equal(v_ref.view(np.uint32), v_gpu.view(np.uint32))
    assert np.abs(v_gpu).max() > 0.3 and t.percentage == pytest.approx(100.0)
    # the constraint points pull v towards their half-difference
    c = cons[0]
    cx, cy = int(round((c[0] + c[2]) / 2)), int(round((c[1] + c[3]) / 2))
    assert np.abs(v_gpu[cy, cx] - [(c[2] - c[0]) / 2, (c[3] - c[1]) / 2]).max() < 0.6
    # --- Poisson extension of both sides
    e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
    # both sides sample the ORIGINAL other image (CPoissonExt clones the crops first, :26-27)
    crops = {1: e1[ex:ex + h, ex:ex + w].copy(), 2: e0[ex:ex + h, ex:ex + w].copy()}
    for side, ext in ((1, e0), (2, e1)):
        ref, _, _ = oracle.poisson_extend(ext, w, h, ex, crops[side], v_gpu, side, tol=1e-9)
        fr.poisson_extend(side, tol=1e-6)           # holes far from any anchor need the tight tolerance
        out = fr.download_ext(side)
        d = np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int))
        # max abs colour diff <= 1 (at config[4]'s full size: tests/test_gpu_fullsize_compositor.py, against
        # fixtures of the oracle's CG at 1e-9, at every tolerance bench.py times)
        assert d.max() <= 1, (d.max(), (d > 0).mean())
        (e0, e1)[side - 1][...] = out
    # --- render three morph times from the device-resident canvases
    for tt in (0.0, 0.4, 1.0):
        img = fr.render_halfway(tt, tt, 1)
        ref = oracle.render_halfway(w, h, ex, tt, tt, 1, e0.astype(np.float32), e1.astype(np.float32),
                                    v_gpu, np.zeros_like(v_gpu))
        assert np.array_equal(img, ref)
    assert np.array_equal(fr.render_halfway(0.0, 0.0, 0)[2:-2, 2:-2] // 64, fr.render_halfway(0.0, 0.0, 1)[2:-2, 2:-2] // 64)


def test_config4_pipeline_at_1080p(gpu_ctx):
    """full size, FAST: runs through, the extension is smooth and opaque-free, the frame at
    t = 0 / t = 1 reproduces the sources where v is small"""
    w, h = 1920, 1080
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    cons = synth.make_constraints(w, h, 8)
    try:
        pyr, fr, ex, _ = _run_gpu(gpu_ctx, w, h, rgb0, rgb1, cons, 40, capi.MATH_FAST)
        assert ex == 192
        v = pyr._vector[0]
        assert np.isfinite(v).all() and np.abs(v).max() < 60
        for side in (1, 2):
            it, rr, ms = fr.poisson_extend(side, tol=1e-4)
            out = fr.download_ext(side)
            assert rr <= 1e-4 and out[..., 3].max() == 0
            src = (rgb0, rgb1)[side - 1]
            assert np.array_equal(out[ex + 1:ex + h - 1, ex + 1:ex + w - 1, :3], src[1:-1, 1:-1])
            band = out[:ex, :, :3].astype(np.float32)
            assert np.abs(np.diff(band, axis=1)).mean() < 4.0          # smooth fill, no marker colour left
            assert not ((band[..., 0] == 255) & (band[..., 1] == 0) & (band[..., 2] == 255)).any()
        f0, f1 = fr.render_halfway(0.0, 0.0, 1), fr.render_halfway(1.0, 1.0, 1)
        assert f0.shape == (h, w, 3)
        # at t = 0 the frame is image 0 resampled at p - v(p'), p' = p + v: identity up to interpolation
        assert np.abs(f0.astype(int) - rgb0.astype(int)).mean() < 6.0
        assert np.abs(f1.astype(int) - rgb1.astype(int)).mean() < 6.0
    finally:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def _frame_pair(w, h, ex, seed):
    rng = np.random.RandomState(seed)
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=seed)
    v = (0.6 * synth.displacement(w, h) + 0.2 * rng.randn(h, w, 2)).astype(np.float32)
    return morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v


@pytest.mark.parametrize("w,h,ex", [(160, 110, 16), (333, 47, 33), (20, 12, 3), (40, 30, 6), (96, 64, 10)])
def test_poisson_batch_of_both_sides_and_frames_equals_one_side_at_a_time(gpu_ctx, oracle, w, h, ex):
    """vm_poisson_extend_frames: the 2 n systems of n frames in one batch (blockIdx.z = system) give what n x 2 calls of
    vm_poisson_extend give -- same iteration counts, colours within one level (the double-precision dot products are
    accumulated by atomics, whose order differs from run to run) -- and side 1 of the first frame agrees with the
    oracle's CG like the single call does.  Sizes: hierarchies of one level (26 x 18 canvas), two levels, odd
    canvases, ex > the block width."""
    frames = [_frame_pair(w, h, ex, s) for s in (0, 3, 7)]
    single, batch = [], []
    fr = morph.Frame(gpu_ctx, w, h, ex)
    for e0, e1, v in frames:
        fr.upload(e0, e1, v, None)
        r1, r2 = fr.poisson_extend(1, tol=1e-6), fr.poisson_extend(2, tol=1e-6)
        single.append((r1[0], r2[0], fr.download_ext(1), fr.download_ext(2)))
    fr.close()
    frs = [morph.Frame(gpu_ctx, w, h, ex) for _ in frames]
    for f, (e0, e1, v) in zip(frs, frames):
        f.upload(e0, e1, v, None)
    res, ms = morph.poisson_extend_frames(frs, tol=1e-6)
    assert ms > 0
    for f, r, s in zip(frs, res, single):
        assert (r[0][0], r[1][0]) == (s[0], s[1]), (r, s[:2])
        assert r[0][1] <= 1e-6 and r[1][1] <= 1e-6
        for side in (1, 2):
            a, b = f.download_ext(side).astype(int), s[1 + side].astype(int)
            assert np.array_equal(a[..., 3], b[..., 3]) and np.abs(a - b).max() <= 1, (side, np.abs(a - b).max())
    e0, e1, v = frames[0]
    ref, _, _ = oracle.poisson_extend(e0, w, h, ex, e1[ex:ex + h, ex:ex + w].copy(), v, 1, tol=1e-9)
    d = np.abs(frs[0].download_ext(1)[..., :3].astype(int) - ref[..., :3].astype(int))
    assert d.max() <= 1, (d.max(), (d > 0).mean())
    # both sides of one frame through the frame's own method
    (i1, rr1), (i2, rr2), _ = frs[1].poisson_extend_both(tol=1e-6)          # already extended: alpha is 0 everywhere now
    assert (i1, i2) == (0, 0) and rr1 == 0 and rr2 == 0
    for f in frs:
        f.close()


def test_poisson_batch_with_an_irregular_outside_region(gpu_ctx, oracle):
    """the block lists are built from the canvas' alpha, not from the frame rectangle: holes of alpha > 0 INSIDE the
    image (unknowns in otherwise interior blocks) and an image region that is not a rectangle are solved like the
    oracle solves them"""
    w, h, ex = 200, 90, 20
    e0, e1, v = _frame_pair(w, h, ex, 5)
    e0 = e0.copy()
    e0[ex + 30:ex + 41, ex + 100:ex + 131, 3] = 255          # a hole in the middle of image 0
    e0[ex:ex + 12, ex:ex + 70, 3] = 255                       # a bite out of its corner
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    ref, _, _ = oracle.poisson_extend(e0, w, h, ex, e1[ex:ex + h, ex:ex + w].copy(), v, 1, tol=1e-9)
    (i1, rr1), (i2, rr2), _ = fr.poisson_extend_both(tol=1e-6)
    out = fr.download_ext(1)
    assert i1 > 0 and rr1 <= 1e-6 and out[..., 3].max() == 0
    d = np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int))
    assert d.max() <= 1, (d.max(), (d > 0).mean())
    fr.close()


def test_poisson_batch_refuses_mixed_frames(gpu_ctx):
    a, b = morph.Frame(gpu_ctx, 64, 40, 8), morph.Frame(gpu_ctx, 48, 40, 8)
    with pytest.raises(capi.VmError):
        morph.poisson_extend_frames([a, b])
    with pytest.raises(capi.VmError):
        morph.poisson_extend_frames([a, a])
    a.close()
    b.close()


def test_frame_takes_its_field_from_a_pyramid_of_another_context(gpu_ctx):
    """vm_frame_set_v_from_level across two contexts of ONE device (a solver stream beside the compositor's, as
    bench.py's config[4] pipeline runs them): the field equals the one taken from a pyramid of the frame's own context"""
    w, h = 160, 110
    i0, i1 = synth.make_pair(w, h)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 12, 1.0, 32
    other = morph.Context(0, capi.MATH_EXACT)
    fields = []
    for c in (gpu_ctx, other):
        c.set_math_mode(capi.MATH_EXACT)
        c.set_params(morph.KernParameters(prm))
        pyr = morph.Pyramid(c)
        pyr.build(i0, i1, 32)
        morph.solve_batch([pyr], 12, 1.0)
        fr = morph.Frame(gpu_ctx, w, h, 8)
        fr.set_v_from_level(pyr, 1)
        fields.append(fr.download_v())
        fr.close()
        pyr.clear()
    other.close()
    assert np.array_equal(fields[0].view(np.uint32), fields[1].view(np.uint32)) and np.abs(fields[0]).max() > 0.1


def test_two_compositor_lanes_side_by_side(gpu_ctx):
    """the compositor on two contexts driven by two host threads at once (bench.py's config[4] pipeline):
    every frame's extension (both sides, two frames per batch), quadratic path and three renders give what the same
    frames give one lane after the other -- same PCG iteration counts, rendered bytes within one level (the dot
    products are accumulated by atomics) -- over several rounds, so that the lanes' launches really interleave"""
    from concurrent.futures import ThreadPoolExecutor
    w, h, ex = 320, 200, 24
    data = [_frame_pair(w, h, ex, s) for s in (1, 2, 5, 6)]
    lanes = [gpu_ctx, morph.Context(0, capi.MATH_FAST)]
    frs = [[morph.Frame(c, w, h, ex) for _ in range(2)] for c in lanes]

    def work(li, rounds):
        out = None
        for _ in range(rounds):
            for f, (e0, e1, v) in zip(frs[li], data[2 * li:2 * li + 2]):
                f.upload(e0, e1, v, None)
            res, _ = morph.poisson_extend_frames(frs[li], tol=1e-5)
            qp = [f.quadratic_path(tol=1e-4)[0] for f in frs[li]]
            out = ([tuple(s_[0] for s_ in r) for r in res], qp,
                   [f.render_halfway(0.3, g, 1) for f in frs[li] for g in (0.2, 0.5, 0.8)])
        return out

    alone = [work(0, 1), work(1, 1)]
    with ThreadPoolExecutor(max_workers=2) as ex_:
        both = list(ex_.map(lambda li: work(li, 6), (0, 1)))
    for a, b in zip(alone, both):
        assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
        for x, y in zip(a[2], b[2]):
            assert np.abs(x.astype(int) - y.astype(int)).max() <= 1
    for lane in frs:
        for f in lane:
            f.close()
    lanes[1].close()


def test_upload_from_page_locked_host_memory(gpu_ctx):
    """vm_host_register / vm_host_unregister: canvases uploaded from a page-locked buffer arrive unchanged; registering
    twice and unregistering what was never registered are not errors"""
    w, h, ex = 96, 64, 10
    e0, e1, v = _frame_pair(w, h, ex, 2)
    p0 = morph.pin_host(e0)
    morph.pin_host(p0)                                   # already registered: fine
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(p0, e1, v, None)
    assert np.array_equal(fr.download_ext(1), e0) and np.array_equal(fr.download_ext(2), e1)
    morph.unpin_host(p0)
    morph.unpin_host(np.zeros(16, np.uint8))             # never registered: fine
    with pytest.raises(capi.VmError):
        capi.check(capi.load().vm_host_register(None, 16))
    fr.close()


@pytest.mark.parametrize("w,h,ex", [(96, 64, 10), (333, 47, 33), (65, 17, 0), (20, 12, 3)])
def test_canvases_built_on_the_device_equal_pyramid_build(gpu_ctx, w, h, ex):
    """vm_frame_upload_rgb: the extended canvases built on the device from the two RGB8 frames are Pyramid::build's
    (pyramid.cu:186-200: the frame with a zero alpha plane pasted at (ex, ex) into a canvas of (255, 255, 255, 255)) byte for
    byte, pitched input included; the crops the fills sample are the frames (the extension of both sides gives the same
    iteration counts and colours within one level -- atomics -- as after an upload of finished canvases); v is untouched"""
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=3)
    v = (0.5 * synth.displacement(w, h)).astype(np.float32)
    fa, fb = morph.Frame(gpu_ctx, w, h, ex), morph.Frame(gpu_ctx, w, h, ex)
    try:
        fa.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v, None)
        fb.upload(None, None, v, None)
        fb.upload_rgb(rgb0, rgb1)
        assert np.array_equal(fb.download_ext(1), morph.make_extended(rgb0, ex)) and np.array_equal(fb.download_ext(2), morph.make_extended(rgb1, ex))
        assert np.array_equal(fb.download_v(), v)
        # pitched rows (a caller's padded image buffer)
        pad = np.zeros((h, w + 5, 3), np.uint8)
        pad[:, :w] = rgb1
        p0 = np.zeros((h, w + 5, 3), np.uint8)
        p0[:, :w] = rgb0
        capi.check(fb._L.vm_frame_upload_rgb(fb._h, p0.ctypes.data, pad.ctypes.data, 3 * (w + 5)))
        assert np.array_equal(fb.download_ext(1), morph.make_extended(rgb0, ex)) and np.array_equal(fb.download_ext(2), morph.make_extended(rgb1, ex))
        with pytest.raises(capi.VmError):
            capi.check(fb._L.vm_frame_upload_rgb(fb._h, p0.ctypes.data, pad.ctypes.data, 3 * w - 1))
        if ex > 0:
            ra, rb = fa.poisson_extend_both(tol=1e-6), fb.poisson_extend_both(tol=1e-6)
            assert (ra[0][0], ra[1][0]) == (rb[0][0], rb[1][0])
            for side in (1, 2):
                assert np.abs(fa.download_ext(side).astype(int) - fb.download_ext(side).astype(int)).max() <= 1
    finally:
        fa.close()
        fb.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_poisson_on_random_outside_regions(gpu_ctx, oracle, seed):
    """fuzz of the batched solver's geometry handling: canvases whose outside region (alpha > 0) is the frame border plus
    random rectangles and single pixels punched into BOTH images (scattered unknowns, blocks with one unknown, coarse
    cells without any) -- both sides in one batch agree with the oracle's CG within one colour level"""
    rng = np.random.RandomState(seed)
    w, h, ex = 140 + 7 * (seed % 3), 72 + 5 * (seed % 2), 9 + seed % 4
    e0, e1, v = _frame_pair(w, h, ex, seed)
    e0, e1 = e0.copy(), e1.copy()
    for e in (e0, e1):
        for _ in range(6):
            rw, rh = rng.randint(1, 24), rng.randint(1, 12)
            x0, y0 = rng.randint(0, w - rw), rng.randint(0, h - rh)
            e[ex + y0:ex + y0 + rh, ex + x0:ex + x0 + rw, 3] = 255
        for _ in range(10):
            e[ex + rng.randint(0, h), ex + rng.randint(0, w), 3] = 255
    fr = morph.Frame(gpu_ctx, w, h, ex)
    fr.upload(e0, e1, v, None)
    refs = {}
    for side, ext, other in ((1, e0, e1), (2, e1, e0)):
        refs[side], _, _ = oracle.poisson_extend(ext, w, h, ex, other[ex:ex + h, ex:ex + w].copy(), v, side, tol=1e-9)
    (i1, rr1), (i2, rr2), _ = fr.poisson_extend_both(tol=1e-6)
    assert i1 > 0 and i2 > 0 and rr1 <= 1e-6 and rr2 <= 1e-6
    for side in (1, 2):
        out = fr.download_ext(side)
        assert out[..., 3].max() == 0
        d = np.abs(out[..., :3].astype(int) - refs[side][..., :3].astype(int))
        assert d.max() <= 1, (side, d.max(), (d > 0).mean())
    fr.close()


@pytest.mark.parametrize("w,h,ex", [(1, 1698, 1), (2, 3396, 2), (1698, 1, 1), (3, 1700, 1), (5, 1020, 0)])
def test_poisson_on_thin_canvases(gpu_ctx, oracle, w, h, ex):
    """canvases a few cells wide and thousands long: grids of the hierarchy with w = 3 hold more PAIRS of cells
    (ceil(w / 2) h) than the tail's threads are dealt -- 3 x 1700 at level 0, 3 x 1700 as level 1 of a 6 x 3400 canvas -- so
    the one-workgroup tail has to start a level later (VM_MGB_TAIL_PAIRS); rows of one cell; ex = 0 with holes only"""
    rng = np.random.RandomState(w + h)
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=2)
    v = (0.3 * rng.randn(h, w, 2)).astype(np.float32)
    e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
    if ex == 0:
        e0[100:140, 1:3, 3] = 255
        e1[500:520, :, 3] = 255
    fr = morph.Frame(gpu_ctx, w, h, ex)
    try:
        fr.upload(e0, e1, v, None)
        (i1, r1), (i2, r2), _ = fr.poisson_extend_both(tol=1e-6)
        assert 0 < i1 < 40 and 0 < i2 < 40 and r1 <= 1e-6 and r2 <= 1e-6
        for side, ext, other in ((1, e0, e1), (2, e1, e0)):
            ref, _, _ = oracle.poisson_extend(ext, w, h, ex, other[ex:ex + h, ex:ex + w].copy(), v, side, tol=1e-9)
            out = fr.download_ext(side)
            assert out[..., 3].max() == 0
            d = np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int))
            assert d.max() <= 1, (side, d.max())
    finally:
        fr.close()


_CYCLE_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from videomorphing_amd import capi, morph, synth
import oracle
w, h, ex = 300, 180, 40           # canvas 380 x 260: three levels swept by the tile kernels, four in the one-workgroup tail
rng = np.random.RandomState(21)
rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=21)
v = (0.6 * synth.displacement(w, h) + 0.2 * rng.randn(h, w, 2)).astype(np.float32)
e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
e0[ex + 50:ex + 75, ex + 120:ex + 190, 3] = 255
e1[ex:ex + 9, ex + 200:ex + 300, 3] = 255
ctx = morph.Context(0, capi.MATH_FAST)
fr = morph.Frame(ctx, w, h, ex)
fr.upload(e0, e1, v, None)
(i1, r1), (i2, r2), _ = fr.poisson_extend_both(tol=1e-6)
assert 0 < i1 < 40 and 0 < i2 < 40 and r1 <= 1e-6 and r2 <= 1e-6
worst = 0
for side, ext, other in ((1, e0, e1), (2, e1, e0)):
    ref, _, _ = oracle.poisson_extend(ext, w, h, ex, other[ex:ex + h, ex:ex + w].copy(), v, side, tol=1e-9)
    out = fr.download_ext(side)
    assert out[..., 3].max() == 0
    worst = max(worst, int(np.abs(out[..., :3].astype(int) - ref[..., :3].astype(int)).max()))
assert worst <= 1, worst
vs = (0.5 * synth.displacement(w, h) + 0.05 * rng.randn(h, w, 2)).astype(np.float32)
fr.upload(e0, e1, vs, None)
qi = fr.quadratic_path(tol=1e-4)[0]
uo, _, _ = oracle.quadratic_path(vs, tol=1e-10)
dq = float(np.abs(fr.download_qpath() - uo).max())
assert dq <= 2e-3, dq
print("CYCLE_OK", i1, i2, qi, worst, dq)
"""


@pytest.mark.parametrize("nu", ["1", "2", "2,1,3", "1,2,1,2"])
def test_poisson_cycle_variants(nu):
    """the multigrid cycle's sweeps per level are a build-time choice per kind of system (vm_mgb.h: VM_MGB_NU_POISSON /
    _QPATH); VM_MGB_NU overrides it per process.  Every combination the kernels offer -- one or two red-black sweeps each
    way in the tile kernels of level 0 and of the coarser levels, one to nine in the tail -- is the same preconditioned
    CG on the same system: both sides of a frame with holes within one colour level of the oracle's CG, the quadratic
    path within 2e-3 px, whatever the cycle (a fresh process per setting: the library reads the variable once)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VM_MGB_NU=nu)
    r = subprocess.run([sys.executable, "-c", _CYCLE_CHILD, root], env=env, capture_output=True, text=True, timeout=600,
                       stdin=subprocess.DEVNULL)
    assert r.returncode == 0 and "CYCLE_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_streams_overlap_probe_and_context_beside(gpu_ctx):
    """vm_dbg_streams_overlap answers for two contexts of one device whether their streams run side by side (the runtime deals
    streams to hardware queues as it likes; two on one queue -- or on queues that take turns dispatching -- serialise, which
    cost bench.py's two compositor lanes 2.4 -> 3.2 ms per frame in about one process out of five); morph.context_beside
    returns a context that does overlap with all the given ones; misuse is refused"""
    c1, nrej = morph.context_beside(0, capi.MATH_FAST, [gpu_ctx])
    try:
        # The probe is a MEASUREMENT (host-timed do-nothing kernels): what a test may assert is the contract -- a usable
        # context comes back after at most seven rejected candidates, the probe answers with a bool -- not that this process,
        # with the streams of a whole pytest session alive, was dealt a free pipe, nor that two probes of one pair taken a
        # moment apart give the same answer (tools/exp/queue_probe.py and solver_stream_probe.py are where the probe is
        # validated against the work it predicts).
        assert 0 <= nrej < 8
        assert all(isinstance(c1.runs_beside(gpu_ctx), bool) for _ in range(3))
        c2, nrej2 = morph.context_beside(0, capi.MATH_FAST, [gpu_ctx, c1])
        assert 0 <= nrej2 < 8 and isinstance(c2.runs_beside(c1), bool)
        c2.sync()
        c2.close()
        with pytest.raises(capi.VmError):
            gpu_ctx.runs_beside(gpu_ctx)
    finally:
        c1.close()
