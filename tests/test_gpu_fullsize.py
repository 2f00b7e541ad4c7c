"""BASELINE.json's full sizes (config[1] 1080p, config[3] 4K) on the GPU.

Against the oracle's own whole solves through fixtures (tests/golden/full_solve_hashes.json: per-level
iteration counts + SHA-256 of the finest level's state, written by the oracle in the build container):
  * config[1] on twelve synthetic frames and config[3] (4K, 7 levels) on four, plus eight of them in the
    reference binary's texture arithmetic (REF_TEX8)  (test_full_solve_exact_matches_oracle_hashes).
Against the oracle itself, bit for bit (EXACT arithmetic; the oracle needs ~10-20 s of the
box's host cores for each):
  * config[1] as stated -- one 1920x1080 pair, 6 levels, max_iter 500 per level, the
    reference's stopping rule: the whole coarse-to-fine vm_solve equals oracle.solve incl. the
    per-level iteration counts (test_full_solve_exact_1080p);
  * one dense sweep of the 3840x2160 level of config[3] (test_dense_sweep_exact_4k), and the
    TILE and SPLIT schedules walking the same trajectory there (test_schedules_agree_exactly_at_4k).
And through size-independent properties of the solver state, where the run is too long for the
oracle (FAST arithmetic, hundreds of sweeps, batches):
  * window-sum invariants: after any number of sweeps, mean/var/cross of every pixel
    equal the 5x5 (border-clipped) box sums of the stored warped lumas, and tps_b
    equals the thin-plate stencil applied to the final v (both are maintained
    incrementally by commits, so this checks every commit the level ever made);
  * the stored SSIM value is the SSIM of the stored sums;
  * stored lumas are the images sampled at p -/+ v;
  * a level that reports improving == 0 is a fixed point: another sweep changes nothing;
  * the sweep schedules walk the same trajectory (EXACT: TILE / SPLIT; FAST: SPLIT / STEP / PASS,
    TILE / SPARSE).
"""
import ctypes as C
import os

import numpy as np
import pytest

from videomorphing_amd import capi, morph, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _all_cpus():
    """host threads for the oracle: what this process may really use (cgroup quota respected)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, n)


_STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")


def _assert_level_equals_oracle(lo, lv, what):
    for f in _STATE:
        a, b = lo.field(f), lv.field(f)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s: %s differs (%d of %d words)" % (
            what, f, int((a.view(np.uint32) != b.view(np.uint32)).sum()), a.size)


def test_full_solve_exact_1080p(gpu_ctx, oracle):
    """config[1] ITSELF against the oracle: one 1920x1080 frame pair, 6-level pyramid
    (start_res 32), max_iter 500 per level, drop 1, the reference's stopping rule
    (morph.cu:150-168, 1353-1391) -- the EXACT vm_solve is bit-identical to oracle.solve on all
    host cores: per-level iteration counts, the final halfway field and every state array of
    the finest level."""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 1920, 1080
    i0, i1 = synth.make_pair(w, h)
    pyr_imgs = synth.build_pyramid(i0, i1, 6)
    P = oracle.default_params()
    per = []
    lo = oracle.solve(pyr_imgs, P, 500, 1.0, threads=_all_cpus(), per_level=per)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 32)
    assert pyr.size() == 7 and (pyr[5].width, pyr[5].height) == (120, 68)
    gpu_ctx.set_params(morph.KernParameters(prm))
    L = pyr._L
    prog = (capi.Progress * 5)()
    capi.check(L.vm_solve(pyr._h, 500.0, 1.0, None, 0, None, 0, prog))       # without clear_level: the state stays readable
    assert [prog[el].iters for el in (4, 3, 2, 1, 0)] == [p[1] for p in per], ([prog[el].iters for el in (4, 3, 2, 1, 0)], per)
    assert per[0][1] == 500                    # the 120x68 level uses all of its 500 iterations
    _assert_level_equals_oracle(lo, pyr[1], "1080p solve")
    assert np.abs(pyr[1].v).max() > 5.0        # ~19 px of displacement were found


def _full_solve_cases():
    import json
    path = os.path.join(ROOT, "tests", "golden", "full_solve_hashes.json")
    doc = json.load(open(path))
    return sorted(doc["solves"].items())


@pytest.mark.parametrize("key,fix", _full_solve_cases(), ids=[k for k, _ in _full_solve_cases()])
def test_full_solve_exact_matches_oracle_hashes(gpu_ctx, key, fix):
    """Whole coarse-to-fine EXACT solves at BASELINE.json's full sizes against the ORACLE's, through fixtures: the
    oracle solved config[1] (1920x1080, 6 levels, max_iter 500, the reference's stopping rule, morph.cu:150-168,
    1353-1391) on synthetic frames 0..11 and config[3] (3840x2160, 7 levels) on frames 0..3 in the build
    container (tests/golden/make_full_solve_hashes.py; 17-36 s per 1080p solve, minutes for 4K) and left the
    per-level iteration counts and SHA-256 of every state array of the finest level in
    tests/golden/full_solve_hashes.json.  The HIP path must reproduce every count and every hash: bit-identical
    fields without running the oracle on the GPU box.  The fixture also fingerprints the synthetic inputs (numpy's
    float64 sin / cos need not round alike on every host CPU); if this host generates other inputs the 1080p cases
    fall back to running the oracle here (as test_full_solve_exact_1080p does), the 4K case is skipped with that
    reason -- neither is a parity failure."""
    import fullsize_hash as FH
    w, h = fix["size"]
    nlev, frame = fix["levels"], fix["frame"]
    i0, i1 = synth.make_pair(w, h, frame=frame)
    want_iters, want = fix["iters_coarse_to_fine"], fix["sha256"]
    if FH.input_hash(i0, i1) != fix["inputs"]:
        if w > 1920:
            pytest.skip("this host's numpy generates other synthetic inputs than the fixture's; the 4K oracle run takes minutes")
        import oracle as O
        per = []
        O.lib().vmo_set_tex_filter(fix.get("tex_filter", 0))
        try:
            lo = O.solve(synth.build_pyramid(i0, i1, nlev), O.default_params(), 500, 1.0, threads=_all_cpus(), per_level=per)
        finally:
            O.lib().vmo_set_tex_filter(0)
        want_iters, want = [int(p[1]) for p in per], FH.state_hashes(lo)
    # (a "/tex8" case: the oracle solved it with CUDA's 8-bit texture filter weights -- the HIP path's VM_MATH_REF_TEX8 build;
    #  a "/cons" case: config[4]'s solver settings -- the frame's point constraints and BCOND_BORDER, morph.cu:345-388, 471-562)
    cons = np.asarray(fix.get("constraints", []), np.float32).reshape(-1, 5)
    pyr = None
    try:        # the context is shared by the session: whatever fails here, it goes back to EXACT with default parameters
        gpu_ctx.set_math_mode(capi.MATH_REF_TEX8 if fix.get("tex_filter", 0) else capi.MATH_EXACT)
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = 500, 1.0, 32, fix.get("bcond", capi.BCOND_NONE)
        gpu_ctx.set_params(morph.KernParameters(prm))
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        assert pyr.size() == nlev + 1
        prog = (capi.Progress * (nlev - 1))()
        cs, ncs = morph._cons_array(cons)
        capi.check(pyr._L.vm_solve(pyr._h, 500.0, 1.0, cs, ncs, None, 0, prog))
        assert [int(prog[el].iters) for el in range(nlev - 2, -1, -1)] == want_iters
        got = FH.state_hashes(pyr[1])
        if "ui_axy" in want:
            got["ui_axy"] = FH.sha(pyr[1].field("ui_axy"))
        assert got == want, {f: (got[f] == want[f]) for f in want}
        assert abs(float(np.abs(pyr[1].v).max()) - fix["max_abs_v"]) < 1e-6 or FH.input_hash(i0, i1) != fix["inputs"]
    finally:
        if pyr is not None:
            pyr.clear()
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
        gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))


def test_constrained_solves_as_one_batch_match_oracle_hashes(gpu_ctx):
    """config[4]'s solver call itself: the four "/cons" fixtures (1080p frames 0, 7, 15, 29; 8 point constraints each, BCOND_BORDER,
    500 per level) solved TOGETHER by one vm_solve_batch_cons in EXACT -- the way bench.py's config[4] pipeline solves a
    rank's share -- reproduce every per-level iteration count and every SHA-256 the oracle left for the single solves:
    a pair's trajectory does not depend on its batch-mates (UI splat + border lock: morph.cu:345-388, 471-562, 648-668)."""
    import fullsize_hash as FH
    cases = [(k, f) for k, f in _full_solve_cases() if k.endswith("/cons")]
    if len(cases) < 2:
        pytest.skip("no constrained full-size fixtures")
    pyrs = []
    try:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = 500, 1.0, 32, capi.BCOND_BORDER
        gpu_ctx.set_params(morph.KernParameters(prm))
        cons = []
        for key, fix in cases:
            w, h = fix["size"]
            i0, i1 = synth.make_pair(w, h, frame=fix["frame"])
            if FH.input_hash(i0, i1) != fix["inputs"]:
                pytest.skip("this host's numpy generates other synthetic inputs than the fixture's")
            q = morph.Pyramid(gpu_ctx)
            q.build(i0, i1, 32)
            pyrs.append(q)
            cons.append(np.asarray(fix["constraints"], np.float32).reshape(-1, 5))
        res = morph.solve_batch(pyrs, 500, 1.0, fixed_work=False, constraints=cons)
        for (key, fix), q, r in zip(cases, pyrs, res):
            nlev = fix["levels"]
            assert [int(r[el]["iters"]) for el in range(nlev - 2, -1, -1)] == fix["iters_coarse_to_fine"], key
            got = FH.state_hashes(q[1])
            got["ui_axy"] = FH.sha(q[1].field("ui_axy"))
            assert got == fix["sha256"], (key, {f: (got[f] == fix["sha256"][f]) for f in got})
    finally:
        for q in pyrs:
            q.clear()
        gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))


def test_dense_sweep_exact_4k(gpu_ctx, oracle):
    """config[3]'s finest level: one dense sweep (every pixel searched: 23 M line searches) of a
    3840x2160 level from a rough start, EXACT, bit-identical to the oracle in every state array"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 3840, 2160
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(4).randn(h, w, 2)).astype(np.float32)
    oracle.lib().vmo_set_threads(_all_cpus())
    P = oracle.default_params()
    lo = oracle.Level(w, h)
    lo.set_images(i0, i1)
    lo.field("v")[...] = v0
    lo.init(0.0)
    st = np.zeros(4)
    lo.optimize_iter(P, st)
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_levels([(w, h), (1920, 1080)])
    pyr.upload_luma(1, i0, i1)
    pyr[1].v = v0
    capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
    pr = capi.Progress()
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, C.byref(pr)))
    assert pr.commits == st[2] > 1e6 and pr.candidates == st[1], (pr.commits, pr.candidates, st)
    _assert_level_equals_oracle(lo, pyr[1], "4K sweep")


def test_schedules_agree_exactly_at_4k(gpu_ctx):
    """EXACT arithmetic, two sweeps of the 3840x2160 level from the same start: the TILE and the
    SPLIT schedule produce identical bits (schedule independence at config[3]'s size)"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 3840, 2160
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(4).randn(h, w, 2)).astype(np.float32)
    out = []
    try:
        for mode in (capi.SWEEP_TILE, capi.SWEEP_SPLIT):
            gpu_ctx.set_tuning(mode, 0, 2)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), (1920, 1080)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 2.0, None, 1, C.byref(pr)))
            out.append(([pyr[1].field(f) for f in _STATE], pr.commits))
            del pyr
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    assert out[0][1] == out[1][1] > 1e6
    for f, a, b in zip(_STATE, out[0][0], out[1][0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f


def _box5(a):
    """border-clipped 5x5 box sum (float64)"""
    a = np.asarray(a, np.float64)
    p = np.pad(a, ((3, 2), (3, 2)) + ((0, 0),) * (a.ndim - 2))
    c = p.cumsum(0).cumsum(1)
    return c[5:, 5:] - c[:-5, 5:] - c[5:, :-5] + c[:-5, :-5]


def _tps_apply(v):
    """interior thin-plate stencil row (stencils.cpp:156-261) applied to v, interior only"""
    k = np.array([[0, 0, 2, 0, 0], [0, 4, -16, 4, 0], [2, -16, 40, -16, 2], [0, 4, -16, 4, 0], [0, 0, 2, 0, 0]], np.float64)
    h, w = v.shape[:2]
    out = np.zeros((h - 4, w - 4, 2))
    for i in range(5):
        for j in range(5):
            if k[i, j]:
                out += k[i, j] * v[i:h - 4 + i, j:w - 4 + j].astype(np.float64)
    return out


def _bilinear(img, x, y):
    h, w = img.shape
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    a, b = x - x0, y - y0
    cx = lambda i: np.clip(i, 0, w - 1)
    cy = lambda i: np.clip(i, 0, h - 1)
    f = img.astype(np.float64)
    return ((1 - a) * (1 - b) * f[cy(y0), cx(x0)] + a * (1 - b) * f[cy(y0), cx(x0 + 1)] +
            (1 - a) * b * f[cy(y0 + 1), cx(x0)] + a * b * f[cy(y0 + 1), cx(x0 + 1)])


@pytest.mark.parametrize("w,h,nlev", [(1920, 1080, 6), (3840, 2160, 7)])
def test_state_invariants_at_full_size(gpu_ctx, w, h, nlev):
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    try:
        i0, i1 = synth.make_pair(w, h)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        assert pyr.size() == nlev + 1
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor = 12, 1.0     # a short solve reaching the finest level
        m = morph.Morph(prm, pyr)
        m.calculate_halfway_parametrization()
        assert m.progress[1]["commits"] > 1000
        lv = pyr[1]
        v, luma = lv.v, lv.field("luma")
        # lumas are the images at p -/+ v
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        assert np.abs(_bilinear(i0, xx - v[..., 0], yy - v[..., 1]) - luma[..., 0]).max() < 2e-2
        assert np.abs(_bilinear(i1, xx + v[..., 0], yy + v[..., 1]) - luma[..., 1]).max() < 2e-2
        # window sums follow their definition after thousands of incremental commits
        mean, var, cross = lv.field("mean"), lv.field("var"), lv.field("cross")
        assert np.abs(_box5(luma) - mean).max() < 0.05                         # sums ~ 4e3
        assert np.abs(_box5(luma.astype(np.float64) ** 2) - var).max() < 8.0   # sums ~ 6e5, f32 ulp .06
        assert np.abs(_box5(luma[..., 0].astype(np.float64) * luma[..., 1]) - cross).max() < 8.0
        # the linear thin-plate term is the stencil applied to v
        tb = lv.field("tps_b")[2:-2, 2:-2]
        assert np.abs(_tps_apply(v) - tb).max() < 5e-3 * max(1.0, np.abs(tb).max())
        # SSIM value is the SSIM of the sums (interior: count 25)
        n = 25.0
        mx, my = mean[4:-4, 4:-4, 0] / n, mean[4:-4, 4:-4, 1] / n
        vx = np.maximum((var[4:-4, 4:-4, 0] - n * mx * mx) / n, 0)
        vy = np.maximum((var[4:-4, 4:-4, 1] - n * my * my) / n, 0)
        cov = (cross[4:-4, 4:-4] - n * mx * my) / n
        ss = np.sqrt(vx * vy)
        ref = np.minimum(1.0, (2 * ss + 58.5225) / (vx + vy + 58.5225) * (np.abs(cov) + 29.26125) / (ss + 29.26125))
        assert np.abs(ref - lv.field("value")[4:-4, 4:-4]).max() < 5e-3
    finally:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_converged_level_is_a_fixed_point(gpu_ctx):
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    try:
        w, h = 960, 540
        i0, i1 = synth.make_pair(w, h)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 32)
        gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
        L, nl = pyr._L, pyr.size() - 1
        capi.check(L.vm_coarse_solve(pyr._h, nl - 1, w, h, None, 0))
        pr = capi.Progress()
        for el in range(nl - 1, 0, -1):          # Morph::calculate_halfway_parametrization by hand,
            capi.check(L.vm_upsample_v(pyr._h, el - 1, el))      # without clear_level
            capi.check(L.vm_init_level(pyr._h, el - 1, w, h, None, 0))
            capi.check(L.vm_optimize_level(pyr._h, el - 1, 300.0, None, 0, C.byref(pr)))
        assert pr.improving == 0 and pr.iters < 300, (pr.iters, pr.improving)   # this level converges
        before = {f: pyr[1].field(f) for f in ("v", "mean", "value", "tps_b", "impmask")}
        assert not before["impmask"].any()
        p2 = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 3.0, None, 1, C.byref(p2)))   # 3 forced sweeps
        assert p2.iters == 3 and p2.commits == 0 and p2.active_tiles == 0
        for f, a in before.items():
            assert np.array_equal(a, pyr[1].field(f)), f
    finally:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_schedules_agree_exactly_at_1080p(gpu_ctx):
    """EXACT arithmetic, one sweep of the 1080p level from the same start: TILE and SPLIT
    schedules produce identical bits (schedule independence at full size)"""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    w, h = 1920, 1080
    i0, i1 = synth.make_pair(w, h)
    # (a perfectly smooth start is a fixed point at this size: the SSIM term carries 1/(W H))
    v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(1).randn(h, w, 2)).astype(np.float32)
    out = []
    try:
        for mode in (capi.SWEEP_TILE, capi.SWEEP_SPLIT):
            gpu_ctx.set_tuning(mode, 0, 2)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), (960, 540)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 1.0, None, 0, C.byref(pr)))
            out.append((pyr[1].v, pyr[1].field("mean"), pyr[1].field("impmask"), pr.commits))
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    assert out[0][3] == out[1][3] > 10000
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_step_schedule_is_bit_identical_to_split_on_a_long_run(gpu_ctx):
    """FAST, the 240x135 level of a 1080p pyramid, 120 fixed-work iterations (1920 phase
    launches, from the dense start down to a handful of candidates): STEP = SPLIT bit for bit,
    and both satisfy the window-sum invariants (every commit the level ever made)"""
    w, h = 240, 135
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(2).randn(h, w, 2)).astype(np.float32)
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    out = []
    try:
        for mode in (capi.SWEEP_SPLIT, capi.SWEEP_STEP):
            gpu_ctx.set_tuning(mode, 0, 0)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), (120, 68)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 120.0, None, 1, C.byref(pr)))
            lv = pyr[1]
            out.append(([lv.field(n).copy() for n in ("v", "luma", "mean", "var", "cross", "value", "tps_b", "impmask")], pr.commits))
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    assert out[0][1] == out[1][1] > 50000
    for a, b in zip(out[0][0], out[1][0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    v, luma, mean, var, cross = out[1][0][:5]
    assert np.abs(_box5(luma) - mean).max() < 2e-2 * 25 and np.abs(_box5(luma ** 2) - var).max() < 4.0 * 25
    assert np.abs(_box5(luma[..., 0] * luma[..., 1]) - cross).max() < 4.0 * 25


@pytest.mark.parametrize("mode", [capi.MATH_FAST, capi.MATH_EXACT])
def test_pass_schedule_is_bit_identical_to_step_on_long_runs(gpu_ctx, mode):
    """The 120x68 and 240x135 levels of a 1080p pyramid, 150 fixed-work iterations (600 PASS
    launches = 2400 phases behind tile-local barriers, against 2400 STEP launches): same bits in
    every state array, same counters.  Any stale read across a tile barrier -- a record, a tag, a
    folded window sum, a neighbour's v -- would surface here as a differing word."""
    gpu_ctx.set_math_mode(mode)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    try:
        for (w, h, cw, ch, iters) in [(120, 68, 60, 34, 150), (240, 135, 120, 68, 60 if mode == capi.MATH_EXACT else 150)]:
            i0, i1 = synth.make_pair(w, h)
            v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(3).randn(h, w, 2)).astype(np.float32)
            out = []
            # (PASS, parts = 1: the diagnostic form that stores write-through from the start, as a
            # tile group spread over several XCDs does after its first barrier; parts = 2: tile groups of
            # 32 CONSECUTIVE workgroup ids -- every group really spans all eight XCDs, finds that out from
            # the census at its first barrier, writes its phase-0 stores back and goes write-through)
            for sched, parts in ((capi.SWEEP_STEP, 0), (capi.SWEEP_PASS, 0), (capi.SWEEP_PASS, 1), (capi.SWEEP_PASS, 2)):
                gpu_ctx.set_tuning(sched, 0, parts)
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build_levels([(w, h), (cw, ch)])
                pyr.upload_luma(1, i0, i1)
                pyr[1].v = v0
                capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(iters), None, 1, C.byref(pr)))
                lv = pyr[1]
                out.append(([lv.field(n).copy() for n in _STATE], (pr.commits, pr.candidates, pr.evaluations),
                            pr.sched_launches[4], pr.launches))
            assert out[0][1] == out[1][1] == out[2][1] == out[3][1] and out[0][1][0] > 5000, [o[1] for o in out]
            assert out[0][2] == 0 and out[1][2] >= 4 * iters, out[1][2:]      # the PASS kernel really ran
            assert out[1][3] < out[0][3] / 3                    # a quarter of the launches
            for k in (1, 2, 3):
                for f, a, b in zip(_STATE, out[0][0], out[k][0]):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (w, h, f, k)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_pass_schedule_under_uneven_load_equals_step(gpu_ctx):
    """PASS forced onto a BATCH of 6 pairs of the 120x68 level (48 tile groups = 1536 workgroups,
    several rounds on the chip, groups of different pairs finishing their phases at different
    times): every pair ends bit-identical to the STEP schedule -- the hand-offs hold when
    workgroups wait for compute units, share them and arrive out of step."""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h, npairs, iters = 120, 68, 6, 40
    frames = [synth.make_pair(w, h, frame=k) for k in range(npairs)]
    rng = np.random.RandomState(8)
    v0 = [(0.9 * synth.displacement(w, h) + 0.05 * rng.randn(h, w, 2)).astype(np.float32) for _ in range(npairs)]
    out = {}
    try:
        for sched in (capi.SWEEP_STEP, capi.SWEEP_PASS):
            gpu_ctx.set_tuning(sched, 0, 0)
            batch = []
            for k in range(npairs):
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build_levels([(w, h), (60, 34)])
                pyr.upload_luma(1, *frames[k])
                pyr[1].v = v0[k]
                capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
                batch.append(pyr)
            arr = (C.c_void_p * npairs)(*[p._h for p in batch])
            prog = (capi.Progress * npairs)()
            capi.check(batch[0]._L.vm_optimize_level_batch(arr, npairs, 0, float(iters), None, 1, prog))
            out[sched] = ([[b[1].field(n).copy() for n in _STATE] for b in batch],
                          [(prog[k].commits, prog[k].candidates, prog[k].evaluations) for k in range(npairs)])
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    assert out[capi.SWEEP_STEP][1] == out[capi.SWEEP_PASS][1], (out[capi.SWEEP_STEP][1], out[capi.SWEEP_PASS][1])
    for k in range(npairs):
        for f, a, b in zip(_STATE, out[capi.SWEEP_STEP][0][k], out[capi.SWEEP_PASS][0][k]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, f)


def test_pass_groups_land_on_one_xcd_each(gpu_ctx):
    """speed, not correctness: under the dispatch order observed on MI355X the 32 workgroups of a
    tile group (ids b, b + 8, ...) share an XCD -- recorded by the kernel itself (HW_REG_XCC_ID)"""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 120, 68
    i0, i1 = synth.make_pair(w, h)
    buf = (C.c_uint8 * 256)()
    try:
        capi.check(gpu_ctx._L.vm_dbg_pass_placement(gpu_ctx._h, buf, 256))       # arms the recording
        gpu_ctx.set_tuning(capi.SWEEP_PASS, 0, 0)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(w, h), (60, 34)])
        pyr.upload_luma(1, i0, i1)
        pyr[1].v = (0.9 * synth.displacement(w, h)).astype(np.float32)
        capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 2.0, None, 1, None))
        capi.check(gpu_ctx._L.vm_dbg_pass_placement(gpu_ctx._h, buf, 256))
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    xcc = np.frombuffer(buf, dtype=np.uint8).reshape(32, 8)       # [part][group]
    assert (xcc < 8).all(), xcc
    groups_on_one = sum(len(set(xcc[:, g])) == 1 for g in range(8))
    print("XCD of each tile group's workgroups:", [sorted(set(int(x) for x in xcc[:, g])) for g in range(8)])
    assert groups_on_one >= 6, xcc.T          # observed: all 8; a lost group costs time only


def test_auto_policy_admits_pass_for_a_single_small_level_only(gpu_ctx):
    """VM_SWEEP_AUTO: the 120x68 level of ONE pair (8 tile groups = one 256-workgroup chunk) is swept by
    the PASS schedule, a batch of four such levels by STEP (32 groups would run as four chunks in
    turn), a 1080p level by neither; vm_progress.sched_launches says which kernels ran"""
    if os.environ.get("VM_NO_PASS"):
        pytest.skip("VM_NO_PASS set")
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    try:
        def make(w, h, k):
            i0, i1 = synth.make_pair(w, h, frame=k)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = (0.9 * synth.displacement(w, h)).astype(np.float32)
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            return pyr
        one = make(120, 68, 0)
        pr = capi.Progress()
        capi.check(one._L.vm_optimize_level(one._h, 0, 12.0, None, 1, C.byref(pr)))
        # 4 launches per iteration + one tail launch per batch of iterations
        assert 48 <= pr.sched_launches[4] <= 60 and pr.sched_launches[2] == 0, list(pr.sched_launches)
        batch = [make(120, 68, k) for k in range(4)]
        arr = (C.c_void_p * 4)(*[p._h for p in batch])
        prog = (capi.Progress * 4)()
        capi.check(batch[0]._L.vm_optimize_level_batch(arr, 4, 0, 12.0, None, 1, prog))
        assert prog[0].sched_launches[4] == 0 and prog[0].sched_launches[2] > 0, list(prog[0].sched_launches)
        # ... and the batch's first pair ends where the same pair ends alone (the schedule is not part of the result)
        assert np.array_equal(batch[0][1].v.view(np.uint32), one[1].v.view(np.uint32))
        big = make(1920, 1080, 0)
        capi.check(big._L.vm_optimize_level(big._h, 0, 3.0, None, 1, C.byref(pr)))
        assert pr.sched_launches[4] == 0 and pr.sched_launches[2] == 0 and pr.sched_launches[0] > 0, list(pr.sched_launches)
    finally:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def test_pass_timeout_under_auto_reruns_the_batch_with_step(gpu_ctx, oracle):
    """under VM_SWEEP_AUTO a PASS tile barrier that times out must not turn a valid call into an
    error or leave the level half-updated.  vm_dbg_pass_force_timeout makes one workgroup of a tile group
    walk away in the first PASS launch; its group waits the full bounded time and raises the error word.
    AUTO then restores the level to where the batch of iterations began, reruns it with STEP and stays off
    PASS for the context: the result is the oracle's, bit for bit (EXACT), one fallback is counted, no
    PASS launch is credited.  The same under a FORCED PASS schedule is reported as VM_E_DEVICE.  Afterwards
    (hook off) PASS is admitted again."""
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    P = oracle.default_params()
    kp = capi.KernParams()
    for f, _ in capi.KernParams._fields_:
        setattr(kp, f, getattr(P, f))
    gpu_ctx.set_params(kp)
    w, h, iters = 120, 68, 5
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.9 * synth.displacement(w, h)).astype(np.float32)
    lo = oracle.Level(w, h)
    lo.set_images(i0, i1)
    lo.field("v")[...] = v0
    lo.init(0.0)
    for _ in range(iters):
        lo.optimize_iter(P)

    def run(sched):
        gpu_ctx.set_tuning(sched, 0, 0)
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build_levels([(w, h), (60, 34)])
        pyr.upload_luma(1, i0, i1)
        pyr[1].v = v0
        capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
        pr = capi.Progress()
        capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(iters), None, 1, C.byref(pr)))
        return pyr, pr
    L = gpu_ctx._L
    before = L.vm_dbg_pass_fallbacks(gpu_ctx._h)
    try:
        capi.check(L.vm_dbg_pass_force_timeout(gpu_ctx._h, 1))
        pyr, pr = run(capi.SWEEP_AUTO)
        assert L.vm_dbg_pass_fallbacks(gpu_ctx._h) == before + 1
        assert pr.sched_launches[4] == 0 and pr.sched_launches[2] > 0, list(pr.sched_launches)
        _assert_level_equals_oracle(lo, pyr[1], "AUTO after a PASS timeout")
        # latched: the next call does not try PASS again
        pyr2, pr2 = run(capi.SWEEP_AUTO)
        assert L.vm_dbg_pass_fallbacks(gpu_ctx._h) == before + 1 and pr2.sched_launches[4] == 0
        _assert_level_equals_oracle(lo, pyr2[1], "AUTO, PASS latched off")
        with pytest.raises(capi.VmError) as ei:
            run(capi.SWEEP_PASS)
        assert "timed out" in str(ei.value)
    finally:
        capi.check(L.vm_dbg_pass_force_timeout(gpu_ctx._h, 0))
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
    if not os.environ.get("VM_NO_PASS"):
        pyr3, pr3 = run(capi.SWEEP_AUTO)
        assert pr3.sched_launches[4] > 0, list(pr3.sched_launches)
        _assert_level_equals_oracle(lo, pyr3[1], "AUTO, PASS admitted again")


_TOKEN_SCRIPT = r"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, %r)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
w, h = 120, 68
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), (60, 34)])
pyr.upload_luma(1, i0, i1)
pyr[1].v = (0.9 * synth.displacement(w, h)).astype(np.float32)
capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
pr = capi.Progress()
capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 10.0, None, 1, C.byref(pr)))
np.save(sys.argv[1], pyr[1].v)
print(pr.sched_launches[4], pr.sched_launches[2])
"""


def test_pass_token_is_exclusive_across_processes(tmp_path):
    """A device shared by several processes is detected, not configured: the PASS token is an advisory
    flock() on a per-device lock file (VM_LOCK_DIR, default /tmp), so while ANOTHER process holds it a
    context runs STEP instead -- same bits -- and takes PASS again once the file is free."""
    import fcntl
    import subprocess
    import sys
    if os.environ.get("VM_NO_PASS"):
        pytest.skip("VM_NO_PASS set")
    env = dict(os.environ, VM_LOCK_DIR=str(tmp_path))

    def child(tag):
        out = str(tmp_path / ("v_%s.npy" % tag))
        r = subprocess.run([sys.executable, "-c", _TOKEN_SCRIPT % ROOT, out], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.load(out), [int(x) for x in r.stdout.split()[-2:]]
    v_free, n_free = child("free")
    assert n_free[0] > 0 and n_free[1] == 0, n_free                 # PASS ran
    locks = [f for f in os.listdir(str(tmp_path)) if f.startswith("vmorph-pass-") and f.endswith(".lock")]
    assert len(locks) == 1, locks
    with open(os.path.join(str(tmp_path), locks[0]), "r") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)                              # "another process" holds the device's token
        v_held, n_held = child("held")
        fcntl.flock(fh, fcntl.LOCK_UN)
    assert n_held[0] == 0 and n_held[1] > 0, n_held                 # STEP instead
    assert np.array_equal(v_free.view(np.uint32), v_held.view(np.uint32))
    v_again, n_again = child("again")
    assert n_again[0] > 0, n_again


def test_wave_wide_line_search_is_bit_identical_to_the_32_lane_one(gpu_ctx):
    """FAST, STEP schedule, 120x68 and 240x135 levels, 150 fixed-work iterations: with 32 workgroups
    per tile a workgroup holds <= 8 candidates and every candidate gets a whole wave (decide64: two
    line-search points per round, the next golden-section step speculated); with 2 workgroups per
    tile the same candidates go through the 32-lane search.  Same bits, same counters -- the
    speculation may only ever change the time."""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    try:
        for (w, h, cw, ch) in [(120, 68, 60, 34), (240, 135, 120, 68)]:
            i0, i1 = synth.make_pair(w, h)
            v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(3).randn(h, w, 2)).astype(np.float32)
            out = []
            for parts in (32, 2):
                gpu_ctx.set_tuning(capi.SWEEP_STEP, 0, parts)
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build_levels([(w, h), (cw, ch)])
                pyr.upload_luma(1, i0, i1)
                pyr[1].v = v0
                capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 150.0, None, 1, C.byref(pr)))
                lv = pyr[1]
                out.append(([lv.field(n).copy() for n in ("v", "luma", "mean", "var", "cross", "value", "tps_b", "impmask")],
                            (pr.commits, pr.candidates, pr.evaluations)))
            assert out[0][1] == out[1][1] and out[0][1][0] > 5000, out[0][1]
            for a, b in zip(out[0][0], out[1][0]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (w, h)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def _oracle_energy(i0, i1, v):
    """E = w_ssim E_ssim / (W H) + w_tps E_tps of a finest-level field (vmo_energy on the host)"""
    import oracle as O
    h, w = i0.shape
    lv = O.Level(w, h)
    lv.set_images(i0, i1)
    lv.field("v")[...] = v
    lv.init(0.0)
    P = O.default_params()
    e = lv.energy(P)
    return float(P.w_ssim * e[0] / (w * h) + P.w_tps * e[1])


CHAOS_FRAMES = (0, 3, 6, 9, 12, 15)      # r03's bench: frames 6 and 12 are the ones whose finest level cycles in FAST
# The family of equally legal runs of the reference algorithm -- the reference's own expressions in the reference's own
# order, every member: IEEE arithmetic without contraction under the four commit orders of vm_set_commit_order
# (row-major = the oracle, reversed, column-major, c.-m. reversed: the reference leaves the order to float atomics,
# morph.cu:951-1015); with fused multiply-adds (VM_MATH_EXACT_FMA: nvcc's default --fmad=true); and as the reference's
# project file really compiles it, --use_fast_math (VM_MATH_REF_FASTMATH: contraction + approximate division and
# square root, MdiEditor.vcxproj:208-213), each under two of the orders; and (round 5) with every texture fetch filtered
# the way the reference BINARY's tex2D(linear) fetches are -- CUDA's 9-bit fixed-point bilinear weights, 8 fractional
# bits, against a finite-difference step of eps = 0.01 px = 2.56 quanta (VM_MATH_REF_TEX8: morph.cu:316-322, 680-681,
# 763-778; bit-identical to the oracle with vmo_set_tex_filter(1), tests/test_gpu_parity.py), two orders
CHAOS_FAMILY = (("x0", capi.MATH_EXACT, 0), ("x1", capi.MATH_EXACT, 1), ("x2", capi.MATH_EXACT, 2), ("x3", capi.MATH_EXACT, 3),
                ("f0", capi.MATH_EXACT_FMA, 0), ("f2", capi.MATH_EXACT_FMA, 2),
                ("r0", capi.MATH_REF_FASTMATH, 0), ("r2", capi.MATH_REF_FASTMATH, 2),
                ("t0", capi.MATH_REF_TEX8, 0), ("t2", capi.MATH_REF_TEX8, 2))
# (names starting with "t" / "u" are the texture-quantised members: chaos_floor_measure keeps them OUT of the floors
# FAST is judged against -- what rounds 3-4 called the family -- and reports them beside it)


def chaos_floor_measure(ctx, frames=CHAOS_FRAMES, family=CHAOS_FAMILY, w=1920, h=1080, keep_fields_of=None):
    """Per frame of config[1] (1080p, 6 levels, 500 iterations per level, reference stopping rule): the
    final fields of one solve per member of the legal family and of the FAST solve, their pairwise distances
    and their energies (oracle, on the host).  Returns {frame: dict}; also used by tools/dev_chaos_floor.py."""
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
    ctx.set_params(morph.KernParameters(prm))
    res = {}
    try:
        for frame in frames:
            i0, i1 = synth.make_pair(w, h, frame=frame)
            out, its = {}, {}
            for name, mode, order in list(family) + [("fast", capi.MATH_FAST, 0)]:
                ctx.set_math_mode(mode)
                ctx.set_commit_order(order)
                pyr = morph.Pyramid(ctx)
                pyr.build(i0, i1, 32)
                m = morph.Morph(prm, pyr)
                m.calculate_halfway_parametrization()
                out[name] = pyr[1].v
                its[name] = [m.progress[el]["iters"] for el in sorted(m.progress)]
                del pyr
            rms = lambda a, b: float(np.sqrt(((out[a] - out[b]) ** 2).sum(-1).mean()))
            within = lambda a, b: float((np.sqrt(((out[a] - out[b]) ** 2).sum(-1)) < 0.25).mean())
            names = [f[0] for f in family]
            tex = [a for a in names if a[0] in "tu"]         # 8-bit texture weights (the reference BINARY's sampling) ...
            ex = [a for a in names if a[0] not in "tu"]      # ... and the exact-weight members: the family FAST is judged against
            pairs = [(a, b) for k, a in enumerate(ex) for b in ex[k + 1:]]
            opairs = [(a, b) for a, b in pairs if a[0] == "x" and b[0] == "x"]      # commit orders only
            rpairs = [(a, b) for a, b in pairs if (a[0] == "r") != (b[0] == "r")]    # --use_fast_math against IEEE builds
            E = {k: _oracle_energy(i0, i1, out[k]) for k in out}
            Ex = [E[k] for k in ex]
            r = {
                "iters": its,
                "rms_pairs": {a + b: rms(a, b) for a, b in pairs},
                "within_pairs": {a + b: within(a, b) for a, b in pairs},
                "rms_fast": [rms("fast", a) for a in ex],
                "within_fast": [within("fast", a) for a in ex],
                "E_family": Ex, "E_fast": E["fast"],
                # the quantities SURVEY 8(d) names: FAST against the oracle's run (x0), the floor = the family's range
                "rms_floor": max(rms(a, b) for a, b in pairs), "rms_floor_orders_only": max(rms(a, b) for a, b in opairs),
                "rms_fastmath_vs_ieee": [rms(a, b) for a, b in rpairs] if rpairs else None,
                "rms_fast0": rms("fast", ex[0]),
                "within_floor": min(within(a, b) for a, b in pairs), "within_fast0": within("fast", ex[0]),
                "e_floor": (max(Ex) - min(Ex)) / Ex[0], "e_fast0": abs(E["fast"] - Ex[0]) / Ex[0],
                "e_fast_signed": (E["fast"] - float(np.mean(Ex))) / float(np.mean(Ex)),
                # round 5: where the reference binary's texture arithmetic lands -- every TEX8 member against every
                # exact-weight member, the TEX8 members among themselves, FAST against them, their energies (oracle, exact weights)
                "rms_tex8_vs_exact": [rms(a, b) for a in tex for b in ex] if tex else None,
                "rms_tex8_pairs": {a + b: rms(a, b) for k, a in enumerate(tex) for b in tex[k + 1:]} if len(tex) > 1 else None,
                "rms_fast_vs_tex8": [rms("fast", a) for a in tex] if tex else None,
                "within_tex8_vs_exact": [within(a, b) for a in tex for b in ex] if tex else None,
                "E_tex8": [E[a] for a in tex] if tex else None,
                "e_tex8_signed": (float(np.mean([E[a] for a in tex])) - float(np.mean(Ex))) / float(np.mean(Ex)) if tex else None,
            }
            if keep_fields_of is not None and frame == keep_fields_of:
                r["fields"] = (out[ex[0]], out["fast"])
            res[frame] = r
    finally:
        ctx.set_commit_order(0)
        ctx.set_math_mode(capi.MATH_EXACT)
    return res


def chaos_round(r):
    rnd = lambda v: {k: rnd(x) for k, x in v.items()} if isinstance(v, dict) else (np.round(v, 5).tolist() if not isinstance(v, list) or not v or not isinstance(v[0], list) else v)
    return {k: rnd(v) for k, v in r.items()}


def test_fast_solve_sits_at_the_chaos_floor_of_config1(gpu_ctx):
    """config[1] as BASELINE.json states it (1080p, 6 levels, 500 iterations per level, reference
    stopping rule).  The optimizer is chaotic: accept/reject decisions flip on the last bit and a
    flip on the 120x68 level is worth 16 px five levels up.  Its intrinsic reproducibility is
    MEASURED here, PER FRAME, as the spread over a family of six equally legal runs of the reference
    algorithm (CHAOS_FAMILY: four commit orders in EXACT arithmetic -- the first is the oracle's,
    bit for bit -- and two of them with fused multiply-adds, as nvcc's default --fmad=true compiles
    the reference), over six frames, two of them frames whose finest level keeps cycling in FAST.
    FAST (the production arithmetic, ~ the reference's --use_fast_math) is judged, per frame,
    against SURVEY 8(d)'s fixed bounds or 1.25 x that frame's own range, whichever is larger:
      RMS dv(FAST, oracle's run)     <= max(0.05 px, 1.25 x max pairwise RMS dv inside the family)
      |E_FAST - E_oracle| / E_oracle <= max(0.5 %,   1.25 x (max E - min E) / E over the family)
      pixels within 0.25 px          >= the smallest such fraction between two family members - 0.03
    and the SIGNED energy deviation (E_FAST - mean E_family) / mean E_family over the frames must not be
    significantly above zero (a systematically higher final energy would be a quality loss, not chaos):
    mean <= 2 standard errors."""
    res = chaos_floor_measure(gpu_ctx, keep_fields_of=CHAOS_FRAMES[0])
    keep = res[CHAOS_FRAMES[0]].pop("fields")
    table = {f: chaos_round(r) for f, r in res.items()}
    print("chaos floor, per frame:", table)
    for f, r in res.items():
        msg = (f, table[f])
        assert r["rms_floor"] > 0.01, msg                  # the legal orders do diverge at this size
        assert r["rms_fast0"] <= max(0.05, 1.25 * r["rms_floor"]), msg
        assert r["e_fast0"] <= max(0.005, 1.25 * r["e_floor"]), msg
        assert r["within_fast0"] >= r["within_floor"] - 0.03, msg
    signed = np.array([r["e_fast_signed"] for r in res.values()])
    sem = signed.std(ddof=1) / np.sqrt(len(signed))
    assert signed.mean() <= 2.0 * sem, (signed.tolist(), float(signed.mean()), float(sem))
    # Round 5 -- the reference BINARY's texture arithmetic (TEX8 members: CUDA's 8-bit bilinear weights, eps = 2.56
    # weight quanta) does NOT lie inside that family: measured 1.75-1.93 px RMS from every exact-weight member (ten
    # times the family's range), with an energy (oracle, exact weights) 54-63 % above theirs -- the quantised
    # landscape stalls the descent on the small levels (the 240x135 level stops after 17-40 sweeps instead of
    # 100-216).  So no exact-weight build, the oracle of SURVEY appendix A included, can be held to 0.05 px -- or to
    # the family's 0.19 px -- of the reference binary; what CAN be stated and is asserted: (i) the gap is there and it
    # is large, (ii) FAST is no further from the TEX8 runs than the exact-weight legal builds are, (iii) the TEX8 runs
    # never reach a lower energy than the exact-weight family.
    for f, r in res.items():
        msg = (f, table[f])
        lo, hi = min(r["rms_tex8_vs_exact"]), max(r["rms_tex8_vs_exact"])
        assert lo > 3.0 * r["rms_floor"], msg
        assert max(r["rms_fast_vs_tex8"]) <= 1.1 * hi and min(r["rms_fast_vs_tex8"]) >= 0.9 * lo, msg
        assert min(r["E_tex8"]) > max(r["E_family"]), msg
    # the rendered halfway frame from either field: >= 99 % of the bytes within 2 levels
    w, h = 1920, 1080
    ex = int(0.1 * max(w, h))
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    fr = morph.Frame(gpu_ctx, w, h, ex)
    frames = []
    for v in keep:
        fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), v, None)
        frames.append(fr.render_halfway(0.5, 0.5, 1).astype(np.int32))
    dpx = np.abs(frames[0] - frames[1])
    assert (dpx <= 2).mean() >= 0.98 and dpx.mean() < 0.6, ((dpx <= 2).mean(), dpx.mean())


def test_fast_sits_inside_the_family_at_config3(gpu_ctx):
    """The same per-frame comparison at config[3] (3840x2160, 7 levels, 500 iterations per level): one more level
    between the never-converging small levels and the result doubles what separates two legal builds (measured,
    r04: family range 0.35-0.42 px RMS, 65-66 % of the pixels within 0.25 px; energies of the members up to 2x
    apart) -- and FAST lies inside that range (0.27-0.35 px, 71-74 % within 0.25 px, energy not above the
    family's).  Two frames."""
    res = chaos_floor_measure(gpu_ctx, frames=(0, 3), w=3840, h=2160)
    for f, r in res.items():
        msg = (f, chaos_round(r))
        # the TEX8 members (round 5): 3.7-3.95 px RMS from every exact-weight member at this size, FAST among the latter
        assert min(r["rms_tex8_vs_exact"]) > 3.0 * r["rms_floor"], msg
        assert max(r["rms_fast_vs_tex8"]) <= 1.1 * max(r["rms_tex8_vs_exact"]), msg
        assert r["rms_floor"] > 0.05, msg
        assert r["rms_fast0"] <= 1.25 * r["rms_floor"], msg
        assert r["within_fast0"] >= r["within_floor"] - 0.03, msg
        assert r["e_fast0"] <= max(0.005, 1.25 * r["e_floor"]), msg
        assert r["E_fast"] <= 1.05 * max(r["E_family"]), msg


def test_fast_does_not_drift_on_the_never_converging_level(gpu_ctx):
    """Regression test of round 4's finding.  The 120x68 level of config[1] never converges within its 500
    iterations, so the final field of a solve is wherever that slow descent stands -- and whatever separates two
    arithmetics there is multiplied by 16 on the way to full resolution.  With pre-divided window means updated
    by fma(d, 1/n, mean) FAST's line search predicted energies the level did not have after the commit, and ~50
    pixels at the left border crept 0.2-0.37 px away from every legal build of the reference's formula (63-82 % of
    the squared distance in those pixels, uphill in the oracle's energy).  From the SAME start, 250 iterations:
    FAST's distance from EXACT must be that of the legal builds (another commit order, REF_FASTMATH), no pixel may
    be further off than 2.5 x their worst one, and FAST's energy (oracle) must sit where theirs does."""
    import oracle as O
    w, h = 1920, 1080
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    i0, i1 = synth.make_pair(w, h, frame=15)
    P = O.default_params()
    out, imgs = {}, None
    try:
        for name, mode, order in (("exact", capi.MATH_EXACT, 0), ("order2", capi.MATH_EXACT, 2), ("reffm", capi.MATH_REF_FASTMATH, 0),
                                  ("fast", capi.MATH_FAST, 0)):
            gpu_ctx.set_math_mode(capi.MATH_EXACT)
            gpu_ctx.set_commit_order(0)
            p = morph.Pyramid(gpu_ctx)
            p.build(i0, i1, 32)
            L, nl = p._L, p.size() - 1
            el = nl - 1
            capi.check(L.vm_coarse_solve(p._h, nl - 1, w, h, None, 0))
            capi.check(L.vm_upsample_v(p._h, el - 1, el))
            gpu_ctx.set_math_mode(mode)
            gpu_ctx.set_commit_order(order)
            capi.check(L.vm_init_level(p._h, el - 1, w, h, None, 0))
            capi.check(L.vm_optimize_level(p._h, el - 1, 250.0, None, 1, None))
            out[name] = p[el].v
            imgs = (p[el].field("img0"), p[el].field("img1"))
            p.clear()
    finally:
        gpu_ctx.set_commit_order(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)

    def energy(v):
        lv = O.Level(v.shape[1], v.shape[0])
        lv.set_images(*imgs)
        lv.field("v")[...] = v
        lv.init(0.0)
        e = lv.energy(P)
        return float(P.w_ssim * e[0] / (v.shape[0] * v.shape[1]) + P.w_tps * e[1])
    dist = {k: np.sqrt(((out[k] - out["exact"]) ** 2).sum(-1)) for k in ("order2", "reffm", "fast")}
    rms = {k: float(np.sqrt((d ** 2).mean())) for k, d in dist.items()}
    E = {k: energy(v) for k, v in out.items()}
    msg = dict(rms=rms, max={k: float(d.max()) for k, d in dist.items()}, dE={k: (E[k] - E["exact"]) / E["exact"] for k in dist})
    legal_rms, legal_max = max(rms["order2"], rms["reffm"]), max(dist["order2"].max(), dist["reffm"].max())
    assert 0.003 < legal_rms < 0.02, msg                            # measured 0.0068-0.0073 level pixels
    assert rms["fast"] <= 1.3 * legal_rms, msg                      # 0.0070 after the fix, 0.0175 before
    assert dist["fast"].max() <= 2.5 * legal_max, msg               # 0.037 after, 0.365 before (legal: 0.030-0.037)
    assert (E["fast"] - E["exact"]) / E["exact"] <= 0.015, msg      # +0.12 % after, +3.0 % before (legal: +0.2 .. +0.7 %)


def _window_sum_invariants(lv, i0, i1):
    """mean/var/cross are the 5x5 box sums of the stored lumas; lumas are the images at p -/+ v"""
    h, w = i0.shape
    v, luma = lv.v, lv.field("luma")
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    assert np.abs(_bilinear(i0, xx - v[..., 0], yy - v[..., 1]) - luma[..., 0]).max() < 2e-2
    assert np.abs(_bilinear(i1, xx + v[..., 0], yy + v[..., 1]) - luma[..., 1]).max() < 2e-2
    mean, var, cross = lv.field("mean"), lv.field("var"), lv.field("cross")
    assert np.abs(_box5(luma) - mean).max() < 0.05
    assert np.abs(_box5(luma.astype(np.float64) ** 2) - var).max() < 8.0
    assert np.abs(_box5(luma[..., 0].astype(np.float64) * luma[..., 1]) - cross).max() < 8.0
    tb = lv.field("tps_b")[2:-2, 2:-2]
    assert np.abs(_tps_apply(v) - tb).max() < 5e-3 * max(1.0, np.abs(tb).max())


@pytest.mark.parametrize("mode,sched", [(capi.MATH_FAST, capi.SWEEP_TILE), (capi.MATH_EXACT, capi.SWEEP_AUTO)])
def test_config2_batch_of_8_pairs_at_1080p_equals_individual_solves(gpu_ctx, mode, sched):
    """config[2]'s per-GPU unit at full size: 8 frame pairs of 1920x1080 (6 levels), solved by ONE
    vm_solve_batch (all pairs relaxed by the same launches, reference semantics per pair), end
    bit-identical to the 8 pairs solved one at a time, with the same per-level iteration counts;
    the state of a batched pair satisfies the window-sum invariants.  FAST is compared under one
    fixed schedule (its schedules differ in summation order); EXACT under the automatic choice --
    which picks different schedules for a batch and for a single pair -- because every EXACT
    schedule is bit-identical to the oracle."""
    gpu_ctx.set_math_mode(mode)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h, sweeps = 1920, 1080, 40
    frames = [synth.make_pair(w, h, frame=k) for k in range(4)]
    try:
        gpu_ctx.set_tuning(sched, 0, 0)
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor, prm.start_res = sweeps, 1.0, 32
        single, iters = [], []
        for k in range(4):                  # pairs 4..7 repeat frames 0..3: solved once
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(frames[k][0], frames[k][1], 32)
            m = morph.Morph(prm, pyr)
            m.calculate_halfway_parametrization()
            single.append(pyr[1].v)
            iters.append([m.progress[el]["iters"] for el in sorted(m.progress)])
            del pyr
        batch = []
        for k in range(8):
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build(frames[k % 4][0], frames[k % 4][1], 32)
            batch.append(pyr)
        assert batch[0].size() == 7
        prog = morph.solve_batch(batch, sweeps, 1.0)
        for k in range(8):
            assert [p["iters"] for p in prog[k]] == iters[k % 4], k
            assert np.array_equal(single[k % 4].view(np.uint32), batch[k][1].v.view(np.uint32)), k
        assert np.abs(single[0]).max() > 5.0          # the solve moved: ~19 px of displacement at 1080p
        _window_sum_invariants(batch[5][1], *frames[1])
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


@pytest.mark.parametrize("npairs,lds_cap", [(1, 0), (3, 0), (1, 1), (1, 24), (3, 40)])
def test_sparse_schedule_equals_tile_at_1080p(gpu_ctx, npairs, lds_cap):
    """config[1] geometry, FAST, 60 sweeps per level, fixed work: the SPARSE schedule (pruned levels
    swept by one workgroup per pair, 1-2 launches per batch of iterations) ends bit-identical to
    the TILE schedule (4 launches per iteration), alone and as a batch whose pairs advance
    independently inside the sparse kernel; and it needs fewer launches.  lds_cap: the capacity of the
    kernel's LDS-resident word list (0: the built-in 1024) -- 1 keeps the list in memory from the start, 24 / 40
    make it outgrow LDS in mid-run (rescan of the level into the lists in memory, then on from there)."""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    w, h = 1920, 1080
    frames = [synth.make_pair(w, h, frame=k) for k in range(npairs)]
    out = {}
    try:
        for sched in (capi.SWEEP_TILE, capi.SWEEP_SPARSE):
            gpu_ctx.set_tuning(sched, 0, lds_cap if sched == capi.SWEEP_SPARSE else 0)
            batch = []
            for k in range(npairs):
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build(frames[k][0], frames[k][1], 32)
                batch.append(pyr)
            if npairs == 1:
                prm = morph.Parameters()
                prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 60, 1.0, 32
                m = morph.Morph(prm, batch[0], fixed_work=True)
                m.calculate_halfway_parametrization()
                launches = sum(m.progress[el]["launches"] for el in m.progress)
            else:
                prog = morph.solve_batch(batch, 60, 1.0, fixed_work=True)
                launches = sum(p["launches"] for p in prog[0])
            out[sched] = ([b[1].v for b in batch], [b[1].field("impmask") for b in batch], launches)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    for k in range(npairs):
        assert np.array_equal(out[capi.SWEEP_TILE][0][k].view(np.uint32), out[capi.SWEEP_SPARSE][0][k].view(np.uint32)), k
        assert np.array_equal(out[capi.SWEEP_TILE][1][k], out[capi.SWEEP_SPARSE][1][k]), k
    assert out[capi.SWEEP_SPARSE][2] < out[capi.SWEEP_TILE][2], (out[capi.SWEEP_TILE][2], out[capi.SWEEP_SPARSE][2])


@pytest.mark.parametrize("w,h,nlev,frames", [(1920, 1080, 6, (6, 9)), (3840, 2160, 7, (0, 2))])
def test_resident_sparse_visits_equal_list_driven_ones_on_a_cycling_level(gpu_ctx, w, h, nlev, frames):
    """config[1] and config[3], FAST, the automatic schedule, the reference's stopping rule; at 1080p on frames whose
    finest level keeps exchanging rounding-level moves on its bottom border row until iteration 500 (frames 6 and 9
    of the synthetic video; profiles/r04_notes.md): the sparse kernel keeps the window sums around those pixels in LDS
    across passes and iterations (resident visits, vm_dbg_sparse_resident).  Every state array of the finest level,
    the per-level iteration counts and the activity counters (tile visits, candidates, commits, evaluations) are the
    bits of list-driven visits (mode 1) -- automatically (0), with the LDS copy re-centred after every commit (2), and
    with residency given up at the first commit of every batch (3: the rest of the pass as plain visits, the list
    rebuilt by a scan of the level)."""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
    gpu_ctx.set_params(morph.KernParameters(prm))
    cycling = 0
    try:
        for frame in frames:
            i0, i1 = synth.make_pair(w, h, frame=frame)
            ref = None
            for mode in (1, 0, 2, 3):
                gpu_ctx.set_sparse_resident(mode)
                pyr = morph.Pyramid(gpu_ctx)
                pyr.build(i0, i1, 32)
                assert pyr.size() == nlev + 1
                prog = (capi.Progress * (nlev - 1))()
                served = gpu_ctx.sparse_resident_visits()
                capi.check(pyr._L.vm_solve(pyr._h, 500.0, 1.0, None, 0, None, 0, prog))
                served = gpu_ctx.sparse_resident_visits() - served
                # never: none; automatic on a cycling level: most of its ~2000 visits
                if mode == 1:
                    assert served == 0, (frame, served)
                elif prog[0].iters == 500:
                    assert served > (1000 if mode == 0 else 0), (frame, mode, served)
                got = ([pyr[1].field(n).copy() for n in _STATE],
                       [(p.iters, p.improving, p.commits, p.candidates, p.evaluations, p.active_tiles) for p in prog])
                del pyr
                if ref is None:
                    ref = got
                    cycling += prog[0].iters == 500
                    continue
                assert got[1] == ref[1], (frame, mode, got[1], ref[1])
                for n, a, b in zip(_STATE, ref[0], got[0]):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (frame, mode, n)
    finally:
        gpu_ctx.set_sparse_resident(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    assert cycling >= 1 or w != 1920, cycling        # at 1080p at least one of the two frames does cycle


@pytest.mark.parametrize("w,h", [(120, 68), (240, 135), (150, 97)])
def test_dense_sweeps_do_not_depend_on_the_workgroup_size(gpu_ctx, w, h):
    """FAST, dense TILE sweeps of a small level from a rough start (every pixel searched for the first sweeps) with
    256- and with 512-thread workgroups: the same bits in every state array and the same counters -- the lane
    fan-out per candidate, which orders the FAST sums, is a constant of the kernel; a 256-thread workgroup takes
    a full phase in two rounds.  (The library picks 256 for such levels when enough of their workgroups are in
    flight on the device to pair up on the CUs: vm_api.cpp, SmallDensePresence.)"""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
    i0, i1 = synth.make_pair(w, h)
    v0 = (0.9 * synth.displacement(w, h)).astype(np.float32)
    out = []
    try:
        for threads in (512, 256):
            gpu_ctx.set_tuning(capi.SWEEP_TILE, threads, 0)
            pyr = morph.Pyramid(gpu_ctx)
            pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
            pyr.upload_luma(1, i0, i1)
            pyr[1].v = v0
            capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
            pr = capi.Progress()
            capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 24.0, None, 1, C.byref(pr)))
            out.append(([pyr[1].field(n).copy() for n in _STATE], (pr.iters, pr.commits, pr.candidates, pr.evaluations, pr.active_tiles)))
            del pyr
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    assert out[0][1] == out[1][1] and out[0][1][1] > 1000, (out[0][1], out[1][1])
    for n, a, b in zip(_STATE, out[0][0], out[1][0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), n


_FORM_SCRIPT = r"""
import sys, ctypes as C
import numpy as np
sys.path.insert(0, %r)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
ctx.set_tuning(capi.SWEEP_TILE, 0, 0)
w, h = 150, 97
i0, i1 = synth.make_pair(w, h)
pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
pyr.upload_luma(1, i0, i1)
pyr[1].v = (0.9 * synth.displacement(w, h)).astype(np.float32)
capi.check(pyr._L.vm_init_level(pyr._h, 0, w, h, None, 0))
pr = capi.Progress()
capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(sys.argv[1]), None, 1, C.byref(pr)))
np.save(sys.argv[2], pyr[1].v)
print(pr.commits, pr.candidates)
"""


def test_small_level_dense_kernel_against_the_general_one(tmp_path):
    """FAST, dense TILE sweeps: the kernel form of the small levels (VM_DENSE_NOINT=1 forces it, =0
    forbids it; in processes of their own, the switch is read once) has no interior form of the line
    search -- the border form computes at run time the window counts the interior form has as
    constants: the same bits -- and no lean bodies: a phase of <= 16 candidates takes the two-lane
    dense search, whose sums are ordered differently.  One sweep from a fresh level (every phase
    full): bit-identical; 12 sweeps: the FAST rounding band (RMS <= 0.01 px, counters within 2 %)."""
    import subprocess
    import sys
    res = {}
    for iters in (1, 12):
        for form in ("0", "1"):
            env = dict(os.environ, VM_DENSE_NOINT=form, VM_TILE_DENSE="1")
            out = str(tmp_path / ("v_%s_%d.npy" % (form, iters)))
            r = subprocess.run([sys.executable, "-c", _FORM_SCRIPT % ROOT, str(iters), out], capture_output=True, text=True,
                               timeout=300, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            res[(form, iters)] = (np.load(out), [int(float(x)) for x in r.stdout.split()[-2:]])
    a, b = res[("0", 1)], res[("1", 1)]
    assert a[1] == b[1] and a[1][0] > 1000, (a[1], b[1])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    a, b = res[("0", 12)], res[("1", 12)]
    rms = float(np.sqrt(np.mean((a[0] - b[0]) ** 2)))
    assert rms <= 0.01, rms
    assert abs(a[1][0] - b[1][0]) <= 0.02 * a[1][0] and abs(a[1][1] - b[1][1]) <= 0.02 * a[1][1], (a[1], b[1])
