"""Randomised parity: level sizes, boundary conditions, constraint counts and iteration counts
drawn from a seeded generator (ragged sizes, single-tile levels, tiles cut by the border)."""
import ctypes as C

import numpy as np
import pytest

from videomorphing_amd import capi, morph, synth
import test_gpu_parity as T

pytestmark = pytest.mark.gpu

STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")


def _draw(rng, wmax, hmax):
    """size, kernel parameters (boundary condition, weights over three decades, clamp, eps), constraints"""
    w, h = int(rng.randint(10, wmax)), int(rng.randint(10, hmax))
    ncons = int(rng.randint(0, 4))
    cons = synth.make_constraints(w, h, ncons) if ncons and min(w, h) > 40 else ()
    kw = dict(bcond=int(rng.randint(0, 3)), w_tps=float(10 ** rng.uniform(-3, 0)), w_ssim=float(10 ** rng.uniform(0, 3)),
              w_ui=float(10 ** rng.uniform(3, 6)), ssim_clamp=float(rng.choice([0.0, 0.0, 0.3])),
              eps=float(rng.choice([0.01, 0.01, 0.003, 0.03])))
    return w, h, kw, cons


@pytest.mark.parametrize("seed", [11, 12])
def test_exact_sweeps_equal_the_oracle_on_random_levels(gpu_ctx, oracle, seed):
    """EXACT, TILE / SPLIT / STEP / SPARSE / PASS schedules, 1-3 sweeps: every state array bit-identical to the oracle"""
    rng = np.random.RandomState(seed)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    try:
        for trial in range(8):
            w, h, kw, cons = _draw(rng, 300, 120)
            iters = int(rng.randint(1, 4))
            for sched in (capi.SWEEP_TILE, capi.SWEEP_SPLIT, capi.SWEEP_STEP, capi.SWEEP_SPARSE, capi.SWEEP_PASS):
                P = T._params(oracle, **kw)
                lo, pyr, P = T._make_level(gpu_ctx, oracle, w, h, cons=cons, P=P, seed=trial)
                for _ in range(iters):
                    lo.optimize_iter(P)
                gpu_ctx.set_tuning(sched, 0, 0)
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(iters), None, 1, None))
                T._assert_state_equal(lo, pyr[1])
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)


@pytest.mark.parametrize("seed", [21, 22])
def test_fast_step_equals_split_on_random_levels(gpu_ctx, oracle, seed):
    """FAST: the one-launch-per-phase STEP schedule and the one-launch-per-pass PASS schedule (tile-local
    barriers between the phases) against the two-kernel SPLIT schedule, bitwise, incl. the counters"""
    rng = np.random.RandomState(seed)
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    try:
        for trial in range(12):
            w, h, kw, cons = _draw(rng, 420, 160)
            iters = float(rng.randint(1, 7))
            res = []
            for sched in (capi.SWEEP_SPLIT, capi.SWEEP_STEP, capi.SWEEP_PASS):
                P = T._params(oracle, **kw)
                lo, pyr, P = T._make_level(gpu_ctx, oracle, w, h, cons=cons, P=P, seed=trial)
                gpu_ctx.set_tuning(sched, 0, 0)
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, 1, C.byref(pr)))
                res.append(([pyr[1].field(n).copy() for n in STATE], (pr.commits, pr.candidates, pr.evaluations)))
            assert res[0][1] == res[1][1] == res[2][1], (w, h, kw, iters, res[0][1], res[1][1], res[2][1])
            for k in (1, 2):
                for n, a, b in zip(STATE, res[0][0], res[k][0]):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, n, w, h, kw, iters)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)


def _pruned_level(gpu_ctx, oracle, rng, w, h, kw, cons, trial, sweeps=80.0):
    """a level in the pruned regime: the usual random level after 80 sweeps of the TILE schedule
    (most of its pixels have stopped moving; the continuation is what is compared)"""
    P = T._params(oracle, **kw)
    lo, pyr, P = T._make_level(gpu_ctx, oracle, w, h, cons=cons, P=P, seed=trial)
    gpu_ctx.set_tuning(capi.SWEEP_TILE, 0, 0)
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, float(sweeps), None, 1, None))
    return pyr


@pytest.mark.parametrize("mode,seed", [(capi.MATH_FAST, 31), (capi.MATH_FAST, 32), (capi.MATH_EXACT, 33)])
def test_sparse_schedule_equals_tile_on_random_levels(gpu_ctx, oracle, mode, seed):
    """SPARSE (one workgroup per pair walks the active tiles of a pruned level, a whole batch of
    iterations per launch) against TILE (one launch per pass): bit-identical state, iteration
    counts and activity counters over 10-60 iterations, fixed work and reference stopping rule"""
    rng = np.random.RandomState(seed)
    gpu_ctx.set_math_mode(mode)
    used = 0
    try:
        for trial in range(10):
            w, h, kw, cons = _draw(rng, 420, 200)
            w, h = max(w, 40), max(h, 40)
            if trial % 3:      # the reference's weights (the drawn ones often keep every pixel active)
                kw = dict(bcond=kw["bcond"], eps=kw["eps"], ssim_clamp=kw["ssim_clamp"])
            iters = float(rng.randint(10, 60))
            fixed = int(rng.randint(0, 2))
            st = rng.get_state()
            res = []
            for sched in (capi.SWEEP_TILE, capi.SWEEP_SPARSE):
                rng.set_state(st)
                pyr = _pruned_level(gpu_ctx, oracle, rng, w, h, kw, cons, trial)
                gpu_ctx.set_tuning(sched, 0, 0)
                # (FAST, the lean kernel's resident visits: automatic; the LDS copy re-centred after every commit;
                # residency given up at the first commit)
                gpu_ctx.set_sparse_resident((0, 2, 3)[trial % 3])
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, fixed, C.byref(pr)))
                res.append(([pyr[1].field(n).copy() for n in STATE], (pr.iters, pr.improving, pr.commits, pr.candidates, pr.evaluations),
                            pr.sched_launches[3], pr.active_tiles))
            assert res[0][1] == res[1][1], (w, h, kw, iters, fixed, res[0][1], res[1][1])
            # (the count of "active" tiles: exact since round 4 -- a tile is active iff a set bit lies within +-2 of
            # it, bits of its own pixels or of the gaps around it, which no other tile of the pass touches; rounds
            # 1-3 tested whole words of the window, words a neighbouring tile of the same pass may be updating)
            assert res[0][3] == res[1][3], (w, h, kw, iters, fixed, res[0][3], res[1][3])
            for n, a, b in zip(STATE, res[0][0], res[1][0]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n, w, h, kw, iters, fixed)
            assert res[0][2] == 0
            used += res[1][2] > 0
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_sparse_resident(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    assert used >= 2, used          # the sparse kernel really ran (the 1080p tests exercise it at length)


def _plant_clusters(rng, w, h):
    """an improving mask of a few small clusters of set bits: 1-3 clusters (mostly one; a further one mostly close to the
    first: within a tile's reach of it) of 1-6 bits inside a box of <= 12 x 3 pixels, placed anywhere -- with a bias
    towards the image border, its corners and the 5-pixel gaps between the tiles of a pass (x = 64..68 mod 69,
    y = 16..20 mod 21), where a cluster belongs to two tiles of a pass or to none"""
    rows, rs = (h + 4) // 5 + 2, (w + 4) // 5 + 2
    words = np.zeros((rows, rs), np.uint32)
    first = None
    for k in range(int(rng.choice([1, 1, 1, 2, 2, 3]))):
        bw, bh = int(rng.randint(1, 13)), int(rng.randint(1, 4))
        kind = int(rng.randint(0, 4))
        if first is not None and rng.randint(0, 4):   # close to the first cluster
            x0, y0 = first[0] + int(rng.randint(-45, 46)), first[1] + int(rng.randint(-10, 11))
        elif kind == 0:     # anywhere
            x0, y0 = int(rng.randint(0, w)), int(rng.randint(0, h))
        elif kind == 1:     # on a border / in a corner
            x0 = int(rng.choice([0, w - bw, rng.randint(0, w)]))
            y0 = int(rng.choice([0, h - bh, rng.randint(0, h)]))
        else:               # across a gap between tiles
            x0 = 69 * int(rng.randint(0, max(1, w // 69 + 1))) + int(rng.randint(58, 70))
            y0 = 21 * int(rng.randint(0, max(1, h // 21 + 1))) + int(rng.randint(12, 22))
        x0, y0 = min(max(x0, 0), w - 1), min(max(y0, 0), h - 1)
        if first is None:
            first = (x0, y0)
        for _ in range(int(rng.randint(1, 7))):
            x, y = min(x0 + int(rng.randint(0, bw)), w - 1), min(y0 + int(rng.randint(0, bh)), h - 1)
            words[y // 5 + 1, x // 5 + 1] |= np.uint32(1 << (x % 5 + 5 * (y % 5)))
    return words


def test_resident_sparse_visits_on_planted_clusters(gpu_ctx, oracle):
    """FAST, SPARSE against TILE from states whose improving mask is a few planted clusters of set bits (anywhere, on the
    borders, across the gaps between tiles) on a level that has not converged -- the moves they trigger spread: what the
    lean sparse kernel serves from its resident LDS copy, re-centres, or gives up (vm_dbg_sparse_resident 0 / 2 / 3).
    Bit-identical state, iteration counts and activity counters; and the resident copy did serve visits (three seeds x 20
    levels: most planted clusters die at once or outgrow a tile within a few sweeps, a few live on for dozens)."""
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    served0 = gpu_ctx.sparse_resident_visits()
    try:
        for seed in (51, 52, 53):
            rng = np.random.RandomState(seed)
            for trial in range(20):
                w, h, kw, cons = _draw(rng, 420, 200)
                w, h = max(w, 40), max(h, 40)
                if trial % 3:
                    kw = dict(bcond=kw["bcond"], eps=kw["eps"], ssim_clamp=kw["ssim_clamp"])
                iters = float(rng.randint(6, 50))
                fixed = int(rng.randint(0, 2))
                words = _plant_clusters(rng, w, h)
                # how settled the level is when the clusters are planted: after 80 sweeps a planted bit mostly dies at once;
                # after 12-50 its neighbourhood still moves and the active region grows out of the LDS copy
                sweeps = float(rng.choice([12, 25, 50, 80]))
                st = rng.get_state()
                res = []
                for sched in (capi.SWEEP_TILE, capi.SWEEP_SPARSE):
                    rng.set_state(st)
                    pyr = _pruned_level(gpu_ctx, oracle, rng, w, h, kw, cons, trial, sweeps)
                    pyr[1].set_impmask(words)
                    gpu_ctx.set_tuning(sched, 0, 0)
                    gpu_ctx.set_sparse_resident((0, 0, 2, 3)[trial % 4])
                    pr = capi.Progress()
                    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, fixed, C.byref(pr)))
                    res.append(([pyr[1].field(n).copy() for n in STATE],
                                (pr.iters, pr.improving, pr.commits, pr.candidates, pr.evaluations, pr.active_tiles)))
                assert res[0][1] == res[1][1], (trial, w, h, kw, iters, fixed, res[0][1], res[1][1])
                for n, a, b in zip(STATE, res[0][0], res[1][0]):
                    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (trial, n, w, h, kw, iters, fixed)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_sparse_resident(0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
    served = gpu_ctx.sparse_resident_visits() - served0
    print("tile visits served from the resident copy:", served)
    assert served > 100, served


@pytest.mark.parametrize("seed", [61, 62])
def test_listed_tile_passes_equal_direct_ones(gpu_ctx, oracle, seed):
    """FAST, the TILE schedule on pruned levels: a pass launched as tiles x pairs workgroups (direct) against the listed
    form big batches take -- a scan lists the tiles of the pass a set mask bit reaches, a fixed grid of workgroups walks the
    list (vm_set_tuning(VM_SWEEP_TILE, 0, parts = 1) lowers the threshold to one workgroup).  Bit-identical state,
    iteration counts and activity counters, fixed work and reference stopping rule, graph replays included (>= 8
    iterations)."""
    rng = np.random.RandomState(seed)
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    try:
        for trial in range(8):
            w, h, kw, cons = _draw(rng, 420, 200)
            w, h = max(w, 40), max(h, 40)
            if trial % 3:
                kw = dict(bcond=kw["bcond"], eps=kw["eps"], ssim_clamp=kw["ssim_clamp"])
            iters = float(rng.randint(4, 70))
            fixed = int(rng.randint(0, 2))
            st = rng.get_state()
            res = []
            for listed in (0, 1):
                rng.set_state(st)
                pyr = _pruned_level(gpu_ctx, oracle, rng, w, h, kw, cons, trial, 40.0)
                gpu_ctx.set_tuning(capi.SWEEP_TILE, 0, listed)
                pr = capi.Progress()
                capi.check(pyr._L.vm_optimize_level(pyr._h, 0, iters, None, fixed, C.byref(pr)))
                res.append(([pyr[1].field(n).copy() for n in STATE],
                            (pr.iters, pr.improving, pr.commits, pr.candidates, pr.evaluations, pr.active_tiles)))
            assert res[0][1] == res[1][1], (trial, w, h, kw, iters, fixed, res[0][1], res[1][1])
            for n, a, b in zip(STATE, res[0][0], res[1][0]):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (trial, n, w, h, kw, iters, fixed)
    finally:
        gpu_ctx.set_tuning(capi.SWEEP_AUTO, 0, 0)
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
