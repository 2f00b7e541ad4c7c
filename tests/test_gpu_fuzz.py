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


def _pruned_level(gpu_ctx, oracle, rng, w, h, kw, cons, trial):
    """a level in the pruned regime: the usual random level after 80 sweeps of the TILE schedule
    (most of its pixels have stopped moving; the continuation is what is compared)"""
    P = T._params(oracle, **kw)
    lo, pyr, P = T._make_level(gpu_ctx, oracle, w, h, cons=cons, P=P, seed=trial)
    gpu_ctx.set_tuning(capi.SWEEP_TILE, 0, 0)
    capi.check(pyr._L.vm_optimize_level(pyr._h, 0, 80.0, None, 1, None))
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
