"""Host-side logic of the mirror API and the bench accounting (CPU only)."""
import os

import numpy as np

import bench
from videomorphing_amd import capi, dist, morph, synth


def test_level_geometry_tables_of_the_survey():
    # level geometry of the BASELINE configs
    def sizes(w, h, start_res):
        n = synth.num_levels(w, h, start_res)
        out = [(w, h)]
        for _ in range(n - 1):
            w, h = (w + 1) // 2, (h + 1) // 2
            out.append((w, h))
        return out
    assert sizes(256, 256, 64) == [(256, 256), (128, 128), (64, 64)]
    assert sizes(1920, 1080, 32) == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 68), (60, 34)]
    assert sizes(3840, 2160, 32) == [(3840, 2160), (1920, 1080), (960, 540), (480, 270), (240, 135),
                                     (120, 68), (60, 34)]
    i0, i1 = synth.make_pair(97, 61)
    pyr = synth.build_pyramid(i0, i1, 3)
    assert [p[0].shape for p in pyr] == [(61, 97), (31, 49), (16, 25)]


def test_synthetic_inputs_are_deterministic_and_in_range():
    a0, a1 = synth.make_pair(80, 48, frame=2)
    b0, b1 = synth.make_pair(80, 48, frame=2)
    assert np.array_equal(a0, b0) and np.array_equal(a1, b1)
    assert a0.dtype == np.float32 and 16 <= a0.min() and a0.max() <= 240 and a0.std() > 20
    c0, _ = synth.make_pair(80, 48, frame=3)
    assert not np.array_equal(a0, c0)
    d = synth.displacement(1920, 1080)
    assert abs(np.abs(d).max() - 19.2) < 0.05           # A = 0.01 W
    cons = synth.make_constraints(1920, 1080, 8)
    assert cons.shape == (8, 5) and np.all(cons[:, 4] == 1) and np.all(cons[:, :4] == np.rint(cons[:, :4]))


def test_parameters_mirror_defaults_and_constraint_resolution():
    P = morph.Parameters()          # UI/MdiEditor.cpp:131-140
    assert (P.w_ssim, P.ssim_clamp, P.w_tps, P.w_ui, P.w_temp) == (100.0, 0.0, 0.05, 1e5, 10.0)
    assert (P.max_iter, P.max_iter_drop_factor, P.eps, P.start_res, P.bcond) == (1000, 2.0, 0.01, 8, 0)
    kp = morph.KernParameters(P)
    assert [getattr(kp, f) for f, _ in capi.KernParams._fields_][:6] == [
        10.0, np.float32(1e5), np.float32(0.05), 100.0, 0.0, np.float32(0.01)]
    # tracks with one point per frame; only connections whose left point sits on page 0 count
    P.lp = [[morph.Conp(10, 20, 0, 1, 1.0), morph.Conp(11, 21, 1, 0, 0.4)]]
    P.rp = [[morph.Conp(14, 22, 0, 1, 0.6), morph.Conp(15, 23, 1, 0, 0.9)]]
    P.cnt = [[morph.Connect((0, 0), (0, 0)), morph.Connect((0, 1), (0, 1))]]
    c = P.constraints(0)
    assert c.shape == (1, 5) and c[0].tolist() == [10, 20, 14, 22, np.float32(0.6)]
    assert P.constraints(1)[0].tolist() == [11, 21, 15, 23, np.float32(0.4)]
    P2 = morph.Parameters()
    P2.add_point_pair(1, 2, 3, 4, 0.5)
    assert P2.constraints(0).tolist() == [[1, 2, 3, 4, 0.5]]


class _FakeLevel(object):
    def __init__(self, w, h):
        self.width, self.height, self.depth = w, h, 1


class _FakePyramid(object):
    def __init__(self, sizes):
        self._lv = [_FakeLevel(*sizes[0])] + [_FakeLevel(*s) for s in sizes]

    def size(self):
        return len(self._lv)

    def __getitem__(self, i):
        return self._lv[i]


def test_morph_progress_accounting_follows_the_reference_ctor():
    """Morph ctor, morph.cu:122-141: _total_iter sums iter_num*W*H over the optimised
    levels with iter_num divided by the drop factor from coarse to fine"""
    P = morph.Parameters()
    P.max_iter, P.max_iter_drop_factor = 1000, 2.0
    pyr = _FakePyramid([(256, 256), (128, 128), (64, 64)])
    m = morph.Morph(P, pyr)
    assert m._total_l == 3 and m._current_l == 3 and m._max_iter == 1000.0
    assert m._total_iter == 1000 * 128 * 128 + 500 * 256 * 256


def test_make_extended_canvas():
    rgb = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    can = morph.make_extended(rgb, 2)      # pyramid.cu:186-200
    assert can.shape == (6, 7, 4) and can[0, 0].tolist() == [255, 255, 255, 255]
    assert np.array_equal(can[2:4, 2:5, :3], rgb) and can[2:4, 2:5, 3].max() == 0


def test_bench_algorithmic_byte_accounting():
    # visits per pixel per iteration -> 4*64*16/(69*21) = 2.827 for large levels
    v = bench.tile_visits(1920, 1080)
    assert abs(v / (1920.0 * 1080.0) - 2.827) < 0.06   # 2.775: edge tiles are clipped
    # brute force on a small level
    w, h = 150, 50
    cnt = 0
    for ox in (0, 64):
        for oy in (0, 16):
            for bx in range((w + 68) // 69):
                for by in range((h + 20) // 21):
                    for y in range(by * 21 + oy, by * 21 + oy + 16):
                        for x in range(bx * 69 + ox, bx * 69 + ox + 64):
                            cnt += int(x < w and y < h)
    assert bench.tile_visits(w, h) == cnt


def test_param_block_roundtrip_and_sharding():
    blk = capi.ParamBlock()
    blk.kp = morph.KernParameters(morph.Parameters())
    blk.max_iter, blk.max_iter_drop_factor, blk.start_res, blk.math_mode = 500.0, 1.0, 32, 1
    cons = synth.make_constraints(1920, 1080, 8)
    raw = dist.pack_block(blk, cons)
    b2, c2 = dist.unpack_block(raw)
    assert bytes(b2) == bytes(blk) and np.array_equal(c2, cons) and b2.n_constraints == 8
    # 60 pairs over 8 GPUs: blocks of 7 or 8, every pair exactly once
    shards = [dist.shard_pairs(60, 8, r) for r in range(8)]
    assert sorted(sum(shards, [])) == list(range(60))
    assert set(len(s) for s in shards) <= {7, 8}
    assert all(s == list(range(s[0], s[-1] + 1)) for s in shards)
    assert dist.shard_pairs(30, 4, 3) == list(range(23, 30))


def test_bench_reads_the_counters_of_its_own_workload_shape(tmp_path):
    """bench.py quotes PMC bytes and SQ ratios only from the entry of the workload shape it runs (config,
    pairs per launch), template variants of one schedule's kernel combined by their launches.  Tested on a
    fixture written here (re-collecting profiles must not break a CPU unit test); the committed
    profiles/traffic_latest.json only has to keep the schema."""
    import json
    import os
    sq = {"calls": 10.0, "SQ_WAVE_CYCLES": 1000.0, "SQ_ACTIVE_INST_VALU": 160.0, "SQ_WAIT_ANY": 660.0, "SQ_WAIT_INST_ANY": 50.0,
          "SQ_INSTS_VALU": 4000.0, "SQ_WAVES": 2.0, "GRBM_GUI_ACTIVE": 80.0, "SQ_LDS_IDX_ACTIVE": 100.0, "SQ_LDS_BANK_CONFLICT": 28.0}
    fix = {"entries": [
        {"config": 1, "pairs_per_launch": 1, "source": "fixture",
         "per_kernel": {"k_optimize_fast<true, 13, 2, true>": 20e6, "k_optimize_fast<true, 7, 4, true>": 120e6, "k_pass_fast": 1.45e6},
         "per_kernel_launches": {"k_optimize_fast<true, 13, 2, true>": 3, "k_optimize_fast<true, 7, 4, true>": 1, "k_pass_fast": 2000},
         "sq_per_kernel": {"k_pass_fast": sq}},
        {"config": 2, "pairs_per_launch": 30, "source": "fixture", "per_kernel": {"k_optimize_fast<true, 13, 2, false>": 24e6},
         "per_kernel_launches": {"k_optimize_fast<true, 13, 2, false>": 8000}, "sq_per_kernel": {}, "sweep_source_sha256": bench.sweep_source_hash()}]}
    path = str(tmp_path / "traffic.json")
    json.dump(fix, open(path, "w"))
    k, n, src, sqk, stale = bench.load_pmc(1, 1, path)
    assert src and "config 1, 1 pair" in src and "fixture" in src
    assert stale is True              # no fingerprint of the sweep kernels' sources in that entry: cannot be vouched for
    assert bench.pmc_bytes(k, n, "k_pass_fast") == 1.45e6
    assert bench.pmc_bytes(k, n, "k_optimize_fast<true") == (3 * 20e6 + 120e6) / 4       # launch-weighted over both variants
    assert bench.pmc_bytes(k, n, "k_step_fast") is None
    m = bench.sq_measured(sqk, "k_pass_fast")
    assert m["valu_active_of_wave_cycles"] == 0.16 and m["wait_any_of_wave_cycles"] == 0.66
    assert m["valu_insts_per_wave"] == 2000.0 and m["lds_bank_conflict_of_lds_active"] == 0.28
    assert m["valu_issue_slots_used"] == round(4000.0 * 4.0 / (1024.0 * 80.0 / 8.0), 4)
    assert bench.sq_measured(sqk, "k_step_fast") is None
    # a batch line never borrows the single pair's counters; an unknown shape gets nothing
    k30, n30, _, _, stale30 = bench.load_pmc(2, 30, path)
    assert stale30 is False           # stamped with the sources as they are now
    assert bench.pmc_bytes(k30, n30, "k_pass_fast") is None and bench.pmc_bytes(k30, n30, "k_optimize_fast<true") == 24e6
    assert bench.load_pmc(2, 17, path) == ({}, {}, None, {}, None)
    assert bench.load_pmc(1, 1, str(tmp_path / "missing.json")) == ({}, {}, None, {}, None)
    # ... and tools/pmc_summary.py stamps what it merges with the same fingerprint bench.py computes
    import importlib.util
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pmc_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    assert ps.sweep_source_hash() == bench.sweep_source_hash()
    # the committed file: schema, and a fingerprint on every entry (a kernel change that forgets to refresh the profile
    # shows up as "traffic_stale": true in the bench line, not silently)
    tj = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")))
    assert isinstance(tj.get("entries"), list) and tj["entries"]
    for e in tj["entries"]:
        assert isinstance(e["config"], int) and isinstance(e["pairs_per_launch"], int) and isinstance(e["per_kernel"], dict)
        assert len(e.get("sweep_source_sha256", "")) == 64
        # ... and it is the fingerprint of the sources as they are NOW: whoever edits vm_sweep_kernels.hip / vm_morph_common.h /
        # vm_internal.h re-profiles (tools/prof_sweeps.sh + collect_profiles.sh) or the bench line says "traffic_stale": true
        assert e["sweep_source_sha256"] == bench.sweep_source_hash(), "profiles/traffic_latest.json is older than the sweep kernels' sources"


def test_bench_config2_chunks_and_batches():
    """bench.config2_setup / config2_step (no GPU: stand-in contexts and pyramids): a rank's pairs go to its contexts
    in contiguous chunks, a chunk is solved in EVEN batches of at most max_batch pairs, every pair exactly once per
    step, one host thread per context"""
    class Ctx(object):
        pass

    class Pyr(object):
        def __init__(self, ctx, img):
            self._ctx, self.img = ctx, img
    frames = lambda ids: [("f%d" % i,) * 2 for i in ids]
    # (streams as bench.default_streams() picks them: three from 24 pairs per GPU on, two below)
    assert [bench.default_streams(n) for n in (60, 30, 24, 23, 15, 8, 7, 1)] == [3, 3, 3, 2, 2, 2, 2, 2]
    assert os.environ.get("GPU_MAX_HW_QUEUES")          # importing bench.py asks for more hardware queues than HIP's four
    for npairs, nctx_in, max_batch, want_B, want_nctx in ((60, 2, 32, 30, 2), (8, 2, 32, 4, 2), (7, 2, 32, 4, 2), (60, 1, 32, 30, 1),
                                                            (1, 2, 32, 1, 1), (15, 2, 4, 4, 2), (60, 3, 32, 20, 3), (30, 3, 32, 10, 3)):
        ctxs = [Ctx() for _ in range(nctx_in)]
        pyrs, B, nctx, distinct = bench.config2_setup(list(range(100, 100 + npairs)), ctxs, frames, lambda c, img: Pyr(c, img), max_batch)
        assert (B, nctx) == (want_B, want_nctx), (npairs, nctx_in, max_batch, B, nctx)
        assert len(pyrs) == npairs and distinct == npairs        # every pair its own frame (rounds 1-4: 8 reused cyclically)
        owners = [ctxs.index(p._ctx) for p in pyrs]
        assert owners == sorted(owners) and set(owners) == set(range(nctx))            # contiguous chunks, every context used
        assert max(owners.count(k) for k in range(nctx)) - min(owners.count(k) for k in range(nctx)) <= 1
        assert [p.img[0] for p in pyrs] == ["f%d" % (100 + k % distinct) for k in range(npairs)]   # frames reused cyclically
        groups = []

        def solve_group(ps):
            groups.append(list(ps))
            return [id(p) for p in ps]
        res = bench.config2_step(pyrs, ctxs[:nctx], B, solve_group)
        assert sorted(res) == sorted(id(p) for p in pyrs)
        assert all(1 <= len(g) <= B and len(set(p._ctx for p in g)) == 1 for g in groups)
        assert sum(len(g) for g in groups) == npairs


def test_config4_plan_of_30_pairs_on_4_ranks():
    """config[4]'s static plan (`for config 5: 30 pairs/4 GPUs`): block partition 8 / 7 / 8 / 7, every rank's
    pairs in contiguous solver chunks, every pair in exactly one compositor batch of <= 4 frames, the batches dealt to
    the lanes in turn"""
    seen = []
    for r in range(4):
        mine, chunk_of, lanes = bench.config4_plan(30, 4, r)
        assert mine == dist.shard_pairs(30, 4, r) and len(mine) == (8, 7, 8, 7)[r]
        assert chunk_of == sorted(chunk_of) and set(chunk_of) == {0, 1}           # two solver streams for < 24 pairs
        flat = [k for lane in lanes for b in lane for k in b]
        assert sorted(flat) == list(range(len(mine))) and all(1 <= len(b) <= 4 for lane in lanes for b in lane)
        seen += mine
    assert seen == list(range(30))
    mine, chunk_of, lanes = bench.config4_plan(30, 1, 0)
    assert len(mine) == 30 and set(chunk_of) == {0, 1, 2} and [chunk_of.count(c) for c in (0, 1, 2)] == [10, 10, 10]
    assert [len(l) for l in lanes] == [4, 4] and lanes[0][0] == [0, 1, 2, 3] and lanes[1][0] == [4, 5, 6, 7] and lanes[1][-1] == [28, 29]
    assert bench.config4_plan(3, 4, 2)[0] == [2] and bench.config4_plan(3, 4, 3)[0] == [] and bench.config4_plan(2, 4, 1) == ([], [], [[], []])


def test_bench_times_the_poisson_stage_only_at_verified_tolerances():
    import fullsize_fixture as FX
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert bench.POISSON_TOL in FX.POISSON_TIMED_TOLS and 1e-6 in FX.POISSON_TIMED_TOLS and 1e-4 not in FX.POISSON_TIMED_TOLS
    # no literal tolerance is passed to the Poisson entry points but the untimed workspace warm-ups
    import re
    lits = re.findall(r"poisson_extend\w*\([^)]*tol=([0-9.e-]+)", src)
    assert set(lits) <= {"1e-3"}, lits
