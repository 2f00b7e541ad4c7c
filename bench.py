#!/usr/bin/env python
"""bench.py -- headline benchmark of the halfway-domain morph solver.

Metric (BASELINE.json): Mpixel*iters/s of the halfway optimizer on a 1080p frame
pair, 6-level pyramid (start_res 32), 500 iterations per level (config[1]).

A "step" = one complete coarse-to-fine solve (vm_solve: coarse solve, then per
level upsample + init + sweeps) of ONE synthetic 1080p frame pair whose pyramid
is already resident in HBM.  value = sum over levels of W*H*iterations executed
(the reference's own progress unit, morph.cu:1389) divided by wall time, summed
over all ranks.  Independent frame pairs shard across GPUs with no data-path
collective (weak scaling: K pairs per GPU); the only collective is one RCCL
broadcast of the shared parameter block.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

W, H, START_RES, MAX_ITER, DROP = 1920, 1080, 32, 500.0, 1.0
ALG_BYTES_PER_VISIT = 100.0      # 76 B read + 24 B written per pixel-visit
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec


def tile_visits(w, h):
    """pixel-visits of the 4 tile-offset launches of one iteration (64x16 tiles, 69x21 pitch)"""
    def cover(dim, tile, pitch, off):
        n, t = 0, off
        while t < dim:
            n += min(tile, dim - t)
            t += pitch
        return n
    tot = 0
    for ox in (0, 64):
        for oy in (0, 16):
            tot += cover(w, 64, 69, ox) * cover(h, 16, 21, oy)
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--math", default="fast", choices=["fast", "exact"])
    ap.add_argument("--semantics", default="fixed", choices=["fixed", "reference"],
                    help="fixed: every level runs its max_iter sweeps (BASELINE config: 500 iters/level); "
                         "reference: a level stops when no pixel improved (morph.cu:1390)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="independent frame pairs solved concurrently per GPU (one HIP stream and one "
                         "host thread each); 1 = one pair at a time")
    ap.add_argument("--batch", type=int, default=1,
                    help="frame pairs solved together by the same sweep launches (vm_solve_batch); a step "
                         "is then one batch.  1 = config[1], one pair per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--size", default=None, help="WxH override (debug only; invalid as a result)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    w, h = W, H
    if args.size:
        w, h = [int(x) for x in args.size.lower().split("x")]

    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from videomorphing_amd import capi, morph, synth

    # ---- the shared parameter block: rank 0 decides, one RCCL broadcast ----
    from videomorphing_amd import dist as vdist
    blk = capi.ParamBlock()
    if rank == 0:
        P = morph.Parameters()
        blk.kp = morph.KernParameters(P)
        blk.max_iter, blk.max_iter_drop_factor, blk.start_res = MAX_ITER, DROP, START_RES
        blk.math_mode = capi.MATH_FAST if args.math == "fast" else capi.MATH_EXACT
    raw = vdist.pack_block(blk)
    if world > 1:
        # backend "nccl" is RCCL on ROCm (xGMI within the node)
        raw = vdist.broadcast_block(raw, torch.device("cuda", local_rank))
    blk, _cons = vdist.unpack_block(raw)

    nctx = max(1, args.inflight)
    ctxs = [morph.Context(local_rank, blk.math_mode) for _ in range(nctx)]
    for c in ctxs:
        c.set_params(blk.kp)
    ctx = ctxs[0]
    L = capi.load()

    # ---- inputs: K+W independent frame pairs per rank, pyramids resident in HBM ----
    nlev = synth.num_levels(w, h, blk.start_res)
    B = max(1, args.batch)
    npairs = (args.steps + args.warmup) * B
    pyrs = []
    for k in range(npairs):
        frame = rank * npairs + k
        i0, i1 = synth.make_pair(w, h, frame=frame)
        p = morph.Pyramid(ctxs[(k // B) % nctx])
        p.build(i0, i1, blk.start_res, nlevels=nlev)
        pyrs.append(p)
    sizes = [(pyrs[0][el].width, pyrs[0][el].height) for el in range(1, nlev + 1)]

    FIXED = 1 if args.semantics == "fixed" else 0

    def solve(p, fixed=FIXED):
        prog = (capi.Progress * (nlev - 1))()
        capi.check(L.vm_solve(p._h, blk.max_iter, blk.max_iter_drop_factor, None, 0, None, fixed, prog))
        return prog

    def solve_group(ps, fixed=FIXED):
        """one step: a batch of B pairs relaxed by the same launches"""
        if len(ps) == 1:
            return [solve(ps[0], fixed)]
        arr = (C.c_void_p * len(ps))(*[p._h for p in ps])
        prog = (capi.Progress * (len(ps) * (nlev - 1)))()
        capi.check(L.vm_solve_batch(arr, len(ps), blk.max_iter, blk.max_iter_drop_factor, None, fixed, prog))
        out = []
        for i in range(len(ps)):
            one = (capi.Progress * (nlev - 1))(*[prog[i * (nlev - 1) + k] for k in range(nlev - 1)])
            if i > 0:          # elapsed/launches are per batch: count them once
                for k in range(nlev - 1):
                    one[k].elapsed_ms, one[k].launches = 0.0, 0
            out.append(one)
        return out

    def run_steps(ps):
        """solve the pyramids; with several contexts, one host thread per context works
        through that context's pyramids (a vm_ctx is single-threaded by contract)"""
        if B > 1:
            out = []
            for g0 in range(0, len(ps), B):
                out += solve_group(ps[g0:g0 + B])
            return out
        if nctx == 1:
            return [solve(p) for p in ps]
        from concurrent.futures import ThreadPoolExecutor
        groups = {}
        for i, p in enumerate(ps):
            groups.setdefault(id(p._ctx), []).append((i, p))
        out = [None] * len(ps)

        def work(items):
            for i, p in items:
                out[i] = solve(p)
        with ThreadPoolExecutor(max_workers=len(groups)) as ex:   # ctypes calls release the GIL
            list(ex.map(work, groups.values()))
        return out

    def sync_all():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run_steps(pyrs[:args.warmup * B])
    sync_all()
    t0 = time.perf_counter()
    progs = run_steps(pyrs[args.warmup * B:])
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sync_all()

    pix_iters = sum(pr[i].pixel_iters for pr in progs for i in range(nlev - 1))
    kern_ms = sum(pr[i].elapsed_ms for pr in progs for i in range(nlev - 1))
    launches = sum(pr[i].launches for pr in progs for i in range(nlev - 1))
    alg_bytes = 0.0
    for pr in progs:
        for i in range(nlev - 1):
            # launches enqueued past convergence exit at once: count executed iterations only
            alg_bytes += pr[i].iters * tile_visits(*sizes[i]) * ALG_BYTES_PER_VISIT
    iters_per_level = [[pr[i].iters for i in range(nlev - 1)] for pr in progs]
    activity = {"active_tile_visits": sum(pr[i].active_tiles for pr in progs for i in range(nlev - 1)) / len(progs),
                "line_searches": sum(pr[i].candidates for pr in progs for i in range(nlev - 1)) / len(progs),
                "commits": sum(pr[i].commits for pr in progs for i in range(nlev - 1)) / len(progs),
                "pixel_visits": sum(pr[i].iters * tile_visits(*sizes[i]) for pr in progs for i in range(nlev - 1)) / len(progs)}

    if world > 1:
        el_max, pix_total = vdist.reduce_report(el, pix_iters, torch.device("cuda", local_rank))
    else:
        el_max, pix_total = el, pix_iters

    extras = {}
    if rank == 0 and not args.no_extras:
        # the other stopping rule and the other arithmetic mode, one solve each
        p = pyrs[0]
        ctx.sync(); t1 = time.perf_counter(); pr = solve(p, fixed=1 - FIXED); ctx.sync(); dt = time.perf_counter() - t1
        extras["%s_semantics" % ("reference" if FIXED else "fixed")] = {
            "mpix_iters_per_s": round(sum(pr[i].pixel_iters for i in range(nlev - 1)) / dt / 1e6, 2),
            "ms_per_solve": round(dt * 1e3, 2),
            "iters_per_level_fine_to_coarse": [pr[i].iters for i in range(nlev - 1)]}
        other = capi.MATH_EXACT if blk.math_mode == capi.MATH_FAST else capi.MATH_FAST
        ctx.set_math_mode(other)
        ctx.sync(); t1 = time.perf_counter(); pr = solve(p); ctx.sync(); dt = time.perf_counter() - t1
        extras["%s_math_mpix_iters_per_s" % ("exact" if other == capi.MATH_EXACT else "fast")] = round(
            sum(pr[i].pixel_iters for i in range(nlev - 1)) / dt / 1e6, 2)
        ctx.set_math_mode(blk.math_mode)
        # batched throughput: B independent pairs relaxed by the same launches (the per-GPU
        # workload of config[2]: 60 pairs over 8 GPUs); frames are reused cyclically
        if B == 1:
            base_frames = [synth.make_pair(w, h, frame=1000 + k) for k in range(4)]
            bt = {}
            for nb in (8, 32):
                group = []
                for k in range(nb):
                    q = morph.Pyramid(ctx)
                    q.build(base_frames[k % 4][0], base_frames[k % 4][1], blk.start_res, nlevels=nlev)
                    group.append(q)
                solve_group(group)                                    # warm-up
                ctx.sync(); t1 = time.perf_counter(); pr = solve_group(group); ctx.sync()
                dt = time.perf_counter() - t1
                bt["pairs_%d" % nb] = {
                    "mpix_iters_per_s": round(sum(q2[i].pixel_iters for q2 in pr for i in range(nlev - 1)) / dt / 1e6, 1),
                    "ms_per_batch": round(dt * 1e3, 1)}
                del group
            extras["batched_throughput_fixed_work"] = bt
        # compositor: frames/s of render_halfway with device-resident inputs
        ex = int(0.1 * max(w, h))
        rgb0, rgb1 = synth.make_rgb_pair(w, h)
        fr = morph.Frame(ctx, w, h, ex)
        fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), None, None)
        fr.set_v_from_level(p, 1)
        fr.render_halfway_dev(0.5, 0.5, 1)
        ms = [fr.render_halfway_dev(0.5, 0.1 * k, 1) for k in range(1, 10)]
        extras["render_frames_per_s"] = round(1000.0 / (sum(ms) / len(ms)), 1)
        # Poisson boundary extension of both sides of that frame (config[4]'s other stage)
        pe = {}
        fr.poisson_extend(1, tol=1e-3)      # the workspace is allocated on first use: not timed
        for tol in (1e-4, 1e-5):
            fr.upload(morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex), None, None)
            r1, r2 = fr.poisson_extend(1, tol=tol), fr.poisson_extend(2, tol=tol)
            pe["tol_%g" % tol] = {"ms_per_frame": round(r1[2] + r2[2], 1), "cg_iterations": [r1[0], r2[0]]}
        extras["poisson_extend_1080p_ex%d" % ex] = pe
        # quadratic motion path of that frame (QuadraticPath.cpp), SURVEY 8(f) rank 4
        fr.set_v_from_level(p, 1)
        try:
            qp = fr.quadratic_path(tol=1e-4)
            extras["quadratic_path_1080p"] = {"ms_per_frame": round(qp[2], 1), "pcg_iterations": qp[0], "tol": 1e-4,
                                              "residual": float("%.3g" % qp[1])}
        except capi.VmError as e:        # e.g. a folded v: the blend of the Jacobians is 0/0 there
            extras["quadratic_path_1080p"] = {"error": str(e)[-120:]}
        # config[4]'s whole pipeline on one GPU: 8 frame pairs solved as a batch (reference
        # semantics), then per frame v upscale -> Poisson extension of both sides -> 9 rendered
        # in-between frames; canvases uploaded once per frame (PCIe included)
        if B == 1:
            group = []
            for k in range(8):
                q = morph.Pyramid(ctx)
                q.build(base_frames[k % 4][0], base_frames[k % 4][1], blk.start_res, nlevels=nlev)
                group.append(q)
            e0, e1 = morph.make_extended(rgb0, ex), morph.make_extended(rgb1, ex)
            ctx.sync(); t1 = time.perf_counter()
            morph.solve_batch(group, blk.max_iter, blk.max_iter_drop_factor, fixed_work=False)
            t_solve = time.perf_counter() - t1
            for q in group:
                fr.upload(e0, e1, None, None)
                fr.set_v_from_level(q, 1)
                fr.poisson_extend(1, tol=1e-5)
                fr.poisson_extend(2, tol=1e-5)
                for k in range(1, 10):
                    fr.render_halfway_dev(0.1 * k, 0.1 * k, 1)
            ctx.sync(); dt = time.perf_counter() - t1
            extras["pipeline_config4_8_pairs"] = {"ms_per_pair": round(dt * 1e3 / 8, 1), "solve_ms_per_pair": round(t_solve * 1e3 / 8, 1),
                                                  "rendered_frames_per_s": round(8 * 9 / dt, 1)}
            del group
        fr.close()

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(np, synth, blk, pyrs[args.warmup * B:])

    if rank == 0:
        avg_launch_us = kern_ms * 1e3 / max(launches, 1)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixel*iters/s (halfway optimizer, 1080p pair, 6-level pyramid, 500 iters/level)",
            "value": round(pix_total / el_max / 1e6, 2),
            "unit": "Mpixel*iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el_max / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config[1]: %dx%d frame pair, %d-level pyramid (start_res %d), max_iter %d/level, drop %g, %s per step per GPU"
                                   % (w, h, nlev, blk.start_res, int(blk.max_iter), blk.max_iter_drop_factor,
                                      "one pair" if B == 1 else "a batch of %d independent pairs" % B),
                       "math_mode": "fast" if blk.math_mode == capi.MATH_FAST else "exact",
                       "semantics": "fixed work: every sweep of every level is launched" if FIXED
                                    else "reference: a level stops when no pixel improved",
                       "iters_per_level_fine_to_coarse": iters_per_level[0],
                       "pairs_in_flight_per_gpu": nctx, "pairs_per_step": B,
                       "parallelism": "independent frame pairs, %d rank(s), 1 RCCL broadcast" % world},
            "roofline": {"bound": "hbm", "kernel": "sweep kernels (k_optimize | k_step | k_decide + k_commit)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_us": round(avg_launch_us, 2), "launches": launches,
                         "alg_bytes_per_pixel_visit": ALG_BYTES_PER_VISIT},
            "cpu_baseline": cpu,
        }
        out["activity_per_solve"] = {k: round(v) for k, v in activity.items()}
        out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def effective_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(np, synth, blk, gpu_pyrs):
    """The CPU restatement (oracle, OpenMP over tiles) timed on this host on a bounded
    sample of the same workload: level by level (finest first) of the solves just timed,
    each level started -- like the GPU path -- from the upsampled solution of the next
    coarser level and swept until no pixel improves, until about 12 s have been spent."""
    import oracle as O
    threads = effective_cpus()
    O.lib().vmo_set_threads(threads)
    P = O.default_params()
    for f, _ in P._fields_:
        setattr(P, f, getattr(blk.kp, f))
    stats = np.zeros(4)
    units, spent, parts = 0.0, 0.0, []
    nlev = gpu_pyrs[0].size() - 1
    for gp in gpu_pyrs:
        for el in range(1, nlev - 1):
            if spent > 12.0:
                break
            w, h = gp[el].width, gp[el].height
            coarse = O.Level(gp[el + 1].width, gp[el + 1].height)
            coarse.field("v")[...] = gp[el + 1].v
            tgt = O.Level(w, h)
            tgt.set_images(gp[el].field("img0"), gp[el].field("img1"))
            tgt.upsample_from(coarse)
            tgt.init(P.ssim_clamp)
            t0 = time.perf_counter()
            iters = 0
            while time.perf_counter() - t0 < 8.0 and iters < int(blk.max_iter):
                imp = tgt.optimize_iter(P, stats)
                iters += 1
                if not imp:
                    break
            spent += time.perf_counter() - t0
            units += float(w) * h * iters
            parts.append("%dx%d:%d" % (w, h, iters))
    return {"value": round(units / spent / 1e6, 3), "unit": "Mpixel*iters/s",
            "cores": threads, "kind": "port",
            "sample": "reference-semantics sweeps of levels [%s] of the solves just timed (oracle, OpenMP "
                      "over tiles, %d threads), %.1f s; %.0f energy evaluations" % (", ".join(parts), threads, spent, stats[3])}


if __name__ == "__main__":
    main()
