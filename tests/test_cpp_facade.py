"""The C++ host facade (include/vmorph/*.hpp, the mirror of Parameters / Pyramid /
Morph / CMatchingThread) builds with plain g++ against the C-ABI, fails loudly
without a GPU, and on a GPU produces the same field as the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

from videomorphing_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def solve_pair(tmp_path_factory, vmlib):
    exe = str(tmp_path_factory.mktemp("cpp") / "solve_pair")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_pair.cpp"), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def test_facade_builds_and_fails_loudly_without_gpu(solve_pair, vmlib, tmp_path):
    import ctypes as C
    h = C.c_void_p()
    if vmlib.vm_ctx_create(0, C.byref(h)) == capi.VM_OK:
        vmlib.vm_ctx_destroy(h)
        pytest.skip("a HIP device is present")
    z = tmp_path / "z.f32"
    np.zeros(64, np.float32).tofile(str(z))
    r = subprocess.run([solve_pair, "8", "8", str(z), str(z), str(tmp_path / "o.f32")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_facade_matches_python_mirror(solve_pair, gpu_ctx, tmp_path):
    from videomorphing_amd import morph
    w, h = 200, 120
    i0, i1 = synth.make_pair(w, h)
    i0.tofile(str(tmp_path / "a.f32"))
    i1.tofile(str(tmp_path / "b.f32"))
    out = tmp_path / "v.f32"
    r = subprocess.run([solve_pair, str(w), str(h), str(tmp_path / "a.f32"), str(tmp_path / "b.f32"),
                        str(out), "25", "32", "exact"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    v_cpp = np.fromfile(str(out), np.float32).reshape(h, w, 2)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 25, 1.0, 32
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build(i0, i1, 32)
    t = morph.MatchingThread(prm, pyr)
    t.start()
    t.wait()
    assert t.percentage == pytest.approx(100.0)
    assert np.array_equal(v_cpp.view(np.uint32), pyr._vector[0].view(np.uint32))
    assert np.abs(v_cpp).max() > 0.1


@pytest.fixture(scope="module")
def solve_video(tmp_path_factory, vmlib):
    exe = str(tmp_path_factory.mktemp("cppv") / "solve_video")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_video.cpp"), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def test_video_facade_builds(solve_video):
    assert os.path.exists(solve_video)


@pytest.mark.gpu
def test_video_facade_matches_python_mirror(solve_video, gpu_ctx, tmp_path):
    """the C++ VideoPyramid / VideoMatchingThread (device-side image + flow pyramid, coupled solve on a
    worker thread, update_result for every frame) and the Python mirror over the same C-ABI produce
    the same bits"""
    from videomorphing_amd import morph
    w, h, d = 96, 64, 4
    rgbs = [synth.make_rgb_pair(w, h, frame=t) for t in range(d)]
    f0, f1, b0, b1 = synth.constant_flows(w, h, d)
    np.concatenate([np.stack([a, b]).ravel() for a, b in rgbs]).astype(np.uint8).tofile(str(tmp_path / "fr.u8"))
    np.concatenate([np.stack([f0[t], f1[t], b0[t], b1[t]]).ravel() for t in range(d)]).astype(np.float32).tofile(str(tmp_path / "fl.f32"))
    out = tmp_path / "v.f32"
    r = subprocess.run([solve_video, str(w), str(h), str(d), str(tmp_path / "fr.u8"), str(tmp_path / "fl.f32"), str(out),
                        "12", "16", "exact"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    v_cpp = np.fromfile(str(out), np.float32).reshape(d, h, w, 2)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    levels, ft = synth.video_levels(w, h, d, 16)
    vid = morph.VideoPyramid(gpu_ctx)
    vid.build_levels(levels, ft, d)
    for t in range(d):
        vid.build_rgb_frame(t, *rgbs[t])
    vid.build_flows(f0, f1, b0, b1)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 12, 1.0, 16
    th = morph.VideoMatchingThread(prm, vid, w, h)
    th.run()
    for t in range(d):
        assert np.array_equal(v_cpp[t].view(np.uint32), vid._vector[t].view(np.uint32)), t
        assert np.array_equal(v_cpp[t].view(np.uint32), vid.pages[0][t].v.view(np.uint32)), t     # ratio 1: the page itself
    assert np.abs(v_cpp).max() > 0.05


def test_cpp_level_table_equals_the_python_geometry(tmp_path, vmlib):
    """VideoPyramid::level_table (C++) and synth.video_levels (Python, itself checked against the
    float form the reference evaluates) agree on the temporal pyramid's geometry"""
    src = tmp_path / "lt.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "vmorph/video.hpp"\nint main(int c, char **a){'
                   'auto t = vmorph::VideoPyramid::level_table(atoi(a[1]), atoi(a[2]), atoi(a[3]), atoi(a[4]));'
                   'for (auto &l : t) printf("%d %d %d %d\\n", l.width, l.height, l.depth, l.factor_t); return 0;}\n')
    exe = str(tmp_path / "lt")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir])
    for (w, h, d, sr) in [(80, 64, 16, 8), (80, 64, 32, 8), (1920, 1080, 60, 32), (320, 200, 40, 8), (256, 256, 1, 64), (127, 99, 5, 4)]:
        got = [tuple(int(x) for x in l.split()) for l in subprocess.check_output([exe, str(w), str(h), str(d), str(sr)]).decode().splitlines()]
        levels, ft = synth.video_levels(w, h, d, sr)
        assert got == [l + (f,) for l, f in zip(levels, ft)], (w, h, d, sr)


@pytest.fixture(scope="module")
def solve_sync(tmp_path_factory, vmlib):
    exe = str(tmp_path_factory.mktemp("cpps") / "solve_sync")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_sync.cpp"), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def test_sync_facade_builds(solve_sync):
    assert os.path.exists(solve_sync)


@pytest.mark.gpu
def test_sync_facade_matches_python_mirror(solve_sync, gpu_ctx, tmp_path):
    """the C++ SyncPyramid / SyncThread (CSyncThread::run, update_result, the stage-1 renderer) and
    the Python mirror over the same C-ABI produce the same bits"""
    from videomorphing_amd import morph
    w, h, d = 96, 64, 5
    rng = np.random.default_rng(31)
    video = [rng.integers(0, 256, (d, h, w, 4), dtype=np.uint8) for _ in range(2)]
    flows = [rng.standard_normal((d, h, w, 2)).astype(np.float32) for _ in range(2)]
    cons = [(20, 20, 1, 26, 24, 2), (70, 40, 3, 64, 44, 1), (48, 12, 0, 50, 10, 0), (30, 50, 4, 34, 52, 2)]
    np.stack([np.stack([video[0][t], video[1][t]]) for t in range(d)]).tofile(str(tmp_path / "fr.u8"))
    np.stack([np.stack([flows[0][t], flows[1][t]]) for t in range(d)]).tofile(str(tmp_path / "fl.f32"))
    np.asarray(cons, np.int32).tofile(str(tmp_path / "c.i32"))
    r = subprocess.run([solve_sync, str(w), str(h), str(d), str(tmp_path / "fr.u8"), str(tmp_path / "fl.f32"), str(tmp_path / "c.i32"),
                        str(len(cons)), str(tmp_path / "field.f32"), str(tmp_path / "out.u8"), "6", "16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    P = morph.Parameters()
    P.w_ui, P.w_tps, P.max_iter, P.start_res = 100.0, 0.001, 6, 16
    for k, c in enumerate(cons):
        P.lp.append([morph.Conp(c[0], c[1], c[2])])
        P.rp.append([morph.Conp(c[3], c[4], c[5])])
        P.cnt.append([morph.Connect((k, 0), (k, 0))])
    pyr = morph.SyncPyramid(gpu_ctx)
    pyr.build(video[0], video[1], flows[0], flows[1], 16)
    th = morph.SyncThread(P, pyr)
    th.run()
    X, Y, Z = pyr.field(1)
    lw, lh, ld = pyr.levels[1]
    got = np.fromfile(str(tmp_path / "field.f32"), np.float32).reshape(3, ld, lh, lw)
    for c, want in enumerate((X, Y, Z)):
        assert np.array_equal(got[c].view(np.uint32), want.view(np.uint32)), c
    frames = np.fromfile(str(tmp_path / "out.u8"), np.uint8).reshape(d, 2, h, w, 3)
    for t in range(d):
        for side in range(2):
            assert np.array_equal(frames[t, side], pyr.render_resample(float(side), t)), (t, side)
    assert np.abs(X).max() > 0.01 and np.abs(Z).max() > 0.001


@pytest.fixture(scope="module")
def solve_shard(tmp_path_factory, vmlib):
    exe = str(tmp_path_factory.mktemp("cppm") / "solve_shard")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "solve_shard.cpp"), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def test_cpp_shard_plan_equals_the_python_partition(solve_shard):
    """the C++ multi-device driver partitions the pairs like videomorphing_amd.dist.shard_pairs (what bench.py's
    N-process form uses): pair k -> rank floor(k G / N)"""
    from videomorphing_amd import dist as vdist
    for G, N in [(8, 60), (4, 30), (2, 5), (3, 2), (8, 8), (1, 7)]:
        out = subprocess.check_output([solve_shard, "--plan", str(G), str(N)]).decode().splitlines()
        got = [[int(x) for x in ln.split(":")[1].split()] for ln in out]
        assert got == [vdist.shard_pairs(N, G, r) for r in range(G)], (G, N)


@pytest.mark.gpu
def test_cpp_multi_device_driver_matches_python(solve_shard, gpu_ctx, tmp_path):
    """examples/solve_shard.cpp -- one C++ process, G host threads x G contexts, the parameter block handed from
    rank 0 to every context (vm_bcast_params; here its one-device test mode: this box has one GPU and RCCL wants one
    rank per device), static block partition, vm_solve_batch per context, results gathered on the host -- with
    G = 2 contexts on device 0 and 5 pairs: every field bit-identical to the Python path (one vm_solve per pair),
    and the non-default parameters really travelled (max_iter 18, start_res 16)."""
    from videomorphing_amd import morph
    w, h, N = 150, 100, 5
    frames = [synth.make_pair(w, h, frame=k, amp=0.4 + 0.2 * k) for k in range(N)]
    np.concatenate([np.stack([a, b]).ravel() for a, b in frames]).astype(np.float32).tofile(str(tmp_path / "fr.f32"))
    out = tmp_path / "v.f32"
    r = subprocess.run([solve_shard, "2", str(w), str(h), str(N), str(tmp_path / "fr.f32"), str(out), "18", "16", "exact", "--one-device"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "rank 0 on device 0: 3 pairs" in r.stdout and "rank 1 on device 0: 2 pairs" in r.stdout
    assert r.stdout.count("max_iter 18 start_res 16 math 0") == 2, r.stdout
    v_cpp = np.fromfile(str(out), np.float32).reshape(N, h, w, 2)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 18, 1.0, 16
    for k, (i0, i1) in enumerate(frames):
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 16)
        t = morph.MatchingThread(prm, pyr)
        t.start()
        t.wait()
        assert np.array_equal(v_cpp[k].view(np.uint32), pyr._vector[0].view(np.uint32)), k
    assert np.abs(v_cpp).max() > 0.1


@pytest.mark.gpu
def test_cpp_driver_over_rccl_with_one_rank(solve_shard, gpu_ctx, tmp_path):
    """the same driver WITHOUT the one-device switch and G = 1: ncclCommInitAll over this box's one device, the parameter
    block through ncclBroadcast inside an ncclGroup (vm_bcast_params' production route), then the shard -- the fields
    equal the Python path's, and the process exits cleanly with RCCL loaded"""
    from videomorphing_amd import morph
    w, h, N = 120, 80, 2
    frames = [synth.make_pair(w, h, frame=k, amp=0.5 + 0.3 * k) for k in range(N)]
    np.concatenate([np.stack([a, b]).ravel() for a, b in frames]).astype(np.float32).tofile(str(tmp_path / "fr.f32"))
    out = tmp_path / "v.f32"
    r = subprocess.run([solve_shard, "1", str(w), str(h), str(N), str(tmp_path / "fr.f32"), str(out), "15", "16", "exact"],
                       capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert "rank 0 on device 0: 2 pairs" in r.stdout and "max_iter 15 start_res 16 math 0" in r.stdout
    v_cpp = np.fromfile(str(out), np.float32).reshape(N, h, w, 2)
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 15, 1.0, 16
    for k, (i0, i1) in enumerate(frames):
        pyr = morph.Pyramid(gpu_ctx)
        pyr.build(i0, i1, 16)
        t = morph.MatchingThread(prm, pyr)
        t.start()
        t.wait()
        assert np.array_equal(v_cpp[k].view(np.uint32), pyr._vector[0].view(np.uint32)), k


_BCAST_CHILD = """
import ctypes as C, sys
sys.path.insert(0, %r)
from videomorphing_amd import capi, morph
L = capi.load()
ctx = morph.Context(0, capi.MATH_EXACT)
comm = (C.c_void_p * 1)()
dev = (C.c_int * 1)(0)
capi.check(L.vm_rccl_comm_init_all(1, dev, comm))
blk = capi.ParamBlock()
prm = morph.Parameters()
prm.w_tps, prm.eps = 0.07, 0.02
blk.kp = morph.KernParameters(prm)
blk.max_iter, blk.max_iter_drop_factor, blk.start_res, blk.math_mode = 77.0, 2.0, 24, capi.MATH_FAST
got = (capi.ParamBlock * 1)()
hs = (C.c_void_p * 1)(ctx._h)
capi.check(L.vm_bcast_params(hs, comm, 1, 0, C.byref(blk), got))
L.vm_rccl_comm_destroy(comm[0])
assert bytes(got[0]) == bytes(blk)
kp = capi.KernParams()
capi.check(L.vm_get_params(ctx._h, C.byref(kp)))
assert abs(kp.w_tps - 0.07) < 1e-7 and abs(kp.eps - 0.02) < 1e-7
two = (C.c_int * 2)(0, 0)
comms2 = (C.c_void_p * 2)()
assert L.vm_rccl_comm_init_all(2, two, comms2) != capi.VM_OK and "listed twice" in L.vm_last_error().decode()
ctx.close()
print("BCAST-OK")
"""


@pytest.mark.gpu
def test_bcast_params_over_rccl_on_one_device(tmp_path):
    """vm_rccl_comm_init_all + vm_bcast_params with a real communicator (one rank: this box's one GPU): the block goes
    through ncclBroadcast inside an ncclGroup and the context adopts it; two contexts on one device are refused by
    vm_rccl_comm_init_all with the pointer to the test mode.  In a child process of its own, like
    tests/test_gpu_rccl.py: RCCL and the HIP runtime must come from ONE ROCm copy, and the pytest process may already
    hold torch's (measured: the system's librccl beside torch's HIP runtime aborts at interpreter exit)."""
    import sys
    script = tmp_path / "bcast_child.py"
    script.write_text(_BCAST_CHILD % ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BCAST-OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.fixture(scope="module")
def pipeline_shard(tmp_path_factory, vmlib):
    exe = str(tmp_path_factory.mktemp("cppp") / "pipeline_shard")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "pipeline_shard.cpp"), "-o", exe,
                           "-L", libdir, "-lvmorph_hip", "-Wl,-rpath," + libdir, "-lpthread"])
    return exe


def test_pipeline_shard_builds_and_refuses_bad_input(pipeline_shard, tmp_path):
    r = subprocess.run([pipeline_shard], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([pipeline_shard, "2", "8", "8", "1", "2", str(tmp_path / "a"), str(tmp_path / "b"), str(tmp_path / "c"),
                        str(tmp_path / "d"), str(tmp_path / "e")], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot read" in r.stderr


@pytest.mark.gpu
def test_cpp_config4_pipeline_on_two_contexts_matches_python(pipeline_shard, gpu_ctx, tmp_path):
    """examples/pipeline_shard.cpp -- config[4] from one C++ process: the parameter block AND the point constraints in one
    broadcast (vm_bcast_bytes; here its one-device test mode), 5 pairs block-sharded over G = 2 contexts, per rank
    vm_solve_batch_cons (BCOND_BORDER) -> canvases -> v upscaled on the device -> Poisson extension in batches -> render.
    Against the Python mirror on one context: every halfway field bit-identical (EXACT), every rendered frame within one
    colour level (the Poisson solver's dot products are accumulated by atomics)."""
    from videomorphing_amd import morph
    w, h, N, ex = 150, 100, 5, 15
    frames = [synth.make_pair(w, h, frame=k, amp=0.4 + 0.2 * k) for k in range(N)]
    rgbs = [synth.make_rgb_pair(w, h, frame=k, amp=0.4 + 0.2 * k) for k in range(N)]
    cons = synth.make_constraints(w, h, 8)
    np.concatenate([np.stack([a, b]).ravel() for a, b in frames]).astype(np.float32).tofile(str(tmp_path / "fr.f32"))
    np.concatenate([np.stack([a, b]).ravel() for a, b in rgbs]).astype(np.uint8).tofile(str(tmp_path / "rgb.u8"))
    cons.astype(np.float32).tofile(str(tmp_path / "cons.f32"))
    r = subprocess.run([pipeline_shard, "2", str(w), str(h), str(N), str(ex), str(tmp_path / "fr.f32"), str(tmp_path / "rgb.u8"), str(tmp_path / "cons.f32"),
                        str(tmp_path / "v.f32"), str(tmp_path / "out.u8"), "18", "16", "exact", "--one-device"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "rank 0 on device 0: 3 pairs" in r.stdout and "rank 1 on device 0: 2 pairs" in r.stdout
    assert r.stdout.count("max_iter 18 bcond 2 constraints 8") == 2, r.stdout          # block and constraints really travelled
    v_cpp = np.fromfile(str(tmp_path / "v.f32"), np.float32).reshape(N, h, w, 2)
    img_cpp = np.fromfile(str(tmp_path / "out.u8"), np.uint8).reshape(N, h, w, 3)
    try:
        gpu_ctx.set_math_mode(capi.MATH_EXACT)
        prm = morph.Parameters()
        prm.max_iter, prm.max_iter_drop_factor, prm.start_res, prm.bcond = 18, 1.0, 16, capi.BCOND_BORDER
        gpu_ctx.set_params(morph.KernParameters(prm))
        pyrs = []
        for i0, i1 in frames:
            q = morph.Pyramid(gpu_ctx)
            q.build(i0, i1, 16)
            pyrs.append(q)
        morph.solve_batch(pyrs, 18, 1.0, constraints=cons)
        fr = morph.Frame(gpu_ctx, w, h, ex)
        for k, q in enumerate(pyrs):
            fr.upload(morph.make_extended(rgbs[k][0], ex), morph.make_extended(rgbs[k][1], ex), None, None)
            fr.set_v_from_level(q, 1)
            assert np.array_equal(v_cpp[k].view(np.uint32), fr.download_v().view(np.uint32)), k
            fr.poisson_extend_both(tol=1e-5)
            img = fr.render_halfway(0.5, 0.5, 1)
            assert np.abs(img.astype(int) - img_cpp[k].astype(int)).max() <= 1, k
        fr.close()
        assert np.abs(v_cpp).max() > 0.1 and img_cpp.std() > 5
    finally:
        gpu_ctx.set_params(morph.KernParameters(morph.Parameters()))
