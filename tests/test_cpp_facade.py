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
