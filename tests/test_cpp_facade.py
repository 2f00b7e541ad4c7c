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
