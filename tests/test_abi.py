"""The C-ABI shared library: loads, exports every symbol include/vmorph.h declares,
struct layouts agree between the header (compiled with gcc) and the ctypes binding,
and it fails loudly without a GPU.  No compute calls (CPU only)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from videomorphing_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vmorph.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vm_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(vmlib):
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(vmlib, n), "libvmorph_hip.so does not export %s" % n
    assert sorted(capi.SYMBOLS) == names, "capi.SYMBOLS out of sync with include/vmorph.h"


def test_header_is_plain_c_and_struct_layouts_match(tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "vmorph.h"\nint main(void){printf("%zu %zu %zu %zu ",'
                    'sizeof(vm_kern_params),sizeof(vm_constraint),sizeof(vm_progress),sizeof(vm_param_block));'
                    'printf("%zu %zu %zu\\n",sizeof(vm_video_constraint),sizeof(vm_sync_constraint),sizeof(vm_sync_progress));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(capi.KernParams), C.sizeof(capi.Constraint), C.sizeof(capi.Progress),
                   C.sizeof(capi.ParamBlock), C.sizeof(capi.VideoConstraint), C.sizeof(capi.SyncConstraint),
                   C.sizeof(capi.SyncProgress)]
    assert got == [28, 20, 136, 48, 24, 24, 32]    # vm_progress: 104 until round 5 added the clock-probe sums


def test_header_cites_the_reference_interfaces():
    src = open(HEADER).read()
    for cite in ("morph.cu:150-168", "morph.cu:1353-1391", "morph.cu:264-390", "morph.cu:419-590",
                 "upsample.cu:260-286", "render.cu:62-96", "PoissonExt.cpp:49-362",
                 "MatchingThread.cpp:22-100", "parameters.h:54-72", "pyramid.cu:525-543",
                 "SyncThread.cpp:290-480", "render.cu:99-246", "upsample.cu:343-375", "pyramid.cu:143-163"):
        assert cite in src, cite


def test_no_gpu_means_loud_failure(vmlib):
    """the product path has no CPU fallback: without a HIP device vm_ctx_create fails
    with VM_E_DEVICE and a message (skipped on a GPU box)"""
    h = C.c_void_p()
    rc = vmlib.vm_ctx_create(0, C.byref(h))
    if rc == capi.VM_OK:
        vmlib.vm_ctx_destroy(h)
        pytest.skip("a HIP device is present")
    assert rc == capi.VM_E_DEVICE and not h.value
    assert b"no CPU fallback" in vmlib.vm_last_error()
    from videomorphing_amd import morph
    with pytest.raises(capi.VmError):
        morph.Context(0)


def test_null_handles_are_rejected_not_dereferenced(vmlib):
    assert vmlib.vm_ctx_sync(None) == capi.VM_E_INVALID
    assert vmlib.vm_level_dims(None, 0, None, None, None) == capi.VM_E_INVALID
    assert vmlib.vm_solve(None, 10.0, 1.0, None, 0, None, 0, None) == capi.VM_E_INVALID
    assert vmlib.vm_render_halfway(None, 0.5, 0.5, 1, None, 0) == capi.VM_E_INVALID
    assert vmlib.vm_poisson_extend(None, 1, 1e-5, 10, None, None, None) == capi.VM_E_INVALID
    assert vmlib.vm_pyramid_levels(None) == 0
    assert vmlib.vm_sync_solve(None, 10.0, None, None) == capi.VM_E_INVALID
    assert vmlib.vm_sync_render(None, 0.5, 0, None, 0) == capi.VM_E_INVALID
    vmlib.vm_sync_destroy(None)
    vmlib.vm_pyramid_destroy(None)
    vmlib.vm_frame_destroy(None)
    vmlib.vm_ctx_destroy(None)
    assert b"vmorph" in vmlib.vm_version()


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under videomorphing_amd/ or include/
    may import, link or name it"""
    bad = []
    for base in ("videomorphing_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in dp or os.sep + "lib" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"vm_oracle|vmo_|import oracle|from oracle|libvm_oracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
