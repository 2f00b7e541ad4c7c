"""vm_rccl_bcast: the C-ABI's RCCL broadcast of the shared parameter block, exercised on
a one-rank communicator (the GPU box has one device; the N>1 plumbing is covered by the
gloo test and by bench.py under torch.distributed).  Runs in a child process that imports
torch FIRST, like bench.py does: HIP, HSA and RCCL must all come from one ROCm copy."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    import torch
    sys.path.insert(0, %r)
    from videomorphing_amd import capi, morph
    torch.cuda.set_device(0)
    ctx = morph.Context(0)
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    rc = rccl.ncclCommInitAll(C.byref(comm), 1, devs)
    if rc != 0:
        rccl.ncclGetErrorString.restype = C.c_char_p
        print("SKIP", rccl.ncclGetErrorString(rc)); sys.exit(0)
    blk = capi.ParamBlock()
    blk.kp = morph.KernParameters(morph.Parameters())
    blk.max_iter, blk.max_iter_drop_factor, blk.start_res, blk.math_mode = 500.0, 1.0, 32, 1
    raw = np.frombuffer(bytes(blk), dtype=np.uint8).copy()
    t = torch.from_numpy(raw).cuda()
    capi.check(capi.load().vm_rccl_bcast(ctx._h, comm, C.c_void_p(t.data_ptr()), raw.size, 0))
    ctx.sync()
    assert np.array_equal(t.cpu().numpy(), raw)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    print("OK")
""")


def test_rccl_broadcast_of_param_block(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD % ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    if "SKIP" in r.stdout:
        pytest.skip("RCCL communicator could not be created on this box: " + r.stdout.strip())
    assert "OK" in r.stdout
