"""dev: does a process that used the library's RCCL helpers (vm_rccl_comm_init_all / vm_bcast_params /
vm_rccl_comm_destroy) exit cleanly?  argv[1]: plain | torch (import torch first) | nodestroy"""
import ctypes as C, os, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode == "torch":
    import torch
    torch.cuda.set_device(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from videomorphing_amd import capi, morph
L = capi.load()
ctx = morph.Context(0, capi.MATH_EXACT)
comm = (C.c_void_p * 1)()
dev = (C.c_int * 1)(0)
capi.check(L.vm_rccl_comm_init_all(1, dev, comm))
blk = capi.ParamBlock()
blk.kp = morph.KernParameters(morph.Parameters())
blk.max_iter = 77.0
got = (capi.ParamBlock * 1)()
hs = (C.c_void_p * 1)(ctx._h)
capi.check(L.vm_bcast_params(hs, comm, 1, 0, C.byref(blk), got))
if mode != "nodestroy":
    L.vm_rccl_comm_destroy(comm[0])
ctx.close()
print("OK", mode, got[0].max_iter, flush=True)
