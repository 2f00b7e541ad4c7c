"""dev: what vm_dbg_streams_overlap says about the solver contexts of a 2-/3-stream job when k idle contexts were created first
(bench.py's VM_DEV_DUMMY_STREAMS shows which k slow the job down: profiles/r06_notes.md section 4).  usage: solver_stream_probe.py [kmax] [nctx]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["VM_DBG_OVERLAP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                      # bench.py imports it (its streams take queues too)
torch.cuda.set_device(0)
from videomorphing_amd import capi, morph
kmax = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for k in range(kmax):
    d = [morph.Context(0, capi.MATH_FAST) for _ in range(k)]
    c = [morph.Context(0, capi.MATH_FAST) for _ in range(nctx)]
    print("k = %d:" % k, [(i, j, c[i].runs_beside(c[j])) for i in range(nctx) for j in range(i + 1, nctx)], flush=True)
    for x in d + c:
        x.close()
