"""Generates tests/golden/flow_ref.npz: optical-flow fields scaled by the REFERENCE's own
resampling library (oracle/_ref/libresample_ref.so, `make -C oracle ref`) in the call order of
the flow half of Pyramid::build (pyramid.cu:284-321, 375-404).  Run in the build container:
    python tests/golden/make_flow_golden.py
The fixture holds data only: input flows and the expected scaled flows."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libresample_ref.so"))
lib.ref_flow_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

cases = {}
rng = np.random.RandomState(11)
for name, (w, h, wo, ho) in {"same": (40, 28, 40, 28), "half": (40, 28, 20, 14), "odd": (33, 21, 17, 11), "wide": (48, 20, 24, 10)}.items():
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    flow = np.stack([3.0 * np.sin(x / 7.0) + 0.05 * y - 1.5, -2.0 * np.cos(y / 5.0) + 0.02 * x], axis=-1).astype(np.float32)
    flow += rng.randn(h, w, 2).astype(np.float32) * 0.3
    flow[2, 3] = (60.0, -70.0)                 # beyond the [-50, 50] range the reference clamps to
    out = np.zeros((ho, wo, 2), dtype=np.float32)
    lib.ref_flow_scale(np.ascontiguousarray(flow).ctypes.data, w, h, wo, ho, out.ctypes.data)
    cases[name + "_in"] = flow
    cases[name + "_out"] = out
np.savez_compressed(os.path.join(HERE, "flow_ref.npz"), **cases)
print({k: v.shape for k, v in cases.items()})
