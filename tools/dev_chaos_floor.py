"""dev: the per-frame chaos floor of config[1] (1080p, 6 levels, 500 iterations per level, reference
stopping rule): the family of equally legal runs (four commit orders in EXACT arithmetic, two of them with fused
multiply-adds, tests/test_gpu_fullsize.py::CHAOS_FAMILY) and FAST under the automatic schedule, per frame; energies from the oracle's vmo_energy on the host.
The measurement is tests/test_gpu_fullsize.py::chaos_floor_measure; this prints it as JSON lines.
Round 5: the family holds two runs with CUDA's 8-bit texture filter weights (VM_MATH_REF_TEX8); --trunc adds one
with truncated weights (member "u0": sensitivity to the rounding rule the CUDA guide leaves open).
usage: tools/dev_chaos_floor.py [--4k] [--trunc] [frame ...]      (--4k: config[3], 3840x2160, 7 levels)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph  # noqa: E402
import test_gpu_fullsize as T  # noqa: E402

big = "--4k" in sys.argv
trunc = "--trunc" in sys.argv
family = T.CHAOS_FAMILY + ((("u0", capi.MATH_REF_TEX8_TRUNC, 0),) if trunc else ())
frames = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or (list(T.CHAOS_FRAMES) if not big else [0, 3, 6])
size = dict(w=3840, h=2160) if big else {}
ctx = morph.Context(0, capi.MATH_EXACT)
signed = []
for f in frames:
    r = T.chaos_floor_measure(ctx, frames=(f,), family=family, **size)[f]
    signed.append(r["e_fast_signed"])
    print(json.dumps({"frame": f, **T.chaos_round(r)}), flush=True)
s = np.array(signed)
print(json.dumps({"signed_energy_mean": float(s.mean()), "sem": float(s.std(ddof=1) / np.sqrt(len(s))) if len(s) > 1 else None}))
