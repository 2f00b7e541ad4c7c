"""dev: the EXACT vm_solve against oracle.solve, bit for bit, over whole coarse-to-fine solves that are too long for the test
suite: more frames of config[1] (1920x1080, 6 levels, 500 iterations per level, the reference's stopping rule) -- among them
frames whose finest level keeps exchanging moves until iteration 500 -- and config[3] (3840x2160, 7 levels) itself.
Per solve: per-level iteration counts equal, every state array of the finest level equal as bits.
usage: tools/dev_parity_full.py [--4k] [frame ...]          (one JSON line per solve; profiles/r04_parity_full.jsonl)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph, synth  # noqa: E402
import oracle  # noqa: E402  (tests/oracle.py: the ctypes binding of the checker)

STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")
args = sys.argv[1:]
four_k = "--4k" in args
frames = [int(a) for a in args if not a.startswith("--")] or [1, 2, 6, 9]
w, h, nlev = (3840, 2160, 7) if four_k else (1920, 1080, 6)


def all_cpus():
    """host threads for the oracle: what this process may really use (cgroup quota respected; 256 threads on a 16-core
    quota made the first run of this tool 15x slower than it had to be)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, n)


threads = all_cpus()
ctx = morph.Context(0, capi.MATH_EXACT)
prm = morph.Parameters()
prm.max_iter, prm.max_iter_drop_factor, prm.start_res = 500, 1.0, 32
ctx.set_params(morph.KernParameters(prm))
for f in frames:
    i0, i1 = synth.make_pair(w, h, frame=f)
    t0 = time.time()
    per = []
    lo = oracle.solve(synth.build_pyramid(i0, i1, nlev), oracle.default_params(), 500, 1.0, threads=threads, per_level=per)
    t_cpu = time.time() - t0
    pyr = morph.Pyramid(ctx)
    pyr.build(i0, i1, 32)
    assert pyr.size() == nlev + 1, pyr.size()
    prog = (capi.Progress * (nlev - 1))()
    t0 = time.time()
    capi.check(pyr._L.vm_solve(pyr._h, 500.0, 1.0, None, 0, None, 0, prog))
    t_gpu = time.time() - t0
    its_gpu = [int(prog[el].iters) for el in range(nlev - 2, -1, -1)]
    its_cpu = [int(p[1]) for p in per]
    lv = pyr[1]
    diff = {n: int((lo.field(n).view(np.uint32) != lv.field(n).view(np.uint32)).sum()) for n in STATE}
    print(json.dumps({"size": [w, h], "levels": nlev, "frame": f, "iters_oracle": its_cpu, "iters_gpu": its_gpu,
                      "words_differing": diff, "bit_identical": its_cpu == its_gpu and not any(diff.values()),
                      "max_abs_v": float(np.abs(lv.v).max()), "oracle_s": round(t_cpu, 1), "oracle_threads": threads,
                      "gpu_exact_s": round(t_gpu, 3)}), flush=True)
    pyr.clear()
    del pyr, lo
