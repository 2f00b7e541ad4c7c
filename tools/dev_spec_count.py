"""dev: hit rate of the speculative golden-section step (build with -DVM_SPEC_COUNT): one 120x68 level, 40 iterations"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
ctx.set_tuning(0, 0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
i0, i1 = synth.make_pair(1920, 1080)
p = morph.Pyramid(ctx); p.build(i0, i1, 32)
L = p._L
nl = p.size() - 1
prog = (capi.Progress * (nl - 1))()
arr = (C.c_void_p * 1)(p._h)
capi.check(L.vm_solve_batch(arr, 1, float(sys.argv[1]) if len(sys.argv) > 1 else 500.0, 1.0, None, 1, prog))
pr = prog[nl - 2]
print("level %dx%d: candidates %.0f, counted evaluations per candidate %.3f" % (p[nl - 1].width, p[nl - 1].height, pr.candidates, pr.evaluations / pr.candidates))
