"""dev: stage stamps of one dense TILE launch (k_optimize<true, 13, 2>) over a batch of 120x68 levels,
every pixel a candidate -- the launch the 60-pair job spends its time in (needs a build with
VM_DEFS=-DVM_PROF: thread 0 of each tile of the first pair stamps its phases)
   python tools/dev_prof_tile.py [pairs=30] [w=120] [h=68] [iters=3] [lean]
with `lean`: the TILE schedule as AUTO would run it (the lean kernel once the level is pruned): stamps of
the tiles that were still active in the last pass"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LEAN = "lean" in sys.argv
if not LEAN:
    os.environ["VM_TILE_DENSE"] = "1"
os.environ["VM_NO_GRAPH"] = "1"
from videomorphing_amd import capi, morph, synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (120, 68)
iters = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 3
ctx = morph.Context(0, capi.MATH_FAST)
ctx.set_params(morph.KernParameters(morph.Parameters()))
ctx.set_tuning(capi.SWEEP_TILE, 0, 0)
frames = [synth.make_pair(w, h, frame=k) for k in range(min(npairs, 4))]
v0 = (0.9 * synth.displacement(w, h) + 0.05 * np.random.RandomState(1).randn(h, w, 2)).astype(np.float32)
batch = []
for k in range(npairs):
    pyr = morph.Pyramid(ctx); pyr.build_levels([(w, h), ((w + 1) // 2, (h + 1) // 2)])
    pyr.upload_luma(1, *frames[k % len(frames)]); pyr[1].v = v0
    batch.append(pyr)
arr = (C.c_void_p * npairs)(*[p._h for p in batch])
L = batch[0]._L
for rep in range(2):
    for p in batch:
        capi.check(L.vm_init_level(p._h, 0, w, h, None, 0))
    prog = (capi.Progress * npairs)()
    capi.check(L.vm_optimize_level_batch(arr, npairs, 0, float(iters), None, 1, prog))
print("pairs %d %dx%d: %.1f us per pass launch, sched ms %s, line searches per iteration and pair %.0f" % (
    npairs, w, h, prog[0].elapsed_ms * 1e3 / (4 * iters), [round(x, 2) for x in prog[0].sched_ms], prog[0].candidates / iters))
buf = np.zeros(512 * 16 * 2, np.uint64)
L.vm_dbg_prof_read.argtypes = [C.c_void_p, C.c_size_t]
assert L.vm_dbg_prof_read(buf.ctypes.data, buf.nbytes) == 0
ntile = ((w + 68) // 69) * ((h + 20) // 21)
st = buf[:8192].reshape(256, 4, 8).astype(np.int64)[:ntile]
sf = buf[8192 + 512:8192 + 512 + 2048].reshape(256, 8).astype(np.int64)[:ntile]
ok = (sf[:, 3] > sf[:, 0]) & (sf[:, 0] >= sf[:, 0].max() - 2000000)   # this launch's (20 ms window)
st, sf = st[ok], sf[ok]
us = lambda d: "mean %6.2f  min %6.2f  max %6.2f" % (d.mean() / 100.0, d.min() / 100.0, d.max() / 100.0)
print("tiles stamped (last pass of the last iteration):", ok.sum())
print("entry -> state in LDS (mask words, early out, tables, LoadSSIM) :", us(sf[:, 1] - sf[:, 0]))
for ph in range(4):
    a = st[:, ph]
    print("phase %d: candidates + compaction %s" % (ph, us(a[:, 1] - a[:, 0])))
    print("         line searches (wave 0)   %s" % us(a[:, 2] - a[:, 1]))
    print("         wait for the slowest wave %s" % us(a[:, 3] - a[:, 2]))
    print("         own commits + count       %s" % us(a[:, 4] - a[:, 3]))
    print("         gather into the cells     %s" % us(a[:, 5] - a[:, 4]))
    print("         phase total               %s" % us(a[:, 5] - a[:, 0]))
print("SaveSSIM + mask words:", us(sf[:, 3] - sf[:, 2]))
print("tile total:", us(sf[:, 3] - sf[:, 0]))
