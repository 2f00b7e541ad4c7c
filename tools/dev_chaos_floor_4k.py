import json, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from videomorphing_amd import capi, morph
import test_gpu_fullsize as T
ctx = morph.Context(0, capi.MATH_EXACT)
for f in (0, 3, 6):
    r = T.chaos_floor_measure(ctx, frames=(f,), w=3840, h=2160)[f]
    print(json.dumps({"frame": f, **T.chaos_round(r)}), flush=True)
