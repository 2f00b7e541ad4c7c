"""dev: time of the device-side flow pyramid of a 5-frame 1080p video (vm_video_build_flows)"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from videomorphing_amd import capi, morph, synth
ctx = morph.Context(0, capi.MATH_FAST)
w, h, d = 1920, 1080, 5
levels, ft = synth.video_levels(w, h, d, 32)
vid = morph.VideoPyramid(ctx)
vid.build_levels(levels, ft, d)
fl = synth.constant_flows(w, h, d)
for rep in range(3):
    ctx.sync(); t = time.perf_counter()
    vid.build_flows(*fl)
    ctx.sync(); print("flow pyramid: %.1f ms" % ((time.perf_counter() - t) * 1e3))
