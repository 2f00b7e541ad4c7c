"""Device-side pyramid builder (vm_pyramid.hip) against the outputs of the REFERENCE'S
OWN resampling library (tests/golden/pyramid_ref.npz) and against the oracle."""
import os

import numpy as np
import pytest

from videomorphing_amd import capi, morph, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "pyramid_ref.npz"))

# float tolerance on the [0, 255] luma scale: the device's powf differs from glibc's by
# a few ulp, amplified by the x255 of store_gray
TOL = 2e-3


def _device_lumas(gpu_ctx, rgb0, rgb1, nlevels):
    pyr = morph.Pyramid(gpu_ctx)
    pyr.build_rgb(rgb0, rgb1, 8, nlevels=nlevels + 1)   # + the image-less coarsest level
    return pyr, [(pyr[el].field("img0"), pyr[el].field("img1")) for el in range(1, nlevels + 1)]


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_device_pyramid_matches_reference_library(gpu_ctx, case):
    rgb0, rgb1 = GOLD[case + "0_rgb"], GOLD[case + "1_rgb"]
    nl = int(GOLD[case + "0_nlevels"])
    _, lum = _device_lumas(gpu_ctx, rgb0, rgb1, nl)
    for k, key in enumerate((case + "0_luma", case + "1_luma")):
        got = np.concatenate([l[k].ravel() for l in lum])
        want = GOLD[key]
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= TOL, np.abs(got - want).max()


def test_device_pyramid_matches_oracle_and_feeds_the_solver(gpu_ctx, oracle):
    w, h, nl = 300, 200, 4
    rgb0, rgb1 = synth.make_rgb_pair(w, h)
    pyr, lum = _device_lumas(gpu_ctx, rgb0, rgb1, nl)
    for k, rgb in enumerate((rgb0, rgb1)):
        ref = oracle.luma_pyramid(rgb, nl)
        for el in range(nl):
            assert np.abs(lum[el][k] - ref[el]).max() <= TOL
    gpu_ctx.set_math_mode(capi.MATH_FAST)
    prm = morph.Parameters()
    prm.max_iter, prm.max_iter_drop_factor = 40, 1.0
    m = morph.Morph(prm, pyr)
    assert m.calculate_halfway_parametrization()
    gpu_ctx.set_math_mode(capi.MATH_EXACT)
    v, d = pyr[1].v, synth.displacement(w, h)
    assert np.sqrt(((v - d) ** 2).sum(-1).mean()) < 0.8 * np.sqrt((d ** 2).sum(-1).mean())
