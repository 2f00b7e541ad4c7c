"""The pyramid-builder row is pinned by outputs of the REFERENCE ITSELF: the golden
fixture tests/golden/pyramid_ref.npz was produced by the reference's own resampling
library (include/resample, built from its sources by `make -C oracle ref`; generator:
tests/golden/make_pyramid_golden.py).  CPU only."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "pyramid_ref.npz"))
CASES = sorted(k[:-4] for k in GOLD.files if k.endswith("_rgb"))


@pytest.mark.parametrize("case", CASES)
def test_oracle_pyramid_matches_reference_library(oracle, case):
    rgb, nl, want = GOLD[case + "_rgb"], int(GOLD[case + "_nlevels"]), GOLD[case + "_luma"]
    got = np.concatenate([l.ravel() for l in oracle.luma_pyramid(rgb, nl)])
    assert got.shape == want.shape
    d = np.abs(got - want)
    # same float operations in the same order: expected to agree to the last bits
    # (libm powf may differ by an ulp between builds); luma scale is [0, 255]
    assert d.max() <= 2e-4, d.max()
    assert (d == 0).mean() > 0.5


def test_level1_is_the_plain_luma(oracle):
    """el == 0 is a same-size scale(): sRGB round trip + prefilter/reconstruct = identity
    up to round-off, so level 1 is .299 R + .587 G + .114 B of the frame"""
    rgb = GOLD["a0_rgb"].astype(np.float64)
    h, w = rgb.shape[:2]
    lum = 0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]
    assert np.abs(GOLD["a0_luma"][:w * h].reshape(h, w) - lum).max() < 1e-3
