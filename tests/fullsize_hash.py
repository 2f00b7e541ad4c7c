"""SHA-256 fingerprints of a solved level's state arrays: shared by the generator of
tests/golden/full_solve_hashes.json (the ORACLE's whole coarse-to-fine solves at BASELINE.json's
full sizes, run once in the build container) and by the GPU test that compares the HIP path's
EXACT solves with them (tests/test_gpu_fullsize.py::test_full_solve_exact_matches_oracle_hashes).

A fingerprint covers the array's shape and its float / mask words as BITS (no tolerance: -0.0
and +0.0 differ), in the tight (h, w[, 2]) layout both `oracle.Level.field` and
`morph.PyramidLevel.field` return."""
import hashlib

import numpy as np

STATE = ("v", "luma", "mean", "var", "cross", "value", "tps_b", "ui_b", "impmask")


def sha(a):
    a = np.ascontiguousarray(a)
    words = a.view(np.uint32) if a.dtype.itemsize == 4 else a.view(np.uint8)
    h = hashlib.sha256()
    h.update(("%s|%s|" % (a.dtype.str, "x".join(map(str, a.shape)))).encode())
    h.update(words.tobytes())
    return h.hexdigest()


def state_hashes(level):
    """{field: sha256} of every state array of a finest level (oracle.Level or morph.PyramidLevel)"""
    return {f: sha(level.field(f)) for f in STATE}


def input_hash(i0, i1):
    """fingerprint of the synthetic luma pair the solve starts from: numpy's float64 sin / cos are not
    guaranteed to round alike on every host CPU, and a single differing input word changes every output"""
    h = hashlib.sha256()
    h.update(sha(np.asarray(i0, np.float32)).encode())
    h.update(sha(np.asarray(i1, np.float32)).encode())
    return h.hexdigest()
