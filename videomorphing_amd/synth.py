"""Synthetic frame pairs for benchmarks and tests.

img0 = 6-octave value noise scaled to luma [16, 240];
img1(p) = img0(p - 2 d(p)) with the smooth ground-truth halfway displacement
d(p) = A (sin(2 pi x/W) cos(2 pi y/H), sin(2 pi y/H)), A = 0.01 W;
video frame k translates both images by (0.5 k, 0.25 k) pixels.

The noise is an analytic function of continuous coordinates, so img1 is an
exact warp (no resampling of img0).  Host-side numpy: input generation is not
part of the hot path.
"""
import numpy as np

SEED = 1234
OCTAVES = 6


def _lattice(seed, octave, n):
    rng = np.random.RandomState((seed * 1000003 + octave * 7919) % (2 ** 31 - 1))
    return rng.rand(n, n).astype(np.float64)


def value_noise(x, y, base, seed=SEED, octaves=OCTAVES):
    """Sum of `octaves` bilinear value-noise octaves at continuous (x, y)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    out = np.zeros(np.broadcast(x, y).shape, dtype=np.float64)
    n = 64
    for k in range(octaves):
        lat = _lattice(seed, k, n)
        s = base / (2.0 ** k)
        fx, fy = x / s, y / s
        ix, iy = np.floor(fx), np.floor(fy)
        ax, ay = fx - ix, fy - iy
        ix = ix.astype(np.int64)
        iy = iy.astype(np.int64)
        i0, i1 = np.mod(ix, n), np.mod(ix + 1, n)
        j0, j1 = np.mod(iy, n), np.mod(iy + 1, n)
        val = ((1 - ax) * (1 - ay) * lat[j0, i0] + ax * (1 - ay) * lat[j0, i1] +
               (1 - ax) * ay * lat[j1, i0] + ax * ay * lat[j1, i1])
        out += val * (0.5 ** k)
    return out


def displacement(w, h, amp=None):
    """Ground-truth halfway displacement d(p) on the pixel grid, shape (h, w, 2)."""
    amp = 0.01 * w if amp is None else amp
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    dx = amp * np.sin(2 * np.pi * x / w) * np.cos(2 * np.pi * y / h)
    dy = amp * np.sin(2 * np.pi * y / h)
    return np.stack([dx, dy], axis=-1)


def make_pair(w, h, frame=0, seed=SEED, amp=None):
    """Returns (img0, img1) float32 luma in [16, 240], shape (h, w)."""
    base = max(w, h) / 8.0
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    tx, ty = 0.5 * frame, 0.25 * frame
    d = displacement(w, h, amp)
    n0 = value_noise(x - tx, y - ty, base, seed)
    n1 = value_noise(x - 2 * d[..., 0] - tx, y - 2 * d[..., 1] - ty, base, seed)
    # fixed affine map of the noise range [0, 2 - 2^-(octaves-1)] to [16, 240]
    top = 2.0 - 0.5 ** (OCTAVES - 1)
    lo, hi = 0.25 * top, 0.75 * top
    f = lambda n: np.clip(16.0 + (n - lo) / (hi - lo) * 224.0, 16.0, 240.0)
    return f(n0).astype(np.float32), f(n1).astype(np.float32)


def make_rgb_pair(w, h, frame=0, seed=SEED, amp=None):
    """RGB8 versions for the compositor: grey luma + two low-frequency chroma fields."""
    i0, i1 = make_pair(w, h, frame, seed, amp)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    cr = 24.0 * (value_noise(x, y, max(w, h) / 2.0, seed + 1, 2) - 0.75)
    cb = 24.0 * (value_noise(x, y, max(w, h) / 2.0, seed + 2, 2) - 0.75)
    out = []
    for im in (i0, i1):
        rgb = np.stack([im + cr, im, im + cb], axis=-1)
        out.append(np.clip(np.rint(rgb), 0, 255).astype(np.uint8))
    return out[0], out[1]


def downsample2(img):
    """ceil(dim/2) box downsample (edge-replicated when a dimension is odd)."""
    h, w = img.shape
    if h % 2:
        img = np.concatenate([img, img[-1:, :]], axis=0)
    if w % 2:
        img = np.concatenate([img, img[:, -1:]], axis=1)
    return (0.25 * (img[0::2, 0::2] + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2])
            ).astype(np.float32)


def num_levels(w, h, start_res):
    """Level count of Pyramid::build (pyramid.cu:230-240) in integer arithmetic:
    el = floor(log2(dim) - log2(start_res) + 1), el_x = el_y = max; the count
    includes the coarsest (CPU-solved) level."""
    def el(dim):
        n = 1
        while dim // 2 >= start_res:
            dim //= 2
            n += 1
        return n
    return max(el(w), el(h))


def build_pyramid(img0, img1, nlevels):
    """List finest->coarsest of (img0, img1); each level ceil-halves (pyramid.cu:466-467)."""
    pyr = [(np.ascontiguousarray(img0, dtype=np.float32),
            np.ascontiguousarray(img1, dtype=np.float32))]
    for _ in range(nlevels - 1):
        a, b = pyr[-1]
        pyr.append((downsample2(a), downsample2(b)))
    return pyr


def make_constraints(w, h, n=8, amp=None):
    """n point pairs (lx, ly, rx, ry, weight): lp on a lattice, rp = lp + 2 d(lp) rounded."""
    amp = 0.01 * w if amp is None else amp
    pts = []
    cols = max(1, int(np.ceil(np.sqrt(n * w / float(h)))))
    rows = int(np.ceil(n / float(cols)))
    for k in range(n):
        cx = (k % cols + 0.5) / cols * w
        cy = (k // cols + 0.5) / rows * h
        # img1(r) = img0(r - 2 d(r)): the exact correspondence of right point r
        rx, ry = np.rint(cx), np.rint(cy)
        dx = amp * np.sin(2 * np.pi * rx / w) * np.cos(2 * np.pi * ry / h)
        dy = amp * np.sin(2 * np.pi * ry / h)
        lx, ly = np.rint(rx - 2 * dx), np.rint(ry - 2 * dy)
        pts.append((lx, ly, rx, ry, 1.0))
    return np.asarray(pts, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# video pairs with analytic optical flow (the temporal coherence path)

def make_video_pair(w, h, frame, shift0=(0.5, 0.25), shift1=None, seed=SEED, amp=None):
    """Frame `frame` of two synthetic videos: video 0 translates by shift0 px per frame, video 1
    by shift1 (default: the same).  Both are exact warps of the analytic noise field, so the
    optical flow is known in closed form: f0 = shift0, f1 = shift1 everywhere."""
    shift1 = shift0 if shift1 is None else shift1
    base = max(w, h) / 8.0
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    d = displacement(w, h, amp)
    n0 = value_noise(x - shift0[0] * frame, y - shift0[1] * frame, base, seed)
    x1, y1 = x - shift1[0] * frame, y - shift1[1] * frame          # undo video 1's motion ...
    dd = displacement_at(x1, y1, w, h, amp)                          # ... then its warp of frame 0
    n1 = value_noise(x1 - 2 * dd[..., 0], y1 - 2 * dd[..., 1], base, seed)
    top = 2.0 - 0.5 ** (OCTAVES - 1)
    lo, hi = 0.25 * top, 0.75 * top
    f = lambda n: np.clip(16.0 + (n - lo) / (hi - lo) * 224.0, 16.0, 240.0)
    return f(n0).astype(np.float32), f(n1).astype(np.float32)


def displacement_at(x, y, w, h, amp=None):
    """d(p) at continuous coordinates (same field as displacement())"""
    amp = 0.01 * w if amp is None else amp
    dx = amp * np.sin(2 * np.pi * x / w) * np.cos(2 * np.pi * y / h)
    dy = amp * np.sin(2 * np.pi * y / h)
    return np.stack([dx, dy], axis=-1)


def constant_flows(w, h, depth, shift0=(0.5, 0.25), shift1=None):
    """(f0, f1, b0, b1): per frame (h, w, 2) forward flows of video 0 / 1 (frame t -> t+1) and
    backward flows (t -> t-1) of the translating synthetic videos"""
    shift1 = shift0 if shift1 is None else shift1
    mk = lambda s: [np.broadcast_to(np.asarray(s, np.float32), (h, w, 2)).copy() for _ in range(depth)]
    return mk(shift0), mk(shift1), mk((-shift0[0], -shift0[1])), mk((-shift1[0], -shift1[1]))


def video_levels(w, h, d, start_res, max_voxels=14e6):
    """Level table of the stage-2 Pyramid::build (pyramid.cu:223-240, 462-477): returns
    (levels, factor_t); levels = [(w, h, d), ...] finest first incl. the coarsest (host-solved)
    level, factor_t[l] = temporal stride level l is built with.  Level counts in integer
    arithmetic (the reference truncates float32 logarithms: el = int(log2(dim) - log2(start_res)
    + 1), exposed to 1-ulp hazards at exact powers of two); the 14 Mvoxel decimation in float32
    as written there."""
    f32 = np.float32
    fa = max(np.sqrt(f32(w * h * d) / f32(max_voxels)), f32(1))
    w, h = int(f32(w) / fa), int(f32(h) / fa)

    def el(dim):      # largest n with start_res * 2^(n-1) <= dim; 0 below start_res
        n = 0
        while dim >= start_res:
            dim //= 2
            n += 1
        return n
    el_t = el(d)
    el_x = el_y = max(el(w), el(h))
    maxl = max(el_x, el_t)
    levels, factors, factor_t = [], [], 1
    for k in range(maxl):
        levels.append((w, h, d))
        factors.append(factor_t)
        if maxl - k <= el_x:
            w = (w + 1) // 2
        if maxl - k <= el_y:
            h = (h + 1) // 2
        if maxl - k <= el_t:
            d, factor_t = (d + 2) // 2, 2            # ceil((d + 1) / 2)
        else:
            factor_t = 1
    return levels, factors


def page_frames(levels, factors):
    """frame of the video each page shows: page t of level l is scaled from page
    min(t * factor_t, prev_d - 1) of level l-1 (pyramid.cu:363-364)"""
    out = [list(range(levels[0][2]))]
    for l in range(1, len(levels)):
        pd = levels[l - 1][2]
        out.append([out[l - 1][min(t * factors[l], pd - 1)] for t in range(levels[l][2])])
    return out
