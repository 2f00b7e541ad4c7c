"""numpy model of the device's multigrid-preconditioned CG (vm_mgb.hip) on the Poisson extension's system of one side of a
fixture frame: iteration counts of smoother / cycle variants BEFORE any kernel is written (build container, CPU only).

The hierarchy is the device's: 2x2 aggregation, piecewise-constant transfer, Galerkin operator with the edge weights
halved, down to a grid of <= 64 cells that gets 2 symmetric sweeps each way; the smoother is red-black Gauss-Seidel,
`nu` full sweeps before (red, black) and after (black, red) the coarse correction, per level.  PCG in float64 on
channel 0, 1 and 2 (the worst channel counts, as on the device).

usage: python tools/exp/mg_prototype.py [--size 1920x1080] [--ex 192] [--frame 0] [--side 1] [--nu 1,1,1,...;2,2,2,...]
  --nu  one comma list per variant, separated by ';': sweeps per level from level 0 down, the last entry repeats
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import fullsize_fixture as FX  # noqa: E402
from videomorphing_amd import synth  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def make_extended(rgb, ex):
    h, w = rgb.shape[:2]
    out = np.full((h + 2 * ex, w + 2 * ex, 4), 255, np.uint8)
    out[ex:ex + h, ex:ex + w, :3] = rgb
    out[ex:ex + h, ex:ex + w, 3] = 0
    return out


def system(w, h, ex, frame, side):
    """type map, diagonal and right-hand side of oracle/vm_oracle_poisson.c (PoissonExt.cpp:146-312), vectorised"""
    rgb0, rgb1 = synth.make_rgb_pair(w, h, frame=frame)
    v = FX.field(w, h, frame)
    e = [make_extended(rgb0, ex), make_extended(rgb1, ex)]
    other = e[2 - side][ex:ex + h, ex:ex + w].copy()
    ext, typ, _ = oracle.poisson_prepare(e[side - 1], w, h, ex, other, v, side)
    col = ext[..., :3].astype(np.float64)
    marker = (ext[..., 0] == 255) & (ext[..., 1] == 0) & (ext[..., 2] == 255) & (ext[..., 3] == 0)
    t2 = (typ > 1) & ~marker
    gx = np.zeros(col.shape)
    gy = np.zeros(col.shape)
    ok = t2[:, 1:] & t2[:, :-1]
    gx[:, 1:][ok] = (col[:, 1:] - col[:, :-1])[ok]
    ok = t2[1:] & t2[:-1]
    gy[1:][ok] = (col[1:] - col[:-1])[ok]
    unk = typ > 0
    E = unk[:, :-1] & unk[:, 1:]            # edge (x, y) - (x + 1, y)
    S = unk[:-1] & unk[1:]
    B = np.zeros(col.shape)
    B[typ == 1] += col[typ == 1]
    B[1:][S] += gy[1:][S]                   # north neighbour present
    B[:, 1:][E] += gx[:, 1:][E]             # west
    B[:, :-1][E] -= gx[:, 1:][E]            # east
    B[:-1][S] -= gy[1:][S]                  # south
    B[~unk] = 0
    tie = (typ == 1).astype(np.float64)
    return unk, E.astype(np.float64), S.astype(np.float64), tie, B


class Level:
    def __init__(self, we, ws, sc):
        # we[y, x]: weight of the edge to the east (shape h x (w - 1)), ws: to the south ((h - 1) x w), sc: screening
        self.h, self.w = sc.shape
        self.we, self.ws = we, ws
        dg = np.maximum(sc, 0).copy()
        dg[:, :-1] += we
        dg[:, 1:] += we
        dg[:-1] += ws
        dg[1:] += ws
        self.dg = dg
        self.sc = sc
        self.unk = dg > 0
        self.inv = np.where(self.unk, 1.0 / np.where(self.unk, dg, 1), 0.0)
        yy, xx = np.mgrid[0:self.h, 0:self.w]
        self.red = ((xx + yy) & 1) == 0

    def nbsum(self, x):
        s = np.zeros_like(x)
        s[:, :-1] += self.we[..., None] * x[:, 1:]
        s[:, 1:] += self.we[..., None] * x[:, :-1]
        s[:-1] += self.ws[..., None] * x[1:]
        s[1:] += self.ws[..., None] * x[:-1]
        return s

    def apply(self, x):
        return self.dg[..., None] * x - self.nbsum(x)

    def half(self, x, b, red):
        m = (self.red == red) & self.unk
        s = self.nbsum(x)
        x[m] = (self.inv[..., None] * (b + s))[m]

    def coarsen(self):
        h2, w2 = (self.h + 1) // 2, (self.w + 1) // 2

        def pad(a, hh, ww):
            out = np.zeros((hh, ww))
            out[:a.shape[0], :a.shape[1]] = a
            return out
        # edges leaving an aggregate to the east: fine edges at odd x; to the south: odd y; x 1/2
        wef = pad(self.we, 2 * h2, 2 * w2)
        wsf = pad(self.ws, 2 * h2, 2 * w2)
        we = 0.5 * (wef[0::2, 1::2] + wef[1::2, 1::2])[:, :w2 - 1]
        ws = 0.5 * (wsf[1::2, 0::2] + wsf[1::2, 1::2])[:h2 - 1]
        scf = pad(self.sc, 2 * h2, 2 * w2)
        sc = scf[0::2, 0::2] + scf[0::2, 1::2] + scf[1::2, 0::2] + scf[1::2, 1::2]
        return Level(we, ws, sc)


def restrict(r):
    h, w = r.shape[:2]
    h2, w2 = (h + 1) // 2, (w + 1) // 2
    p = np.zeros((2 * h2, 2 * w2, r.shape[2]))
    p[:h, :w] = r
    return p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]


def prolong(xc, h, w):
    return np.repeat(np.repeat(xc, 2, axis=0), 2, axis=1)[:h, :w]


def vcycle(levels, l, b, nu, coarsest_sweeps=2):
    L = levels[l]
    x = np.zeros_like(b)
    if l == len(levels) - 1:
        for _ in range(coarsest_sweeps):
            L.half(x, b, True)
            L.half(x, b, False)
        for _ in range(coarsest_sweeps):
            L.half(x, b, False)
            L.half(x, b, True)
        return x
    n = nu[min(l, len(nu) - 1)]
    for _ in range(n):
        L.half(x, b, True)
        L.half(x, b, False)
    r = b - L.apply(x)
    r[~L.unk] = 0
    xc = vcycle(levels, l + 1, restrict(r), nu, coarsest_sweeps)
    x += prolong(xc, L.h, L.w)
    x[~L.unk] = 0
    for _ in range(n):
        L.half(x, b, False)
        L.half(x, b, True)
    return x


def pcg(levels, B, nu, tols, max_it=60):
    L = levels[0]
    x = np.zeros_like(B)
    r = B.copy()
    bb = (B * B).sum(axis=(0, 1))
    out = {}
    p = None
    rz_old = None
    for it in range(max_it + 1):
        rel = float(np.sqrt(((r * r).sum(axis=(0, 1)) / bb).max()))
        for t in tols:
            if t not in out and rel <= t:
                out[t] = it
        if len(out) == len(tols):
            break
        z = vcycle(levels, 0, r, nu)
        rz = (r * z).sum(axis=(0, 1))
        p = z if p is None else z + (rz / rz_old) * p
        rz_old = rz
        q = L.apply(p)
        al = rz / (p * q).sum(axis=(0, 1))
        x += al * p
        r -= al * q
    return [out.get(t, -1) for t in tols]


def main():
    w, h = (int(t) for t in arg("--size", "1920x1080").split("x"))
    ex = int(arg("--ex", "192"))
    frame, side = int(arg("--frame", "0")), int(arg("--side", "1"))
    variants = [[int(t) for t in v.split(",")] for v in arg("--nu", "1;2").split(";")]
    t0 = time.time()
    unk, E, S, tie, B = system(w, h, ex, frame, side)
    levels = [Level(E, S, tie)]
    while levels[-1].w * levels[-1].h > 64:
        levels.append(levels[-1].coarsen())
    print("canvas %dx%d, %d unknowns, %d levels (coarsest %dx%d), set-up %.0f s" % (
        w + 2 * ex, h + 2 * ex, int(unk.sum()), len(levels), levels[-1].w, levels[-1].h, time.time() - t0), flush=True)
    tols = (1e-4, 1e-5, 1e-6)
    for nu in variants:
        t0 = time.time()
        its = pcg(levels, B, nu, tols)
        print("nu per level %-24s PCG iterations to 1e-4 / 1e-5 / 1e-6: %s   (%.0f s)" % (",".join(map(str, nu)), " / ".join(map(str, its)), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
