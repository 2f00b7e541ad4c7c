"""ctypes binding of the CPU oracle (oracle/vm_oracle.h).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_ODIR = os.path.join(_ROOT, "oracle")
# VM_ORACLE_SO: an alternative build of the oracle (e.g. with -fsanitize=address,undefined)
_SO = os.environ.get("VM_ORACLE_SO") or os.path.join(_ODIR, "_build", "libvm_oracle.so")


def build(force=False):
    srcs = [os.path.join(_ODIR, f) for f in os.listdir(_ODIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if "VM_ORACLE_SO" in os.environ:
        return _SO
    if force or stale:
        subprocess.check_call(["make", "-C", _ODIR], stdout=subprocess.DEVNULL)
    return _SO


class Params(C.Structure):
    _fields_ = [("w_temp", C.c_float), ("w_ui", C.c_float), ("w_tps", C.c_float),
                ("w_ssim", C.c_float), ("ssim_clamp", C.c_float), ("eps", C.c_float),
                ("bcond", C.c_int)]


class Constraint(C.Structure):
    _fields_ = [("lx", C.c_float), ("ly", C.c_float), ("rx", C.c_float),
                ("ry", C.c_float), ("weight", C.c_float)]


def default_params(**kw):
    """Reference defaults, UI/MdiEditor.cpp:131-140."""
    p = Params(w_temp=10.0, w_ui=1e5, w_tps=0.05, w_ssim=100.0, ssim_clamp=0.0,
               eps=0.01, bcond=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def make_constraints(arr):
    arr = np.asarray(arr, dtype=np.float32).reshape(-1, 5)
    cs = (Constraint * max(len(arr), 1))()
    for i, r in enumerate(arr):
        cs[i] = Constraint(*[float(x) for x in r])
    return cs, len(arr)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.vmo_calc_border.restype = C.c_int
    L.vmo_calc_border.argtypes = [C.c_int, C.c_int]
    L.vmo_ssim.restype = C.c_float
    L.vmo_ssim.argtypes = [C.c_float] * 7
    L.vmo_tex2d.restype = C.c_float
    L.vmo_tex2d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
    for f in ("vmo_tps_stencil", "vmo_tps_rows_from_dense", "vmo_io_stencil",
              "vmo_improvmask_stencil"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.vmo_level_create.restype = C.c_void_p
    L.vmo_level_create.argtypes = [C.c_int, C.c_int]
    L.vmo_level_destroy.argtypes = [C.c_void_p]
    L.vmo_level_destroy.restype = None
    L.vmo_level_field.restype = C.c_void_p
    L.vmo_level_field.argtypes = [C.c_void_p, C.c_int]
    L.vmo_init_level.argtypes = [C.c_void_p, C.c_float]
    L.vmo_init_level.restype = None
    L.vmo_splat_constraints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.vmo_splat_constraints.restype = None
    L.vmo_optimize_iter.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p]
    L.vmo_optimize_iter.restype = C.c_int
    L.vmo_optimize_level.argtypes = [C.c_void_p, C.POINTER(Params), C.c_float, C.c_void_p]
    L.vmo_optimize_level.restype = C.c_int
    L.vmo_upsample_v.argtypes = [C.c_void_p, C.c_void_p]
    L.vmo_upsample_v.restype = None
    L.vmo_coarse_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params),
                                   C.c_void_p, C.c_int]
    L.vmo_coarse_solve.restype = C.c_int
    L.vmo_energy.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p]
    L.vmo_energy.restype = None
    L.vmo_render_halfway.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                     C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.vmo_render_halfway.restype = None
    L.vmo_upscale_result.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.vmo_upscale_result.restype = None
    L.vmo_poisson_extend.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
    L.vmo_poisson_extend.restype = C.c_int
    L.vmo_poisson_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p]
    L.vmo_poisson_prepare.restype = C.c_int
    L.vmo_dbg_foldover.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_float, C.c_float]
    L.vmo_dbg_foldover.restype = C.c_float
    L.vmo_dbg_energy_change.argtypes = [C.c_void_p, C.POINTER(Params), C.c_int, C.c_int, C.c_float, C.c_float]
    L.vmo_dbg_energy_change.restype = C.c_float
    L.vmo_quadratic_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                     C.c_void_p, C.POINTER(C.c_double)]
    L.vmo_quadratic_path.restype = C.c_int
    L.vmo_luma_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.vmo_luma_pyramid.restype = None
    L.vmo_set_threads.argtypes = [C.c_int]
    L.vmo_set_threads.restype = None
    L.vmo_get_threads.restype = C.c_int
    L.vmo_set_threads(default_threads())
    _lib = L
    return L


def default_threads():
    """OpenMP threads for the oracle: the CPUs this process may really use (affinity mask capped
    by the cgroup quota -- a 256-thread team on a 16-CPU quota is ~100x slower), at most 8"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except Exception:
        pass
    return max(1, min(n, 8))


_FIELDS = {  # name -> (id, channels, dtype)
    "img0": (0, 1, np.float32), "img1": (1, 1, np.float32), "v": (2, 2, np.float32),
    "luma": (3, 2, np.float32), "mean": (4, 2, np.float32), "var": (5, 2, np.float32),
    "cross": (6, 1, np.float32), "value": (7, 1, np.float32), "counter": (8, 1, np.float32),
    "tps_axy": (9, 1, np.float32), "tps_b": (10, 2, np.float32),
    "ui_axy": (11, 1, np.float32), "ui_b": (12, 2, np.float32),
}


class Level:
    """One page of one pyramid level in the oracle (tight rows)."""

    def __init__(self, w, h):
        self.w, self.h = int(w), int(h)
        self.imp_rs = (self.w + 4) // 5 + 2
        self.imp_rows = (self.h + 4) // 5 + 2
        self._p = lib().vmo_level_create(self.w, self.h)

    def __del__(self):
        try:
            if self._p:
                lib().vmo_level_destroy(self._p)
                self._p = None
        except Exception:
            pass

    def field(self, name):
        """numpy VIEW of a field (h,w) or (h,w,2)."""
        if name == "impmask":
            ptr = lib().vmo_level_field(self._p, 13)
            n = self.imp_rs * self.imp_rows
            buf = (C.c_uint32 * n).from_address(ptr)
            return np.frombuffer(buf, dtype=np.uint32).reshape(self.imp_rows, self.imp_rs)
        fid, ch, dt = _FIELDS[name]
        ptr = lib().vmo_level_field(self._p, fid)
        n = self.w * self.h * ch
        buf = (C.c_float * n).from_address(ptr)
        a = np.frombuffer(buf, dtype=dt)
        return a.reshape(self.h, self.w, 2) if ch == 2 else a.reshape(self.h, self.w)

    def set_images(self, img0, img1):
        self.field("img0")[...] = img0
        self.field("img1")[...] = img1

    def init(self, ssim_clamp=0.0):
        lib().vmo_init_level(self._p, ssim_clamp)

    def splat(self, w0, h0, cons):
        cs, n = make_constraints(cons)
        lib().vmo_splat_constraints(self._p, w0, h0, cs, n)

    def optimize_iter(self, params, stats=None):
        sp = stats.ctypes.data if stats is not None else None
        return lib().vmo_optimize_iter(self._p, C.byref(params), sp)

    def optimize(self, params, max_iter, stats=None):
        sp = stats.ctypes.data if stats is not None else None
        return lib().vmo_optimize_level(self._p, C.byref(params), float(max_iter), sp)

    def upsample_from(self, src):
        lib().vmo_upsample_v(self._p, src._p)

    def coarse_solve(self, w0, h0, params, cons=()):
        cs, n = make_constraints(cons)
        return lib().vmo_coarse_solve(self._p, w0, h0, C.byref(params), cs, n)

    def energy(self, params):
        out = np.zeros(3, dtype=np.float64)
        lib().vmo_energy(self._p, C.byref(params), out.ctypes.data)
        return out


def level_sizes(w, h, nlevels):
    """Level geometry rule of pyramid.cu:466-467 (ceil halving), integer math."""
    out = [(w, h)]
    for _ in range(nlevels - 1):
        w, h = (w + 1) // 2, (h + 1) // 2
        out.append((w, h))
    return out


def solve(pyr_imgs, params, max_iter, drop=1.0, cons=(), stats=None, threads=None,
          per_level=None):
    """Morph::calculate_halfway_parametrization, morph.cu:150-168, on the oracle.

    pyr_imgs: list finest->coarsest of (img0, img1) float32 arrays; the last
    entry is the coarsest level (its images are not used: dense solve only).
    Returns the finest-level Level.
    """
    if threads:
        lib().vmo_set_threads(threads)
    h0, w0 = pyr_imgs[0][0].shape
    sizes = [im[0].shape[::-1] for im in pyr_imgs]
    cur = Level(*sizes[-1])
    cur.coarse_solve(w0, h0, params, cons)
    mi = float(max_iter)
    for el in range(len(pyr_imgs) - 2, -1, -1):
        lvl = Level(*sizes[el])
        lvl.set_images(*pyr_imgs[el])
        lvl.upsample_from(cur)
        lvl.init(params.ssim_clamp)
        lvl.splat(w0, h0, cons)
        it = lvl.optimize(params, mi, stats)
        if per_level is not None:
            per_level.append((sizes[el], it))
        mi = mi / drop
        cur = lvl
    return cur


def render_halfway(w, h, ex, color_fa, geo_fa, color_from, ext0, ext1, v, u):
    out = np.zeros((h, w, 3), dtype=np.uint8)
    ext0 = np.ascontiguousarray(ext0, dtype=np.float32)
    ext1 = np.ascontiguousarray(ext1, dtype=np.float32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    u = np.ascontiguousarray(u, dtype=np.float32)
    lib().vmo_render_halfway(out.ctypes.data, w, h, ex, color_fa, geo_fa, color_from,
                             ext0.ctypes.data, ext1.ctypes.data, v.ctypes.data, u.ctypes.data)
    return out


def upscale_result(v, w0, h0):
    v = np.ascontiguousarray(v, dtype=np.float32)
    h, w = v.shape[:2]
    out = np.zeros((h0, w0, 2), dtype=np.float32)
    lib().vmo_upscale_result(out.ctypes.data, w0, h0, v.ctypes.data, w, h)
    return out


def poisson_extend(ext_rgba, w, h, ex, other_rgba, v, side, tol=1e-8, max_it=20000):
    ext = np.ascontiguousarray(ext_rgba, dtype=np.uint8).copy()
    other = np.ascontiguousarray(other_rgba, dtype=np.uint8)
    v = np.ascontiguousarray(v, dtype=np.float32)
    rr = C.c_double(0)
    it = lib().vmo_poisson_extend(ext.ctypes.data, w, h, ex, other.ctypes.data,
                                  v.ctypes.data, side, tol, max_it, C.byref(rr))
    return ext, it, rr.value


def quadratic_path(v, tol=1e-10, max_it=200000, want_jopt=False):
    """CQuadraticPath::optimize for one frame: (u, iterations, residual[, j_opt])"""
    v = np.ascontiguousarray(v, dtype=np.float32)
    h, w = v.shape[:2]
    u = np.zeros((h, w, 2), dtype=np.float32)
    jo = np.zeros((h, w, 4), dtype=np.float32)
    rr = C.c_double(0)
    it = lib().vmo_quadratic_path(v.ctypes.data, w, h, tol, max_it, u.ctypes.data, jo.ctypes.data, C.byref(rr))
    return (u, it, rr.value, jo) if want_jopt else (u, it, rr.value)


def poisson_prepare(ext_rgba, w, h, ex, other_rgba, v, side):
    ext = np.ascontiguousarray(ext_rgba, dtype=np.uint8).copy()
    other = np.ascontiguousarray(other_rgba, dtype=np.uint8)
    v = np.ascontiguousarray(v, dtype=np.float32)
    typ = np.zeros(((h + 2 * ex), (w + 2 * ex)), dtype=np.int32)
    n = lib().vmo_poisson_prepare(ext.ctypes.data, w, h, ex, other.ctypes.data,
                                  v.ctypes.data, side, typ.ctypes.data)
    return ext, typ, n


def luma_pyramid(rgb, nlevels):
    """list of level lumas (finest first) of the reference's Pyramid::build chain"""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    sizes = level_sizes(w, h, nlevels)
    out = np.zeros(sum(a * b for a, b in sizes), dtype=np.float32)
    lib().vmo_luma_pyramid(rgb.ctypes.data, w, h, nlevels, out.ctypes.data)
    res, off = [], 0
    for (a, b) in sizes:
        res.append(out[off:off + a * b].reshape(b, a))
        off += a * b
    return res
