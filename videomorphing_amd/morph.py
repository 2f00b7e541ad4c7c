"""Host-side mirror of the reference's L3 solver API over the C-ABI.

Same names, argument meaning and progress/cancellation behaviour as
Algorithm/parameters.h (Parameters, KernParameters, Conp, Connect,
BoundaryCondition), Algorithm/Pyramid.h (Pyramid, PyramidLevel),
Algorithm/morph.h (Morph) and Algorithm/MatchingThread.h (CMatchingThread), so
that tests read like drivers of the reference.  This module holds no
numerics: everything runs in libvmorph_hip.so.  (The C++ facade with the same
shape is include/vmorph/*.hpp.)

Scope: one frame pair per Pyramid (depth 1, the independent-pair formulation).
"""
import ctypes as C
import os
import threading
import time
import weakref

import numpy as np

from . import capi
from .capi import BCOND_NONE, BCOND_CORNER, BCOND_BORDER, MATH_EXACT, MATH_FAST  # noqa: F401
from . import synth


class Conp(object):
    """struct Conp, parameters.h:22-26: p = (x, y, frame, is_key), weight."""

    def __init__(self, x, y, z=0, w=1, weight=1.0):
        self.p = (int(x), int(y), int(z), int(w))
        self.weight = float(weight)


class Connect(object):
    """struct Connect, parameters.h:16-20: li/ri = (track, frame) indices."""

    def __init__(self, li, ri):
        self.li = tuple(li)
        self.ri = tuple(ri)


class Parameters(object):
    """struct Parameters, parameters.h:29-52, defaults of UI/MdiEditor.cpp:131-140."""

    def __init__(self):
        self.w_ui, self.w_tps, self.w_ssim, self.w_temp = 1e5, 0.05, 100.0, 10.0
        self.ssim_clamp = 0.0
        self.eps = 0.01
        self.max_iter = 1000
        self.start_res = 8
        self.max_iter_drop_factor = 2.0
        self.bcond = BCOND_NONE
        self.lp, self.rp, self.cnt = [], [], []
        self.verbose = False

    def add_point_pair(self, lx, ly, rx, ry, weight=1.0, frame=0):
        """Convenience: one key-point track per side plus its connection."""
        self.lp.append([Conp(lx, ly, frame, 1, weight)])
        self.rp.append([Conp(rx, ry, frame, 1, weight)])
        k = len(self.lp) - 1
        self.cnt.append([Connect((k, 0), (k, 0))])

    def constraints(self, conz=0):
        """Resolve lp/rp/cnt as morph.cu:354-366 does for page `conz`."""
        out = []
        for row in self.cnt:
            for c in row:
                l = self.lp[c.li[0]][c.li[1]]
                r = self.rp[c.ri[0]][c.ri[1]]
                if l.p[2] != conz:
                    continue
                out.append((l.p[0], l.p[1], r.p[0], r.p[1], min(l.weight, r.weight)))
        return np.asarray(out, dtype=np.float32).reshape(-1, 5)


class KernParameters(capi.KernParams):
    """struct KernParameters(const Parameters&), parameters.h:54-72."""

    def __init__(self, p=None):
        capi.KernParams.__init__(self)
        if p is not None:
            self.w_temp, self.w_ui, self.w_tps, self.w_ssim = p.w_temp, p.w_ui, p.w_tps, p.w_ssim
            self.ssim_clamp, self.eps, self.bcond = p.ssim_clamp, p.eps, int(p.bcond)


class Context(object):
    """One HIP device + stream (vm_ctx)."""

    def __init__(self, device=0, math_mode=MATH_EXACT):
        self._L = capi.load()
        h = C.c_void_p()
        capi.check(self._L.vm_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.set_math_mode(math_mode)

    def close(self):
        if getattr(self, "_h", None):
            self._L.vm_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def set_params(self, kp):
        capi.check(self._L.vm_set_params(self._h, C.byref(kp)))

    def set_math_mode(self, mode):
        capi.check(self._L.vm_set_math_mode(self._h, int(mode)))
        self.math_mode = int(mode)

    def set_tuning(self, sweep_mode=0, threads=0, parts=0):
        capi.check(self._L.vm_set_tuning(self._h, int(sweep_mode), int(threads), int(parts)))

    def runs_beside(self, other):
        """vm_dbg_streams_overlap: do this context's stream and `other`'s run their kernels side by side (True) or
        did the runtime put them on one hardware queue (False)?"""
        ov = C.c_int(0)
        capi.check(self._L.vm_dbg_streams_overlap(self._h, other._h, C.byref(ov)))
        return bool(ov.value)

    def poisson_profile(self, on):
        """vm_dbg_poisson_profile: arm (on = True) the HIP-event probe around the linear solver's dominant kernel (the launch
        that carries the PCG update), or disarm it and return (summed us, launches, active systems summed over the launches,
        launches of the form with the update fused into the level-0 restriction)"""
        us, n, act, fused = C.c_double(0), C.c_int(0), C.c_double(0), C.c_int(0)
        capi.check(self._L.vm_dbg_poisson_profile(self._h, int(bool(on)), C.byref(us), C.byref(n), C.byref(act), C.byref(fused)))
        return us.value, n.value, act.value, fused.value

    def set_commit_order(self, order=0):
        """diagnostic (EXACT): the order a phase's commits are folded in -- 0 row-major (the oracle's),
        1 reversed, 2 column-major, 3 column-major reversed (vm_set_commit_order)"""
        capi.check(self._L.vm_set_commit_order(self._h, int(order)))

    def set_sparse_resident(self, mode=0):
        """test hook of the SPARSE schedule's resident visits (FAST): 0 automatic, 1 never, 2 re-centre the LDS copy
        after every commit, 3 give residency up at the first commit (vm_dbg_sparse_resident)"""
        capi.check(self._L.vm_dbg_sparse_resident(self._h, int(mode)))

    def sparse_resident_visits(self):
        """tile visits of this context's solves served from the resident LDS copy so far (vm_dbg_sparse_resident_visits)"""
        return int(self._L.vm_dbg_sparse_resident_visits(self._h))

    def sync(self):
        capi.check(self._L.vm_ctx_sync(self._h))

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_uint64(0)
        capi.check(self._L.vm_device_info(self._h, name, C.byref(cus), C.byref(mem)))
        return name.value.decode(), cus.value, mem.value


def _cons_array(cons):
    cons = np.asarray(cons if cons is not None else [], dtype=np.float32).reshape(-1, 5)
    n = len(cons)
    arr = (capi.Constraint * max(n, 1))()
    for k in range(n):
        arr[k] = capi.Constraint(*[float(x) for x in cons[k]])
    return arr, n


class PyramidLevel(object):
    """struct PyramidLevel, Pyramid.h:52-98 (geometry + device-state access)."""

    def __init__(self, pyr, el, w, h):
        # a weak back-reference: no Pyramid <-> PyramidLevel cycle, so a pyramid is released by
        # reference counting the moment its last user lets go -- before its Context, which it
        # keeps alive -- and never by the cycle collector in an arbitrary order
        self._pyr_ref, self._el = weakref.ref(pyr), el
        self.width, self.height, self.depth = int(w), int(h), 1
        self.rowstride = (self.width + 31) // 32 * 32         # pyramid.cu:535
        self.pagestride = self.rowstride * self.height
        self.inv_wh = np.float32(1.0) / np.float32(self.width * self.height)
        self.impmask_rowstride = (self.width + 4) // 5 + 2
        self.impmask_pagestride = self.impmask_rowstride * ((self.height + 4) // 5 + 2)
        self.factor_d = 1.0

    @property
    def _pyr(self):
        # lifetime rule: a level is a view into its Pyramid and does not keep
        # it alive -- hold the Pyramid for as long as its levels are used
        p = self._pyr_ref()
        if p is None:
            raise capi.VmError(capi.VM_E_STATE, "this PyramidLevel outlived its Pyramid: keep a reference to the "
                                                "Pyramid while its levels are in use")
        return p

    def _lvl(self):
        if self._el < 1:
            raise capi.VmError(capi.VM_E_STATE, "pyramid[0] is the full-resolution placeholder")
        return self._el - 1

    def field(self, name):
        """Copy a device-state array to the host: (h,w), (h,w,2) or the mask words."""
        fid, ch = capi.FIELDS[name]
        L = self._pyr._L
        if name == "impmask":
            out = np.zeros(((self.height + 4) // 5 + 2, self.impmask_rowstride), dtype=np.uint32)
        elif ch == 2:
            out = np.zeros((self.height, self.width, 2), dtype=np.float32)
        else:
            out = np.zeros((self.height, self.width), dtype=np.float32)
        capi.check(L.vm_level_get_field(self._pyr._h, self._lvl(), fid, out.ctypes.data))
        return out

    def set_impmask(self, words):
        """test hook: overwrite the improving mask (the array field("impmask") returns), vm_dbg_level_set_mask"""
        words = np.ascontiguousarray(words, dtype=np.uint32)
        assert words.shape == ((self.height + 4) // 5 + 2, self.impmask_rowstride)
        capi.check(self._pyr._L.vm_dbg_level_set_mask(self._pyr._h, self._lvl(), words.ctypes.data))

    @property
    def v(self):
        out = np.zeros((self.height, self.width, 2), dtype=np.float32)
        capi.check(self._pyr._L.vm_level_get_v(self._pyr._h, self._lvl(), out.ctypes.data, 0))
        return out

    @v.setter
    def v(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        assert arr.shape == (self.height, self.width, 2)
        capi.check(self._pyr._L.vm_level_set_v(self._pyr._h, self._lvl(), arr.ctypes.data, 0))


class Pyramid(object):
    """class Pyramid, Pyramid.h:14-48.  pyramid[0] is the full-resolution
    placeholder (pyramid.cu:220); pyramid[1..size()-1] run finest to coarsest;
    the last level holds no images (pyramid.cu:329)."""

    def __init__(self, ctx):
        self._ctx = ctx
        self._L = capi.load()
        self._h = None
        self._levels = []
        self._vector, self._qpath = [], []
        self._extends1, self._extends2, self._results = [], [], []

    def size(self):
        return len(self._levels)

    __len__ = size

    def __getitem__(self, idx):
        return self._levels[idx]

    def back(self):
        return self._levels[-1]

    def clear(self):
        if self._h:
            self._L.vm_pyramid_destroy(self._h)
            self._h = None
        self._levels = []
        self._vector, self._qpath = [], []
        self._extends1, self._extends2 = [], []

    __del__ = clear

    def build_levels(self, sizes):
        """Allocate from explicit (w, h) level sizes, finest first."""
        self.clear()
        n = len(sizes)
        ws = (C.c_int * n)(*[int(s[0]) for s in sizes])
        hs = (C.c_int * n)(*[int(s[1]) for s in sizes])
        h = C.c_void_p()
        capi.check(self._L.vm_pyramid_create(self._ctx._h, n, ws, hs, C.byref(h)))
        self._h = h
        self._levels = [PyramidLevel(self, 0, sizes[0][0], sizes[0][1])]
        for k, (w, hh) in enumerate(sizes):
            self._levels.append(PyramidLevel(self, k + 1, w, hh))

    def upload_luma(self, el, img0, img1):
        img0 = np.ascontiguousarray(img0, dtype=np.float32)
        img1 = np.ascontiguousarray(img1, dtype=np.float32)
        lv = self._levels[el]
        assert img0.shape == (lv.height, lv.width) and img1.shape == img0.shape
        capi.check(self._L.vm_level_upload_luma(self._h, el - 1, img0.ctypes.data,
                                                img1.ctypes.data, 0))

    def build(self, img0, img1, start_res, nlevels=None):
        """Pyramid::build(video0, video1, ..., start_res), pyramid.cu:166-485, for
        one frame pair of float luma images.  Level geometry follows the
        reference (integer form of pyramid.cu:230-240, ceil halving :466-467);
        the images of the coarser levels come from a 2x2 box filter (the
        reference's Nehab-Hoppe B-spline prefiltered scale() is a 'next' row)."""
        h, w = img0.shape
        n = nlevels if nlevels is not None else synth.num_levels(w, h, start_res)
        n = max(n, 2)
        pyr = synth.build_pyramid(img0, img1, n)
        self.build_levels([(p[0].shape[1], p[0].shape[0]) for p in pyr])
        for k in range(n - 1):
            self.upload_luma(k + 1, pyr[k][0], pyr[k][1])
        self._vector = [np.zeros((h, w, 2), dtype=np.float32)]
        self._qpath = [np.zeros((h, w, 2), dtype=np.float32)]
        return pyr


def _pyramid_build_rgb(self, rgb0, rgb1, start_res, nlevels=None):
    """Pyramid::build for one pair of RGB8 frames with the reference's own image chain
    (pyramid.cu:203-211, 268-279, 355-364: load, Nehab-Hoppe cubic B-spline scale() per
    level, store_gray), run on the device (vm_pyramid.hip)."""
    rgb0 = np.ascontiguousarray(rgb0, dtype=np.uint8)
    rgb1 = np.ascontiguousarray(rgb1, dtype=np.uint8)
    h, w = rgb0.shape[:2]
    assert rgb0.shape == (h, w, 3) and rgb1.shape == rgb0.shape
    n = max(nlevels if nlevels is not None else synth.num_levels(w, h, start_res), 2)
    sizes = [(w, h)]
    for _ in range(n - 1):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    self.build_levels(sizes)
    capi.check(self._L.vm_pyramid_build_rgb(self._h, rgb0.ctypes.data, rgb1.ctypes.data, 0))
    self._vector = [np.zeros((h, w, 2), dtype=np.float32)]
    self._qpath = [np.zeros((h, w, 2), dtype=np.float32)]


Pyramid.build_rgb = _pyramid_build_rgb


class Morph(object):
    """class Morph, morph.h:10-31."""

    def __init__(self, params, pyramid, run_flag=None, fixed_work=False):
        self.m_params, self.m_pyramid = params, pyramid
        # bool& run_flag of the reference: a shared int the caller may clear
        self.m_cb = run_flag if run_flag is not None else C.c_int(1)
        self.fixed_work = bool(fixed_work)
        # ctor, morph.cu:122-141
        self._total_l = pyramid.size() - 1
        self._current_l = self._total_l
        self._total_iter = self._current_iter = 0.0
        self._max_iter = float(params.max_iter)
        iter_num = int(params.max_iter)
        for el in range(self._total_l - 1, 0, -1):
            lv = pyramid[el]
            self._total_iter += iter_num * lv.width * lv.height * lv.depth
            iter_num = int(iter_num / params.max_iter_drop_factor)
        self.progress = {}

    def params(self):
        return self.m_params

    def calculate_halfway_parametrization(self):
        """morph.cu:150-168; returns True like the reference.  Raises VmError
        (the reference throws std::runtime_error from rod::check_cuda_error)."""
        P, pyr, L = self.m_params, self.m_pyramid, self.m_pyramid._L
        ctx = pyr._ctx
        ctx.set_params(KernParameters(P))
        cons, n = _cons_array(P.constraints(0))
        w0, h0 = pyr[0].width, pyr[0].height
        flag = C.cast(C.pointer(self.m_cb), C.c_void_p)
        capi.check(L.vm_coarse_solve(pyr._h, self._total_l - 1, w0, h0, cons, n))
        self._current_l = self._total_l - 1
        while self._current_l > 0:
            if self.m_cb.value:
                el = self._current_l
                capi.check(L.vm_upsample_v(pyr._h, el - 1, el))
                capi.check(L.vm_init_level(pyr._h, el - 1, w0, h0, cons, n))
                pr = capi.Progress()
                rc = L.vm_optimize_level(pyr._h, el - 1, self._max_iter, flag,
                                         int(self.fixed_work), C.byref(pr))
                if rc != capi.VM_E_CANCELLED:
                    capi.check(rc)
                lv = pyr[el]
                # morph.cu:1389-1391: the level is accounted as max_iter sweeps
                self._current_iter += lv.width * lv.height * self._max_iter
                self.progress[el] = dict(iters=pr.iters, iters_live=pr.iters_live, improving=pr.improving,
                                         pixel_iters=pr.pixel_iters, elapsed_ms=pr.elapsed_ms,
                                         launches=pr.launches, active_tiles=pr.active_tiles,
                                         candidates=pr.candidates, commits=pr.commits,
                                         evaluations=pr.evaluations,
                                         width=lv.width, height=lv.height)
                capi.check(L.vm_level_clear(pyr._h, el - 1))
                self._max_iter /= P.max_iter_drop_factor
            self._current_l -= 1
        return True


def solve_batch(pyramids, max_iter, max_iter_drop_factor=1.0, fixed_work=False, run_flag=None, constraints=None):
    """Morph::calculate_halfway_parametrization for a batch of frame pairs (same context,
    same geometry) in lockstep: vm_solve_batch / vm_solve_batch_cons.  constraints: None, one (n, 5) array of
    (lx, ly, rx, ry, weight) shared by every pair, or a list with one such array per pair.  Returns one list of
    per-level progress dicts (finest first) per pair."""
    n = len(pyramids)
    L = pyramids[0]._L
    nl = pyramids[0].size() - 2
    arr = (C.c_void_p * n)(*[p._h for p in pyramids])
    prog = (capi.Progress * (n * nl))()
    flag = C.cast(C.pointer(run_flag), C.c_void_p) if run_flag is not None else None
    if constraints is None:
        capi.check(L.vm_solve_batch(arr, n, float(max_iter), float(max_iter_drop_factor), flag,
                                    int(bool(fixed_work)), prog))
    else:
        per = constraints if isinstance(constraints, (list, tuple)) and len(constraints) == n and np.ndim(constraints[0]) == 2 else [constraints] * n
        keep = [_cons_array(c) for c in per]
        ptrs = (C.c_void_p * n)(*[C.cast(a, C.c_void_p).value for a, _ in keep])
        cnts = (C.c_int * n)(*[k for _, k in keep])
        capi.check(L.vm_solve_batch_cons(arr, n, float(max_iter), float(max_iter_drop_factor), ptrs, cnts, flag,
                                         int(bool(fixed_work)), prog))
    out = []
    for i in range(n):
        out.append([dict(iters=prog[i * nl + k].iters, iters_live=prog[i * nl + k].iters_live, improving=prog[i * nl + k].improving,
                         pixel_iters=prog[i * nl + k].pixel_iters, elapsed_ms=prog[i * nl + k].elapsed_ms,
                         launches=prog[i * nl + k].launches, commits=prog[i * nl + k].commits,
                         candidates=prog[i * nl + k].candidates, active_tiles=prog[i * nl + k].active_tiles,
                         evaluations=prog[i * nl + k].evaluations)
                    for k in range(nl)])
    return out


class MatchingThread(object):
    """class CMatchingThread, MatchingThread.h:7-37, on threading.Thread."""

    def __init__(self, parameters, pyramids, fixed_work=False):
        self._parameters, self._pyramids = parameters, pyramids
        self._flag = C.c_int(1)
        self.percentage = 0.0
        self.run_time = 0.0
        self.gpu_morph = Morph(parameters, pyramids, self._flag, fixed_work)
        self._thread = None
        self.error = None

    @property
    def runflag(self):
        return bool(self._flag.value)

    @runflag.setter
    def runflag(self, v):
        self._flag.value = 1 if v else 0

    def run(self):
        """MatchingThread.cpp:138-150"""
        t0 = time.time()
        try:   # the whole body: an error of the delivery must reach wait() like one of the solve
            self.gpu_morph.calculate_halfway_parametrization()
            self.run_time = time.time() - t0
            self.update_result()
        except Exception as e:  # surfaced to the caller of wait()
            self.error = e
            self.run_time = time.time() - t0

    def start(self):
        self._thread = threading.Thread(target=self.run)
        self._thread.start()

    def wait(self):
        if self._thread is not None:
            self._thread.join()
        if self.error is not None:
            raise self.error

    def update_result(self):
        """MatchingThread.cpp:22-84: fetch v of the current level, scale to full
        resolution and store it in pyramid._vector[0]."""
        pyr = self._pyramids
        el = max(self.gpu_morph._current_l, 1)
        w0, h0 = pyr[0].width, pyr[0].height
        out = np.zeros((h0, w0, 2), dtype=np.float32)
        capi.check(pyr._L.vm_upscale_result(pyr._h, el - 1, w0, h0, out.ctypes.data, 0))
        pyr._vector = [out]
        m = self.gpu_morph
        self.percentage = (m._current_iter / m._total_iter * 100.0) if m._total_iter else 100.0
        return out


class Frame(object):
    """Device-resident inputs of one output frame (vm_frame): the compositor
    side of RenderWidget::RenderStage2 and CPoissonExt."""

    def __init__(self, ctx, w, h, ex):
        self._ctx, self._L = ctx, capi.load()
        self.w, self.h, self.ex = int(w), int(h), int(ex)
        hh = C.c_void_p()
        capi.check(self._L.vm_frame_create(ctx._h, self.w, self.h, self.ex, C.byref(hh)))
        self._h = hh

    def close(self):
        if getattr(self, "_h", None):
            self._L.vm_frame_destroy(self._h)
            self._h = None

    __del__ = close

    def upload_rgb(self, rgb0, rgb1):
        """the two frames as (h, w, 3) uint8: the extended canvases are built on the device (Pyramid::build,
        pyramid.cu:186-200) -- vm_frame_upload_rgb; v and the quadratic path stay what they are"""
        a0, a1 = np.ascontiguousarray(rgb0, dtype=np.uint8), np.ascontiguousarray(rgb1, dtype=np.uint8)
        assert a0.shape == (self.h, self.w, 3) and a1.shape == a0.shape
        capi.check(self._L.vm_frame_upload_rgb(self._h, a0.ctypes.data, a1.ctypes.data, 0))

    def upload(self, ext0=None, ext1=None, v=None, qpath=None):
        def ptr(a, dt):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=dt)
            return a, a.ctypes.data
        cw, ch = self.w + 2 * self.ex, self.h + 2 * self.ex
        a0, p0 = ptr(ext0, np.uint8)
        a1, p1 = ptr(ext1, np.uint8)
        av, pv = ptr(v, np.float32)
        aq, pq = ptr(qpath, np.float32)
        for a in (a0, a1):
            assert a is None or a.shape == (ch, cw, 4)
        for a in (av, aq):
            assert a is None or a.shape == (self.h, self.w, 2)
        capi.check(self._L.vm_frame_upload(self._h, p0, p1, pv, pq))

    def set_v_from_level(self, pyramid, el):
        capi.check(self._L.vm_frame_set_v_from_level(self._h, pyramid._h, el - 1))

    def set_v_from_video(self, video, lvl, frame):
        """frame `frame` of CMatchingThread::update_result over a video pair, left in this frame's v on
        the device (the frame's size is the result's w0 x h0)"""
        capi.check(self._L.vm_frame_set_v_from_video(self._h, video._h, int(lvl), int(frame)))

    def download_ext(self, side):
        out = np.zeros((self.h + 2 * self.ex, self.w + 2 * self.ex, 4), dtype=np.uint8)
        capi.check(self._L.vm_frame_download_ext(self._h, side, out.ctypes.data))
        return out

    def render_halfway(self, color_fa, geo_fa, color_from):
        """render_halfway_image, render.cu:62-96 -> (h, w, 3) uint8."""
        out = np.zeros((self.h, self.w, 3), dtype=np.uint8)
        capi.check(self._L.vm_render_halfway(self._h, color_fa, geo_fa, color_from,
                                             out.ctypes.data, 0))
        return out

    def render_halfway_dev(self, color_fa, geo_fa, color_from):
        ms = C.c_float(0)
        capi.check(self._L.vm_render_halfway_dev(self._h, color_fa, geo_fa, color_from, C.byref(ms)))
        return ms.value

    def quadratic_path(self, tol=1e-4, max_it=200):
        """CQuadraticPath::optimize for this frame's v (QuadraticPath.cpp:24-223); the result
        stays in the frame for render_halfway.  Returns (iterations, residual, ms)."""
        it, rr, ms = C.c_int(0), C.c_float(0), C.c_float(0)
        capi.check(self._L.vm_frame_quadratic_path(self._h, float(tol), int(max_it), C.byref(it), C.byref(rr), C.byref(ms)))
        return it.value, rr.value, ms.value

    def download_v(self):
        out = np.empty((self.h, self.w, 2), dtype=np.float32)
        capi.check(self._L.vm_frame_download_v(self._h, out.ctypes.data))
        return out

    def download_qpath(self):
        out = np.empty((self.h, self.w, 2), dtype=np.float32)
        capi.check(self._L.vm_frame_download_qpath(self._h, out.ctypes.data))
        return out

    def poisson_extend(self, side, tol=1e-5, max_it=20000):
        """CPoissonExt::prepare + poissonExtend for one side (PoissonExt.cpp:19-41)."""
        it, rr, ms = C.c_int(0), C.c_float(0), C.c_float(0)
        capi.check(self._L.vm_poisson_extend(self._h, side, tol, max_it, C.byref(it),
                                             C.byref(rr), C.byref(ms)))
        return it.value, rr.value, ms.value

    def poisson_extend_both(self, tol=1e-5, max_it=20000):
        """both sides of this frame as one batch: ((iters1, rel1), (iters2, rel2), ms)"""
        per_frame, ms = poisson_extend_frames([self], tol, max_it)
        return per_frame[0][0], per_frame[0][1], ms


def poisson_extend_frames(frames, tol=1e-5, max_it=20000):
    """CPoissonExt::run's loop body (PoissonExt.cpp:24-36) for several frames of one context at once: both sides of
    every frame are independent systems and share every launch.  Returns ([((iters, rel) side 1, (iters, rel) side 2)
    per frame], elapsed ms)."""
    n = len(frames)
    arr = (C.c_void_p * n)(*[f._h for f in frames])
    it, rr, ms = (C.c_int * (2 * n))(), (C.c_float * (2 * n))(), C.c_float(0)
    capi.check(frames[0]._L.vm_poisson_extend_frames(arr, n, tol, max_it, it, rr, C.byref(ms)))
    return [((it[2 * i], rr[2 * i]), (it[2 * i + 1], rr[2 * i + 1])) for i in range(n)], ms.value


def context_beside(device, math_mode, others, tries=8):
    """A new Context whose stream runs side by side with the streams of all `others` (contexts of the same device): the
    runtime deals new streams to its GPU_MAX_HW_QUEUES hardware queues as it likes, and two streams on one queue run their
    kernels one after the other.  Candidates that share a queue with one of `others` are kept alive while the search goes on
    (so that the next stream gets another queue) and closed at the end; after `tries` candidates the last one is returned
    whatever it shares.  Returns (context, number of rejected candidates)."""
    rejected = []
    probe = not os.environ.get("VM_NO_STREAM_PROBE")          # development switch: take the first stream, as rounds 1-5 did
    while True:
        c = Context(device, math_mode)
        if not probe or len(rejected) >= tries - 1 or all(c.runs_beside(o) for o in others):
            for r in rejected:
                r.close()
            return c, len(rejected)
        rejected.append(c)


def pin_host(array):
    """page-lock a numpy array frames are uploaded from (vm_host_register); returns the array"""
    a = np.ascontiguousarray(array)
    capi.check(capi.load().vm_host_register(a.ctypes.data, a.nbytes))
    return a


def unpin_host(array):
    capi.check(capi.load().vm_host_unregister(array.ctypes.data))


def make_extended(rgb, ex):
    """Extended RGBA8 canvas of Pyramid::build, pyramid.cu:186-200: filled with
    (255,255,255,255), the image pasted at (ex,ex) with alpha 0."""
    h, w = rgb.shape[:2]
    can = np.full((h + 2 * ex, w + 2 * ex, 4), 255, dtype=np.uint8)
    can[ex:ex + h, ex:ex + w, :3] = rgb
    can[ex:ex + h, ex:ex + w, 3] = 0
    return can


# ---------------------------------------------------------------------------------------------
# video pairs: class Pyramid with depth > 1 and the temporally coupled Morph

def _vcons_array(cons):
    """rows (lx, ly, rx, ry, weight, frame) -> VideoConstraint array"""
    cons = np.asarray(cons if cons is not None else [], dtype=np.float32).reshape(-1, 6)
    n = len(cons)
    arr = (capi.VideoConstraint * max(n, 1))()
    for k in range(n):
        arr[k] = capi.VideoConstraint(float(cons[k][0]), float(cons[k][1]), float(cons[k][2]), float(cons[k][3]),
                                      float(cons[k][4]), int(cons[k][5]))
    return arr, n


def video_constraints(P):
    """Parameters::lp/rp/cnt resolved with the frame of the left point (morph.cu:354-366)"""
    out = []
    for row in P.cnt:
        for c in row:
            l = P.lp[c.li[0]][c.li[1]]
            r = P.rp[c.ri[0]][c.ri[1]]
            out.append((l.p[0], l.p[1], r.p[0], r.p[1], min(l.weight, r.weight), l.p[2]))
    return np.asarray(out, dtype=np.float32).reshape(-1, 6)


class VideoPage(object):
    """One page of a PyramidLevel of depth > 1 (device-state access)."""

    def __init__(self, vid, lvl, page, w, h):
        self._vid_ref, self._lvl, self._page = weakref.ref(vid), lvl, page   # no cycle (see PyramidLevel)
        self.width, self.height = int(w), int(h)

    @property
    def _vid(self):
        v = self._vid_ref()
        if v is None:
            raise capi.VmError(capi.VM_E_STATE, "this VideoPage outlived its VideoPyramid: keep a reference to the "
                                                "VideoPyramid while its pages are in use")
        return v

    def field(self, name):
        fid, ch = capi.FIELDS[name]
        if name == "impmask":
            out = np.zeros(((self.height + 4) // 5 + 2, (self.width + 4) // 5 + 2), dtype=np.uint32)
        elif ch == 2:
            out = np.zeros((self.height, self.width, 2), dtype=np.float32)
        else:
            out = np.zeros((self.height, self.width), dtype=np.float32)
        capi.check(self._vid._L.vm_video_get_field(self._vid._h, self._lvl, self._page, fid, out.ctypes.data))
        return out

    @property
    def v(self):
        out = np.zeros((self.height, self.width, 2), dtype=np.float32)
        capi.check(self._vid._L.vm_video_get_v(self._vid._h, self._lvl, self._page, out.ctypes.data, 0))
        return out

    @v.setter
    def v(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        assert arr.shape == (self.height, self.width, 2)
        capi.check(self._vid._L.vm_video_set_v(self._vid._h, self._lvl, self._page, arr.ctypes.data, 0))


class VideoPyramid(object):
    """class Pyramid (Pyramid.h:14-48) for a video pair: level l holds depth[l] pages and four
    flow fields per page.  pages[l][t]: l = 0 finest ... last = coarsest (v only)."""

    def __init__(self, ctx):
        self._ctx, self._L = ctx, capi.load()
        self._h = None
        self.levels, self.factor_t, self.pages = [], [], []

    def clear(self):
        if getattr(self, "_h", None):
            self._L.vm_video_destroy(self._h)
            self._h = None

    __del__ = clear

    def build_levels(self, levels, factor_t=None, depth0=None):
        """levels: [(w, h, d), ...] finest first incl. the coarsest"""
        self.clear()
        n = len(levels)
        ws = (C.c_int * n)(*[int(l[0]) for l in levels])
        hs = (C.c_int * n)(*[int(l[1]) for l in levels])
        ds = (C.c_int * n)(*[int(l[2]) for l in levels])
        ft = (C.c_int * n)(*[int(x) for x in factor_t]) if factor_t is not None else None
        h = C.c_void_p()
        capi.check(self._L.vm_video_create(self._ctx._h, n, ws, hs, ds, ft, int(depth0 if depth0 is not None else levels[0][2]), C.byref(h)))
        self._h = h
        self.levels = [tuple(int(x) for x in l) for l in levels]
        self.depth0 = int(depth0 if depth0 is not None else levels[0][2])
        self._vector = []
        self.factor_t = list(factor_t) if factor_t is not None else [1] + [2 if levels[i][2] != levels[i - 1][2] else 1 for i in range(1, n)]
        self.pages = [[VideoPage(self, l, t, levels[l][0], levels[l][1]) for t in range(levels[l][2])] for l in range(n)]

    def result(self, lvl, w0=None, h0=None):
        """CMatchingThread::update_result for depth > 1 (MatchingThread.cpp:22-84): the depth0
        full-resolution frames of the halfway field from level `lvl` -- every page scaled and
        resized, the frames the temporal pyramid skipped blended from their neighbours"""
        w0 = int(w0 if w0 is not None else self.levels[0][0])
        h0 = int(h0 if h0 is not None else self.levels[0][1])
        out = np.zeros((self.depth0, h0, w0, 2), dtype=np.float32)
        capi.check(self._L.vm_video_result(self._h, int(lvl), w0, h0, out.ctypes.data))
        return out

    def factor_d(self, lvl):
        f = C.c_float(0)
        capi.check(self._L.vm_video_level_dims(self._h, lvl, None, None, None, C.byref(f)))
        return f.value

    def upload_luma(self, lvl, page, img0, img1):
        img0 = np.ascontiguousarray(img0, dtype=np.float32)
        img1 = np.ascontiguousarray(img1, dtype=np.float32)
        capi.check(self._L.vm_video_upload_luma(self._h, lvl, page, img0.ctypes.data, img1.ctypes.data, 0))

    def upload_flows(self, lvl, page, f0, f1, b0, b1):
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in (f0, f1, b0, b1)]
        capi.check(self._L.vm_video_upload_flows(self._h, lvl, page, *[x.ctypes.data for x in a], 0))

    def build_flows(self, f0, f1, b0, b1):
        """flow half of Pyramid::build on the device from the full-resolution flows of every frame"""
        fam = [[np.ascontiguousarray(x, dtype=np.float32) for x in f] for f in (f0, f1, b0, b1)]
        n = len(fam[0])
        ptrs = [(C.c_void_p * n)(*[a.ctypes.data for a in f]) for f in fam]
        capi.check(self._L.vm_video_build_flows(self._h, *ptrs))

    def build_rgb_frame(self, frame, rgb0, rgb1):
        rgb0 = np.ascontiguousarray(rgb0, dtype=np.uint8)
        rgb1 = np.ascontiguousarray(rgb1, dtype=np.uint8)
        capi.check(self._L.vm_video_build_rgb(self._h, int(frame), rgb0.ctypes.data, rgb1.ctypes.data, 0))

    def build(self, video0, video1, f0, f1, b0, b1, start_res):
        """Pyramid::build(video0, video1, f0, f1, b0, b1, start_res), pyramid.cu:166-485, for float
        luma frames: geometry incl. the temporal pyramid (synth.video_levels), box-filtered lumas
        per page (as Pyramid.build of a pair does), flows through the device builder."""
        h, w = video0[0].shape
        d = len(video0)
        levels, factor_t = synth.video_levels(w, h, d, start_res)
        self.build_levels(levels, factor_t, d)
        frames = synth.page_frames(levels, factor_t)
        pyr = [synth.build_pyramid(video0[t], video1[t], len(levels)) for t in range(d)]
        for l in range(len(levels) - 1):
            for t in range(levels[l][2]):
                self.upload_luma(l, t, *pyr[frames[l][t]][l])
        self.build_flows(f0, f1, b0, b1)


class VideoMorph(object):
    """class Morph (morph.h:10-31) over a VideoPyramid: the temporally coupled solve."""

    def __init__(self, params, pyramid, run_flag=None, fixed_work=False):
        self.m_params, self.m_pyramid = params, pyramid
        self.m_cb = run_flag if run_flag is not None else C.c_int(1)
        self.fixed_work = bool(fixed_work)
        self._max_iter = float(params.max_iter)
        self.progress = {}

    def calculate_halfway_parametrization(self):
        """morph.cu:150-168 with the page schedule of optimize_level (:1353-1441)"""
        P, vid = self.m_params, self.m_pyramid
        vid._ctx.set_params(KernParameters(P))
        cons, n = _vcons_array(video_constraints(P))
        flag = C.cast(C.pointer(self.m_cb), C.c_void_p)
        total = sum(l[2] for l in vid.levels[:-1])
        prog = (capi.Progress * total)()
        capi.check(vid._L.vm_video_solve(vid._h, self._max_iter, float(P.max_iter_drop_factor), cons, n, flag,
                                         int(self.fixed_work), prog))
        k = 0
        for l in range(len(vid.levels) - 1):
            for t in range(vid.levels[l][2]):
                pr = prog[k]
                self.progress[(l, t)] = dict(iters=pr.iters, iters_live=pr.iters_live, improving=pr.improving, commits=pr.commits,
                                             candidates=pr.candidates, elapsed_ms=pr.elapsed_ms)
                k += 1
        return True


class VideoMatchingThread(object):
    """class CMatchingThread (MatchingThread.h:7-37) over a video pair: the temporally coupled solve on
    a worker thread, then update_result() -- pyramid._vector[frame] for every frame of the video,
    at w0 x h0 (the size of the reference's placeholder level pyramid[0]; default: the finest level's)."""

    def __init__(self, parameters, pyramids, w0=None, h0=None, fixed_work=False):
        self._parameters, self._pyramids = parameters, pyramids
        self._flag = C.c_int(1)
        self.w0 = int(w0 if w0 is not None else pyramids.levels[0][0])
        self.h0 = int(h0 if h0 is not None else pyramids.levels[0][1])
        self.percentage = 0.0
        self.run_time = 0.0
        self.gpu_morph = VideoMorph(parameters, pyramids, self._flag, fixed_work)
        self._thread = None
        self.error = None

    @property
    def runflag(self):
        return bool(self._flag.value)

    @runflag.setter
    def runflag(self, v):
        self._flag.value = 1 if v else 0

    def run(self):
        """MatchingThread.cpp:138-150"""
        t0 = time.time()
        try:   # the whole body: an error of the delivery (vm_video_result) must reach wait() like one of the solve
            self.gpu_morph.calculate_halfway_parametrization()
            self.run_time = time.time() - t0
            self.update_result()
        except Exception as e:  # surfaced to the caller of wait()
            self.error = e
            self.run_time = time.time() - t0

    def start(self):
        self._thread = threading.Thread(target=self.run)
        self._thread.start()

    def wait(self):
        if self._thread is not None:
            self._thread.join()
        if self.error is not None:
            raise self.error

    def update_result(self, lvl=0):
        """MatchingThread.cpp:22-84 (the level the solver has reached; after run(): the finest)"""
        self._pyramids._vector = list(self._pyramids.result(lvl, self.w0, self.h0))
        self.percentage = 100.0


# ---- synchronisation stage: CSyncThread (SyncThread.h:7-39) + the stage-1 renderer ------------

def sync_constraints(P):
    """Resolve lp / rp / cnt the way CSyncThread::genMatrix walks them (SyncThread.cpp:155-165):
    every Connect ties (x, y, frame) of video 0 to (x, y, frame) of video 1."""
    out = []
    for row in P.cnt:
        for c in row:
            l, r = P.lp[c.li[0]][c.li[1]], P.rp[c.ri[0]][c.ri[1]]
            out.append((int(l.p[0]), int(l.p[1]), int(l.p[2]), int(r.p[0]), int(r.p[1]), int(r.p[2])))
    return out


def sync_level_table(w, h, d, start_res):
    """[(w, h, d)] of Pyramid::build(video0, video1, f0, f1, start_res): entry 0 = full resolution"""
    L = capi.load()
    lw, lh, ld = ((C.c_int * 64)() for _ in range(3))
    n = C.c_int(0)
    capi.check(L.vm_sync_level_table(int(w), int(h), int(d), int(start_res), lw, lh, ld, 64, C.byref(n)))
    return [(lw[i], lh[i], ld[i]) for i in range(n.value)]


class SyncPyramid(object):
    """What class Pyramid holds after build(video0, video1, f0, f1, start_res) (pyramid.cu:57-165):
    the level table, the four layered arrays the stage-1 renderer samples, and _vector."""

    def __init__(self, ctx):
        self._ctx, self._L = ctx, capi.load()
        self._h = None
        self.levels = []
        self._vector = []

    def clear(self):
        if getattr(self, "_h", None):
            self._L.vm_sync_destroy(self._h)
            self._h = None

    __del__ = clear

    def size(self):
        return len(self.levels)

    def build_levels(self, levels):
        self.clear()
        n = len(levels)
        ws = (C.c_int * n)(*[int(l[0]) for l in levels])
        hs = (C.c_int * n)(*[int(l[1]) for l in levels])
        ds = (C.c_int * n)(*[int(l[2]) for l in levels])
        h = C.c_void_p()
        capi.check(self._L.vm_sync_create(self._ctx._h, n, ws, hs, ds, C.byref(h)))
        self._h = h
        self.levels = [tuple(int(x) for x in l) for l in levels]
        w0, h0, d0 = self.levels[0]
        self._vector = [np.zeros((h0, w0, 4), np.float32) for _ in range(d0)]

    def build(self, video0, video1, f0, f1, start_res):
        """video*: (d, h, w, 3 or 4) u8; f*: (d, h, w, 2) float32 forward flows"""
        d, h, w = video0.shape[:3]
        self.build_levels(sync_level_table(w, h, d, start_res))
        for side, (vid, fl) in enumerate(((video0, f0), (video1, f1))):
            for t in range(d):
                self.upload_frame(side, t, vid[t])
                self.upload_flow(side, t, fl[t])

    def upload_frame(self, side, frame, rgb):
        rgb = np.asarray(rgb, np.uint8)
        if rgb.shape[-1] == 3:
            rgb = np.concatenate([rgb, np.zeros(rgb.shape[:2] + (1,), np.uint8)], axis=-1)
        rgb = np.ascontiguousarray(rgb)
        capi.check(self._L.vm_sync_upload_frame(self._h, side, frame, rgb.ctypes.data, rgb.shape[1] * 4))

    def upload_flow(self, side, frame, flow):
        flow = np.ascontiguousarray(flow, np.float32)
        capi.check(self._L.vm_sync_upload_flow(self._h, side, frame, flow.ctypes.data, flow.shape[1] * 2))

    def field(self, lvl):
        """(X, Y, Z) of level lvl, each (d, h, w) float32"""
        w, h, d = self.levels[lvl]
        out = [np.zeros((d, h, w), np.float32) for _ in range(3)]
        capi.check(self._L.vm_sync_get_field(self._h, lvl, *[o.ctypes.data for o in out]))
        return out

    def set_field(self, lvl, X, Y, Z):
        a = [np.ascontiguousarray(t, np.float32) for t in (X, Y, Z)]
        capi.check(self._L.vm_sync_set_field(self._h, lvl, *[t.ctypes.data for t in a]))

    def result(self, lvl, frame):
        w0, h0, _ = self.levels[0]
        out = np.zeros((h0, w0, 4), np.float32)
        capi.check(self._L.vm_sync_result(self._h, lvl, frame, out.ctypes.data))
        return out

    def render_resample(self, fa, frame):
        """RenderWidget::RenderStage1 (UI/RenderWidget.cpp:205-227) -> (h, w, 3) u8"""
        w0, h0, _ = self.levels[0]
        out = np.zeros((h0, w0, 3), np.uint8)
        capi.check(self._L.vm_sync_render(self._h, float(fa), int(frame), out.ctypes.data, w0 * 3))
        return out

    def render_resample_dev(self, fa, frame):
        ms = C.c_float(0)
        capi.check(self._L.vm_sync_render_dev(self._h, float(fa), int(frame), C.byref(ms)))
        return ms.value


class SyncThread(object):
    """class CSyncThread (SyncThread.h:7-39) on threading.Thread: runflag, percentage, run_time,
    run(), update_result()."""

    def __init__(self, parameters, pyramids):
        self._parameters, self._pyramids = parameters, pyramids
        self._flag = C.c_int(1)
        self.percentage = 0.0
        self.run_time = 0.0
        self._total_l = pyramids.size() - 1
        self._current_l = self._total_l
        self._max_iter = float(parameters.max_iter * 10)
        self._current_iter = 0.0
        self._total_iter = 0.0
        it = parameters.max_iter * 10
        for el in range(self._total_l, 0, -1):  # SyncThread.cpp:15-22 (integer division by the drop factor)
            w, h, d = pyramids.levels[el]
            self._total_iter += float(it) * w * h * d
            it = int(it / parameters.max_iter_drop_factor)
        self.progress = {}
        self._thread = None
        self.error = None

    @property
    def runflag(self):
        return bool(self._flag.value)

    @runflag.setter
    def runflag(self, v):
        self._flag.value = 1 if v else 0

    def load_identity(self, el):
        capi.check(self._pyramids._L.vm_sync_load_identity(self._pyramids._h, el))

    def upsample_level(self, el, pel=None):
        capi.check(self._pyramids._L.vm_sync_upsample_level(self._pyramids._h, el))

    def optimize_level(self, el):
        pyr = self._pyramids
        pyr._ctx.set_params(KernParameters(self._parameters))
        cons = sync_constraints(self._parameters)
        arr = (capi.SyncConstraint * max(len(cons), 1))()
        for i, c in enumerate(cons):
            arr[i] = capi.SyncConstraint(*c)
        capi.check(pyr._L.vm_sync_set_constraints(pyr._h, arr, len(cons)))
        pr = capi.SyncProgress()
        flag = C.cast(C.pointer(self._flag), C.c_void_p)
        capi.check(pyr._L.vm_sync_optimize_level(pyr._h, el, self._max_iter, flag, C.byref(pr)))
        self.progress[el] = dict(iters=pr.iters, launches=pr.launches, voxel_iters=pr.voxel_iters,
                                 elapsed_ms=pr.elapsed_ms, resid=tuple(pr.resid))
        self._current_iter += pr.voxel_iters
        return pr

    def run(self):
        """SyncThread.cpp:58-84"""
        t0 = time.time()
        try:
            self._current_l = self._total_l
            while self._current_l > 0:
                el = self._current_l
                if el == self._total_l:
                    self.load_identity(el)
                else:
                    self.upsample_level(el, el + 1)
                self.optimize_level(el)
                self._max_iter = float(np.float32(self._max_iter) / np.float32(2))
                if not self.runflag:
                    break
                self._current_l -= 1
        except Exception as e:
            self.error = e
        self.run_time = time.time() - t0
        if self.error is None:
            self.update_result()

    def start(self):
        self._thread = threading.Thread(target=self.run)
        self._thread.start()

    def wait(self):
        if self._thread is not None:
            self._thread.join()
        if self.error is not None:
            raise self.error

    def update_result(self):
        """SyncThread.cpp:482-521: _vector[z] = (X ratio_x, Y ratio_y, Z, 0) resized to full size"""
        pyr = self._pyramids
        el = max(self._current_l, 1)
        for z in range(pyr.levels[el][2]):
            pyr._vector[z] = pyr.result(el, z)
        self.percentage = (self._current_iter / self._total_iter * 100.0) if self._total_iter else 100.0
