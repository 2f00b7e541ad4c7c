"""ctypes binding of the C-ABI (include/vmorph.h) -- thin, no logic.

The shared library is built in-tree by videomorphing_amd.build.  Loading
fails loudly when it is missing; creating a context fails loudly when there is
no HIP device (there is no CPU fallback in the product path).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VM_LIB_PATH: a variant build of the same library, development only -- tools/gpu_variant.sh)
LIB_PATH = os.environ.get("VM_LIB_PATH") or os.path.join(_HERE, "lib", "libvmorph_hip.so")

VM_OK = 0
VM_E_INVALID, VM_E_DEVICE, VM_E_STATE, VM_E_NUMERIC, VM_E_CANCELLED = -1, -2, -3, -4, -5
BCOND_NONE, BCOND_CORNER, BCOND_BORDER = 0, 1, 2
MATH_EXACT, MATH_FAST, MATH_EXACT_FMA, MATH_REF_FASTMATH, MATH_REF_TEX8, MATH_REF_TEX8_TRUNC = 0, 1, 2, 3, 4, 5
SWEEP_AUTO, SWEEP_TILE, SWEEP_SPLIT, SWEEP_STEP, SWEEP_SPARSE, SWEEP_PASS = 0, 1, 2, 3, 4, 5

FIELDS = {  # name -> (id, channels)
    "img0": (0, 1), "img1": (1, 1), "v": (2, 2), "luma": (3, 2), "mean": (4, 2), "var": (5, 2),
    "cross": (6, 1), "value": (7, 1), "counter": (8, 1), "tps_axy": (9, 1), "tps_b": (10, 2),
    "ui_axy": (11, 1), "ui_b": (12, 2), "impmask": (13, 1),
    "temp_ref": (14, 2), "temp_mask": (15, 1), "f0": (16, 2), "f1": (17, 2), "b0": (18, 2), "b1": (19, 2),
}

# every symbol include/vmorph.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vm_last_error", "vm_version", "vm_ctx_create", "vm_ctx_destroy", "vm_ctx_sync",
    "vm_set_params", "vm_get_params", "vm_set_math_mode", "vm_set_tuning", "vm_set_commit_order", "vm_dbg_pass_placement", "vm_dbg_pass_force_timeout",
    "vm_dbg_pass_fallbacks", "vm_dbg_level_set_mask", "vm_dbg_sparse_resident", "vm_dbg_sparse_resident_visits", "vm_dbg_streams_overlap", "vm_device_info",
    "vm_pyramid_create", "vm_pyramid_destroy", "vm_pyramid_levels", "vm_level_dims",
    "vm_level_upload_luma", "vm_pyramid_build_rgb", "vm_level_set_v", "vm_level_get_v", "vm_level_get_field",
    "vm_level_clear", "vm_coarse_solve", "vm_upsample_v", "vm_init_level", "vm_optimize_level",
    "vm_solve", "vm_optimize_level_batch", "vm_solve_batch", "vm_solve_batch_cons", "vm_upscale_result", "vm_frame_create", "vm_frame_destroy", "vm_frame_upload", "vm_frame_upload_rgb",
    "vm_frame_download_ext", "vm_host_register", "vm_host_unregister", "vm_frame_set_v_from_level", "vm_render_halfway",
    "vm_render_halfway_dev", "vm_poisson_extend", "vm_poisson_extend_frames", "vm_frame_quadratic_path", "vm_frame_download_qpath", "vm_frame_download_v",
    "vm_rccl_bcast", "vm_rccl_comm_init_all", "vm_rccl_comm_destroy", "vm_bcast_params", "vm_bcast_bytes", "vm_dbg_poisson_profile",
    "vm_video_create", "vm_video_destroy", "vm_video_levels", "vm_video_level_dims", "vm_video_upload_luma",
    "vm_video_upload_flows", "vm_video_build_rgb", "vm_video_build_flows", "vm_video_set_v", "vm_video_get_v",
    "vm_video_get_field", "vm_video_coarse_solve", "vm_video_upsample", "vm_video_init_level",
    "vm_video_initialize_temp", "vm_video_optimize_level", "vm_video_solve", "vm_video_result", "vm_frame_set_v_from_video",
    "vm_sync_level_table", "vm_sync_create", "vm_sync_destroy", "vm_sync_set_constraints", "vm_sync_load_identity",
    "vm_sync_upsample_level", "vm_sync_optimize_level", "vm_sync_solve", "vm_sync_get_field", "vm_sync_set_field",
    "vm_sync_result", "vm_sync_upload_frame", "vm_sync_upload_flow", "vm_sync_render", "vm_sync_render_dev",
]


class KernParams(C.Structure):
    """struct KernParameters, Algorithm/parameters.h:54-72."""
    _fields_ = [("w_temp", C.c_float), ("w_ui", C.c_float), ("w_tps", C.c_float),
                ("w_ssim", C.c_float), ("ssim_clamp", C.c_float), ("eps", C.c_float),
                ("bcond", C.c_int)]


class Constraint(C.Structure):
    _fields_ = [("lx", C.c_float), ("ly", C.c_float), ("rx", C.c_float), ("ry", C.c_float),
                ("weight", C.c_float)]


class VideoConstraint(C.Structure):
    _fields_ = [("lx", C.c_float), ("ly", C.c_float), ("rx", C.c_float), ("ry", C.c_float),
                ("weight", C.c_float), ("frame", C.c_int)]


class Progress(C.Structure):
    _fields_ = [("iters", C.c_int), ("improving", C.c_int), ("pixel_iters", C.c_double),
                ("elapsed_ms", C.c_float), ("launches", C.c_int),
                ("active_tiles", C.c_double), ("candidates", C.c_double), ("commits", C.c_double),
                ("evaluations", C.c_double), ("sched_ms", C.c_float * 5), ("sched_launches", C.c_int * 5),
                ("iters_live", C.c_int), ("clk_shader_ticks", C.c_double * 2), ("clk_wall_ticks", C.c_double * 2)]


class SyncConstraint(C.Structure):
    """a Connect between lp[li] and rp[ri]: full-resolution pixel + frame on either side"""
    _fields_ = [("lx", C.c_int), ("ly", C.c_int), ("lz", C.c_int), ("rx", C.c_int), ("ry", C.c_int), ("rz", C.c_int)]


class SyncProgress(C.Structure):
    _fields_ = [("iters", C.c_int), ("launches", C.c_int), ("voxel_iters", C.c_double),
                ("elapsed_ms", C.c_float), ("resid", C.c_float * 3)]


class ParamBlock(C.Structure):
    _fields_ = [("kp", KernParams), ("max_iter", C.c_float), ("max_iter_drop_factor", C.c_float),
                ("start_res", C.c_int), ("math_mode", C.c_int), ("n_constraints", C.c_int)]


class VmError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "vmorph error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -m videomorphing_amd.build` "
                          "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.vm_last_error.restype = C.c_char_p
    L.vm_version.restype = C.c_char_p
    sig = {
        "vm_ctx_create": [i, C.POINTER(vp)],
        "vm_ctx_sync": [vp],
        "vm_set_params": [vp, C.POINTER(KernParams)],
        "vm_get_params": [vp, C.POINTER(KernParams)],
        "vm_set_math_mode": [vp, i],
        "vm_set_tuning": [vp, i, i, i],
        "vm_set_commit_order": [vp, i],
        "vm_dbg_pass_placement": [vp, vp, i],
        "vm_dbg_pass_force_timeout": [vp, i],
        "vm_dbg_pass_fallbacks": [vp],
        "vm_dbg_sparse_resident": [vp, C.c_int],
        "vm_dbg_sparse_resident_visits": [vp],
        "vm_device_info": [vp, C.c_char_p, C.POINTER(i), C.POINTER(C.c_uint64)],
        "vm_pyramid_create": [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(vp)],
        "vm_pyramid_levels": [vp],
        "vm_level_dims": [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)],
        "vm_level_upload_luma": [vp, i, vp, vp, i],
        "vm_pyramid_build_rgb": [vp, vp, vp, i],
        "vm_level_set_v": [vp, i, vp, i],
        "vm_level_get_v": [vp, i, vp, i],
        "vm_level_get_field": [vp, i, i, vp],
        "vm_level_clear": [vp, i],
        "vm_dbg_level_set_mask": [vp, i, vp],
        "vm_coarse_solve": [vp, i, i, i, vp, i],
        "vm_upsample_v": [vp, i, i],
        "vm_init_level": [vp, i, i, i, vp, i],
        "vm_optimize_level": [vp, i, f, vp, i, C.POINTER(Progress)],
        "vm_solve": [vp, f, f, vp, i, vp, i, vp],
        "vm_optimize_level_batch": [vp, i, i, f, vp, i, vp],
        "vm_solve_batch": [vp, i, f, f, vp, i, vp],
        "vm_solve_batch_cons": [vp, i, f, f, vp, vp, vp, i, vp],
        "vm_upscale_result": [vp, i, i, i, vp, i],
        "vm_frame_create": [vp, i, i, i, C.POINTER(vp)],
        "vm_frame_upload": [vp, vp, vp, vp, vp],
        "vm_frame_upload_rgb": [vp, vp, vp, i],
        "vm_dbg_streams_overlap": [vp, vp, C.POINTER(i)],
        "vm_frame_download_ext": [vp, i, vp],
        "vm_frame_set_v_from_level": [vp, vp, i],
        "vm_render_halfway": [vp, f, f, i, vp, i],
        "vm_render_halfway_dev": [vp, f, f, i, C.POINTER(f)],
        "vm_poisson_extend": [vp, i, f, i, C.POINTER(i), C.POINTER(f), C.POINTER(f)],
        "vm_poisson_extend_frames": [C.POINTER(vp), i, f, i, C.POINTER(i), C.POINTER(f), C.POINTER(f)],
        "vm_frame_quadratic_path": [vp, f, i, C.POINTER(i), C.POINTER(f), C.POINTER(f)],
        "vm_frame_download_qpath": [vp, vp],
        "vm_frame_download_v": [vp, vp],
        "vm_rccl_bcast": [vp, vp, vp, C.c_uint64, i],
        "vm_host_register": [vp, C.c_uint64],
        "vm_host_unregister": [vp],
        "vm_rccl_comm_init_all": [i, C.POINTER(i), C.POINTER(vp)],
        "vm_bcast_params": [C.POINTER(vp), C.POINTER(vp), i, i, C.POINTER(ParamBlock), C.POINTER(ParamBlock)],
        "vm_bcast_bytes": [C.POINTER(vp), C.POINTER(vp), i, i, vp, C.c_uint64, C.POINTER(vp)],
        "vm_dbg_poisson_profile": [vp, i, C.POINTER(C.c_double), C.POINTER(i), C.POINTER(C.c_double), C.POINTER(i)],
        "vm_video_create": [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i), i, C.POINTER(vp)],
        "vm_video_levels": [vp],
        "vm_video_level_dims": [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(f)],
        "vm_video_upload_luma": [vp, i, i, vp, vp, i],
        "vm_video_upload_flows": [vp, i, i, vp, vp, vp, vp, i],
        "vm_video_build_rgb": [vp, i, vp, vp, i],
        "vm_video_build_flows": [vp, vp, vp, vp, vp],
        "vm_video_set_v": [vp, i, i, vp, i],
        "vm_video_get_v": [vp, i, i, vp, i],
        "vm_video_get_field": [vp, i, i, i, vp],
        "vm_video_coarse_solve": [vp, vp, i],
        "vm_video_upsample": [vp, i],
        "vm_video_init_level": [vp, i, vp, i],
        "vm_video_initialize_temp": [vp, i, i, i],
        "vm_video_optimize_level": [vp, i, f, vp, i, vp],
        "vm_video_solve": [vp, f, f, vp, i, vp, i, vp],
        "vm_video_result": [vp, i, i, i, vp],
        "vm_frame_set_v_from_video": [vp, vp, i, i],
        "vm_sync_level_table": [i, i, i, i, vp, vp, vp, i, C.POINTER(i)],
        "vm_sync_create": [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(vp)],
        "vm_sync_set_constraints": [vp, vp, i],
        "vm_sync_load_identity": [vp, i],
        "vm_sync_upsample_level": [vp, i],
        "vm_sync_optimize_level": [vp, i, f, vp, C.POINTER(SyncProgress)],
        "vm_sync_solve": [vp, f, vp, vp],
        "vm_sync_get_field": [vp, i, vp, vp, vp],
        "vm_sync_set_field": [vp, i, vp, vp, vp],
        "vm_sync_result": [vp, i, i, vp],
        "vm_sync_upload_frame": [vp, i, i, vp, i],
        "vm_sync_upload_flow": [vp, i, i, vp, i],
        "vm_sync_render": [vp, f, i, vp, i],
        "vm_sync_render_dev": [vp, f, i, C.POINTER(f)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i
    for name in ("vm_ctx_destroy", "vm_pyramid_destroy", "vm_frame_destroy", "vm_video_destroy", "vm_sync_destroy", "vm_rccl_comm_destroy"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = None
    _lib = L
    return L


def check(rc):
    if rc != VM_OK:
        raise VmError(rc, load().vm_last_error().decode("utf-8", "replace"))
    return rc
