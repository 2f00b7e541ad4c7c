"""Builds libvmorph_hip.so (HIP kernels + C-ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The optimizer kernels are compiled twice
from one source: the EXACT arithmetic mode with -ffp-contract=off (bit-identical
to the CPU oracle) and the FAST mode with fused multiply-adds.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# A variant build (VM_DEFS = extra compiler flags, development: tools/gpu_variant.sh) gets object and library
# directories of its own, named after the flags: its objects can never be linked into the product library by a
# later incremental build, and nothing has to be restored afterwards.  Load it with VM_LIB_PATH=<printed path>.
_VARIANT = os.environ.get("VM_DEFS", "").strip()
_TAG = ("_" + hashlib.sha1(_VARIANT.encode()).hexdigest()[:10]) if _VARIANT else ""
OBJ = os.path.join(HERE, "build" + _TAG)
LIBDIR = os.path.join(HERE, "lib" + _TAG)
LIB = os.path.join(LIBDIR, "libvmorph_hip.so")
ARCH = "gfx950"

COMMON = _VARIANT.split() + ["-O3", "-fPIC", "-std=c++17", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
          "-I" + os.path.join(HERE, "..", "include")]

# (source, object, extra flags)
UNITS = [
    ("vm_morph_kernels.hip", "vm_morph_kernels_exact.o", ["-DVM_EXACT=1", "-ffp-contract=off"]),
    ("vm_morph_kernels.hip", "vm_morph_kernels_fast.o", ["-DVM_EXACT=0", "-ffp-contract=fast"]),
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_exact.o", ["-DVM_EXACT=1", "-ffp-contract=off"]),
    # VM_MATH_EXACT_FMA (diagnostic): the EXACT source with contraction on -- what nvcc's default --fmad=true
    # makes of the reference source; one more legal rounding of the algorithm for the chaos-floor tests
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_exactf.o", ["-DVM_EXACT=2", "-ffp-contract=fast"]),
    # VM_MATH_REF_FASTMATH (diagnostic): the same source as the reference's project compiles it -- --use_fast_math:
    # contraction, approximate division and square root
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_reffm.o", ["-DVM_EXACT=3", "-ffp-contract=fast"]),
    # VM_MATH_REF_TEX8 / _TRUNC (diagnostic): the EXACT source, IEEE, with CUDA's 8-bit bilinear filter weights in
    # every texture fetch (images and the inter-level upsample): what separates the reference BINARY's arithmetic
    # from every other member of the family of legal builds (vm_morph_common.h: tex8_weight)
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_tex8.o", ["-DVM_EXACT=4", "-ffp-contract=off"]),
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_tex8t.o", ["-DVM_EXACT=5", "-ffp-contract=off"]),
    ("vm_morph_kernels.hip", "vm_morph_kernels_tex8.o", ["-DVM_EXACT=4", "-ffp-contract=off"]),
    ("vm_morph_kernels.hip", "vm_morph_kernels_tex8t.o", ["-DVM_EXACT=5", "-ffp-contract=off"]),
    # FAST fuses multiply-adds where the source says fmaf(), nowhere else: with -ffp-contract=fast
    # two inlined copies of one expression (a*b + c*d) may be contracted differently, and the
    # schedules would agree bit for bit only while their code is laid out alike (measured cost of
    # "off": 1-2 % on the sweep kernels, profiles/r02_notes.md)
    ("vm_sweep_kernels.hip", "vm_sweep_kernels_fast.o", ["-DVM_EXACT=0", "-ffp-contract=off"]),
    ("vm_render.hip", "vm_render.o", ["-ffp-contract=off"]),
    ("vm_poisson.hip", "vm_poisson.o", ["-ffp-contract=off"]),
    ("vm_mgb.hip", "vm_mgb.o", ["-ffp-contract=fast"]),
    ("vm_pyramid.hip", "vm_pyramid.o", ["-ffp-contract=off"]),
    ("vm_temporal.hip", "vm_temporal.o", ["-ffp-contract=off"]),
    ("vm_sync.hip", "vm_sync_kernels.o", ["-ffp-contract=off"]),
    ("vm_sync.cpp", "vm_sync.o", ["-x", "hip"]),
    ("vm_video.cpp", "vm_video.o", ["-x", "hip"]),
    ("vm_pyramid_api.cpp", "vm_pyramid_api.o", ["-x", "hip"]),
    ("vm_api.cpp", "vm_api.o", ["-x", "hip"]),
    ("vm_host.cpp", "vm_host.o", ["-x", "hip"]),
    ("vm_frame.cpp", "vm_frame.o", ["-x", "hip"]),
    ("vm_poisson_api.cpp", "vm_poisson_api.o", ["-x", "hip"]),
]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "vmorph.h"))
    headers.append(os.path.abspath(__file__))
    hipcc = _hipcc()
    jobs = []
    for src, obj, extra in UNITS:
        s, o = os.path.join(CSRC, src), os.path.join(OBJ, obj)
        if not os.path.exists(s):
            raise FileNotFoundError(s)
        if force or _newer(o, [s] + headers):
            jobs.append([hipcc] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    objs = [os.path.join(OBJ, o) for _, o, _ in UNITS]
    if force or jobs or _newer(LIB, objs):
        run([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
