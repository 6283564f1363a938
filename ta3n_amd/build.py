"""Builds libta3n_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built
.so travels with the repo snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# TA3N_LIBDIR: another build directory (A/B builds with other -D flags, e.g. ta3n_amd/lib_ab/ built with TA3N_EXTRA_FLAGS="-DTA3N_DMA_INTERLEAVE=0");
# _lib.py loads from the same place
LIBDIR = os.environ.get("TA3N_LIBDIR") or os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libta3n_hip.so")
SOURCES = ["ta3n_api.hip", "ta3n_gemm.hip", "ta3n_gemm_i0.hip", "ta3n_gemm_i1.hip", "ta3n_gemm_i2.hip", "ta3n_gemm_i3.hip", "ta3n_gemm_i4.hip", "ta3n_gemm_i5.hip", "ta3n_pointwise.hip", "ta3n_heads.hip", "ta3n_comm.hip", "ta3n_peer.hip", "ta3n_mmd.hip", "ta3n_plan.cpp", "ta3n_index.cpp"]
HEADERS = ["ta3n_types.h", "ta3n_kernels.h", "ta3n_plan.h", "ta3n_gemm_kernel.h", os.path.join("..", "..", "include", "ta3n_hip.h")]


COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel / plan sources and headers: identifies the binary a measurement was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build(extra_flags=()) -> bool:
    if not os.path.exists(LIB):
        return True
    try:                                  # another set of -D flags than the library was built with (A/B build directories)
        with open(os.path.join(LIBDIR, ".flags")) as f:
            if f.read() != " ".join([*COMMON_FLAGS, *extra_flags]):
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    """Compile what is out of date (an object is rebuilt when its source or any header is newer), in parallel, and link."""
    if not force and not needs_build(extra_flags):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    common = list(COMMON_FLAGS)
    t_hdr = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    flags_file = os.path.join(LIBDIR, ".flags")
    flags_now = " ".join([*common, *extra_flags])
    try:
        with open(flags_file) as f:
            same_flags = f.read() == flags_now
    except OSError:
        same_flags = False
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), t_hdr):
            jobs.append([hipcc, *common, *extra_flags, "-x", "hip", "-c", path, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    with open(flags_file, "w") as f:
        f.write(flags_now)
    run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB, *objs, "-ldl"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=tuple(os.environ.get("TA3N_EXTRA_FLAGS", "").split()))
    print(LIB)
