"""Builds libta3n_hip.so in-tree with hipcc for gfx950 (no JIT cache: the built
.so travels with the repo snapshot to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# TA3N_LIBDIR: another build directory (A/B builds with other -D flags, e.g. ta3n_amd/lib_ab/ built with TA3N_EXTRA_FLAGS="-DTA3N_EXPERIMENTS=1");
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


HASH_FILE = os.path.join(LIBDIR, ".source_hash")


def needs_build(extra_flags=()) -> bool:
    """True unless the library on disk was linked from exactly these sources with exactly these flags: the content hash of every
    source and header (source_hash) and the flag string are stored beside the .so at link time and compared here - file times say
    nothing about a binary that was copied in, or about a checkout that reset them."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(os.path.join(LIBDIR, ".flags")) as f:
            if f.read() != " ".join([*COMMON_FLAGS, *extra_flags]):
                return True
        with open(HASH_FILE) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True, extra_flags=()) -> str:
    """Compile what is out of date (an object is rebuilt when its source or any header is newer), in parallel, and link."""
    if not force and not needs_build(extra_flags):
        if verbose:
            print(f"[build] reused {LIB}: source hash {source_hash()} and flags match the ones it was linked from", flush=True)
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    common = list(COMMON_FLAGS)
    flags_file = os.path.join(LIBDIR, ".flags")
    flags_now = " ".join([*common, *extra_flags])
    import hashlib
    hdr = hashlib.sha256()
    for h_ in sorted(HEADERS):
        with open(os.path.join(CSRC, h_), "rb") as fh:
            hdr.update(fh.read())
    objs, jobs, stamps = [], [], {}
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        path = os.path.join(CSRC, src)
        # an object is current when the hash of (its source, every header, the flags) equals the one stored beside it at compile time
        with open(path, "rb") as fh:
            want = hashlib.sha256(fh.read() + hdr.digest() + flags_now.encode()).hexdigest()
        try:
            with open(obj + ".hash") as fh:
                have = fh.read().strip()
        except OSError:
            have = None
        if force or not os.path.exists(obj) or have != want:
            jobs.append([hipcc, *common, *extra_flags, "-x", "hip", "-c", path, "-o", obj])
            stamps[obj] = want

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        obj = cmd[-1]
        if obj in stamps and os.path.exists(obj + ".hash"):
            os.remove(obj + ".hash")
        subprocess.check_call(cmd)
        if obj in stamps:
            with open(obj + ".hash", "w") as fh:
                fh.write(stamps[obj])
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    with open(flags_file, "w") as f:
        f.write(flags_now)
    if os.path.exists(HASH_FILE):
        os.remove(HASH_FILE)               # (a failed link must not leave a hash that vouches for the old binary)
    h = source_hash()                      # before the link: a source edited meanwhile makes the next call rebuild
    run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB, *objs, "-ldl"])
    with open(HASH_FILE, "w") as f:
        f.write(h)
    if verbose:
        print(f"[build] rebuilt {LIB} ({len(jobs)} of {len(SOURCES)} objects recompiled), source hash {h}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, extra_flags=tuple(os.environ.get("TA3N_EXTRA_FLAGS", "").split()))
    print(LIB)
