"""Build libveritasfi_hip.so for gfx950 with hipcc, in-tree (veritasfi_amd/lib/).

``python -m veritasfi_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU, so this runs in the build container; the .so is git-ignored and travels to the GPU box with the
repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, os.environ.get("VF_BUILD_LIB", "libveritasfi_hip.so"))   # VF_BUILD_LIB: name of an A/B variant
SOURCES = ["vf_kernels.hip", "vf_api.hip", "vf_transformer.hip"]
HEADERS = ["vf_internal.h", os.path.join("..", "..", "include", "veritasfi_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# VF_BUILD_FLAGS="-DVF_EXPERIMENTS" compiles the measured-and-rejected experiment kernels back in
EXTRA = os.environ.get("VF_BUILD_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built (no fallback exists)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for s in srcs:   # the three translation units compile concurrently (the transformer TU alone takes ~40 s)
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", os.environ.get("VF_BUILD_TAG", "") + ".o"))
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + EXTRA + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    failed = [cmd for cmd, proc in jobs if proc.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
