"""Build libveritasfi_hip.so for gfx950 with hipcc, in-tree (veritasfi_amd/lib/).

``python -m veritasfi_amd.build`` or ``__graft_entry__.build()``.  hipcc cross-compiles without a
GPU, so this runs in the build container; the .so is git-ignored and travels to the GPU box with the
repo snapshot.

Two libraries come out of it:
* ``libveritasfi_hip.so`` -- the product: exports exactly the entry points ``include/veritasfi_hip.h`` declares (a linker version
  script written from the header) plus the two test hooks of the default suite (``KEPT_HOOKS``); kernels that were measured and
  rejected are not compiled in.
* ``libvf_test.so`` (``build_test_variant``) -- the same sources with ``-DVF_EXPERIMENTS``: every ``vf_debug_*`` hook exported (kernel-
  level parity tests drive single kernels through them) and the rejected variants built for A/B runs.  Selected with
  ``VF_LIB_PATH``; ``tests/test_gpu_hooks.py`` runs the hook-driven tests against it in a child process.
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


KEPT_HOOKS = ("vf_debug_force_no_peer", "vf_debug_small_allocs")   # the staged-exchange rehearsal and the arena steady-state test
TEST_LIB = os.path.join(LIBDIR, "libvf_test.so")


def api_symbols() -> list:
    """Entry points declared in include/veritasfi_hip.h, in order of appearance."""
    import re
    text = open(os.path.join(CSRC, HEADERS[1])).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    seen = []
    for name in re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", text):
        if name not in seen:
            seen.append(name)
    return seen


def _version_script(path: str, everything: bool) -> str:
    names = ["vf_*"] if everything else api_symbols() + list(KEPT_HOOKS)
    body = "{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local: *;\n};\n"
    if not os.path.exists(path) or open(path).read() != body:
        with open(path, "w") as f:
            f.write(body)
    return path


def build_test_variant(force: bool = False, verbose: bool = False) -> str:
    """libvf_test.so: -DVF_EXPERIMENTS, every hook exported (a child of this process compiles it, so the module's globals stay put)."""
    env = dict(os.environ, VF_BUILD_LIB=os.path.basename(TEST_LIB), VF_BUILD_TAG="_test",
               VF_BUILD_FLAGS=(os.environ.get("VF_BUILD_FLAGS", "") + " -DVF_EXPERIMENTS").strip(), VF_BUILD_EXPORT_ALL="1")
    cmd = [sys.executable, "-m", "veritasfi_amd.build"] + (["--force"] if force else [])
    out = subprocess.run(cmd, env=env, cwd=os.path.dirname(HERE), capture_output=not verbose, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libvf_test.so failed" + ("" if verbose else ": " + (out.stderr or "")[-2000:]))
    return TEST_LIB


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
    export_all = os.environ.get("VF_BUILD_EXPORT_ALL") == "1" or "-DVF_EXPERIMENTS" in EXTRA
    vs = _version_script(os.path.join(LIBDIR, os.path.basename(LIB) + ".map"), export_all)
    if force or _stale(LIB, objs + [vs]):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={vs}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    if "--with-test-variant" in sys.argv:
        print(build_test_variant(force="--force" in sys.argv, verbose=True))
