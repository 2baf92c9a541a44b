"""The product's HOST C++ under AddressSanitizer + UBSan on the CPU box (round 5): vf_api.hip's host pass is compiled with
-fsanitize=address,undefined, linked with the regular kernel objects into a small driver (tools/host_sanitize_check.cc) and run:
the .vfc header / payload parser with truncated, oversized, wrapped and randomly mutated headers, the row-range and device-list
checks of the loaders and the sharded handles, and the argument checks of every index / small-dense entry point -- all of it in
front of the first HIP call, so no GPU is needed (and GPU sanitizers are not available on this pool: not attempted).  The oracle's
own sanitizer run is tests/test_oracle_golden.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("hipcc not available")
    from veritasfi_amd import build as vf_build
    vf_build.build_hip()                                                     # the regular objects (kernels, transformer) the driver links
    lib = vf_build.LIBDIR
    objs = [os.path.join(lib, "vf_kernels.o"), os.path.join(lib, "vf_transformer.o")]
    assert all(os.path.exists(o) for o in objs)
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined"]
    api_o = str(tmp_path / "vf_api_san.o")
    subprocess.check_call([hipcc, "-O1", "-g", "-std=c++17", "-fPIC", f"--offload-arch={vf_build.ARCH}", "-Wno-unused-function"] + san +
                          ["-c", os.path.join(vf_build.CSRC, "vf_api.hip"), "-o", api_o], cwd=vf_build.CSRC)
    exe, drv_o = str(tmp_path / "host_sanitize_check"), str(tmp_path / "driver.o")
    clangxx = "/opt/rocm/lib/llvm/bin/clang++"
    # (the driver is plain C++: compiled on its own -- given a .cc beside objects, hipcc applies "-x hip" to the objects as well)
    subprocess.check_call([clangxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-c", os.path.join(ROOT, "tools", "host_sanitize_check.cc"), "-o", drv_o])
    subprocess.check_call([hipcc, f"--offload-arch={vf_build.ARCH}", "-fsanitize=address,undefined", drv_o, api_o] + objs + ["-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    work = tmp_path / "files"
    work.mkdir()
    run = subprocess.run([exe, str(work)], capture_output=True, text=True, env=env, timeout=600)
    print(run.stdout[-2000:], run.stderr[-4000:])
    assert run.returncode == 0, "the sanitized host build reported a problem (see the captured output)"
    assert "host sanitize check: 0 failure(s)" in run.stdout and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
