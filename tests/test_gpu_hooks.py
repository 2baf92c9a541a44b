"""The kernel-level tests that need test hooks, against the test build of the library.

The product library (libveritasfi_hip.so) exports the C ABI and two hooks; the tests that launch ONE kernel through ``vf_debug_gemm`` /
``vf_debug_attention`` / ..., toggle a dispatch rule, or ask for a measured-and-rejected variant (stream-K, the 4-wave wide scan, the
LayerNorm fold) are skipped in the main process (tests/conftest.py) and run HERE: one child pytest bound to ``libvf_test.so``
(``VF_LIB_PATH``, built by ``__graft_entry__.build()`` with -DVF_EXPERIMENTS) that collects only those tests."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(ROOT, "veritasfi_amd", "lib", "libvf_test.so")


def test_product_library_exports_the_header_and_two_hooks_only():
    """nm -D of the product library: the header's entry points + KEPT_HOOKS, nothing else (no C++ internals, no other vf_debug_*)."""
    from veritasfi_amd import build as B
    out = subprocess.run(["nm", "-D", "--defined-only", B.LIB], capture_output=True, text=True, check=True).stdout
    got = sorted(line.split()[-1] for line in out.splitlines() if " T " in line)
    assert got == sorted(B.api_symbols() + list(B.KEPT_HOOKS)), sorted(set(got) ^ set(B.api_symbols() + list(B.KEPT_HOOKS)))
    assert len(got) <= 60


@pytest.mark.gpu
def test_hook_driven_kernel_tests_pass_on_the_test_build():
    assert os.path.exists(TEST_LIB), "libvf_test.so is missing: __graft_entry__.build() (veritasfi_amd.build.build_test_variant) makes it"
    env = dict(os.environ, VF_LIB_PATH=TEST_LIB, VF_HOOK_TESTS_ONLY="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
           "--deselect", "tests/test_gpu_hooks.py"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=840)
    tail = "\n".join((out.stdout or "").splitlines()[-25:])
    print(tail)
    assert out.returncode == 0, tail + "\n" + (out.stderr or "")[-1500:]
    m = re.search(r"(\d+) passed", out.stdout)
    assert m and int(m.group(1)) >= 40, tail      # the hook tests exist and ran (none skipped for want of hooks)
    assert " skipped" not in tail.splitlines()[-1], tail
