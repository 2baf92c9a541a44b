"""No shipped scan / product / attention kernel may touch scratch memory unless it is named here with a reason.

hipcc's own remarks (-Rpass-analysis=kernel-resource-usage, device pass of the product's flags; tools/resource_usage.py) are the
source: a kernel with hand-counted s_waitcnt and LDS-DMA queues cannot afford a compiler-placed scratch access -- it lands in
the in-order vector-memory queue, so the wait in front of its use is a vmcnt(0) that drains every prefetch (and a reload in front
of a hand-written wait was a silent-wrong-result mechanism in round 3).  hipcc cross-compiles: no GPU needed.
"""
import concurrent.futures
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (as tools/resource_usage.py prints it; a regex) -> why its scratch is tolerated
ALLOW = {
    r"k_gemm8p_tn<\d+>": "first-generation 8-phase product, kept for split-K tails and K > 2048: 24 registers around the phase "
                         "loop (256 accumulators + fragments); no scratch access inside the K loop (checked in round 4's ISA)",
    r"k_gemm9_tn<\d+,1>": "the persistent kernel's whole-product K cut: the 24 registers are sk_coop_finish's (two blocks of partners' partials "
                           "beside the 128 accumulators), every scratch access sits behind the main loop (ISA checked, round 5)",
    r"k_scan<(2,4,1,0|2,4,0,0)>": "k_scan on fp16 rows with 4 segments per stage at 64 queries: only reached when k_scan2's LDS "
                                   "image does not fit (d > 1216); register-staged loads, compiler-counted waits "
                                   "(the 3-segment and 32-query forms lost their scratch with round 6's epilogue)",
}
HOT = re.compile(r"k_scan|k_gemm|k_attention|k_final|k_sel0|k_layernorm|k_embed|k_pool|k_dec|k_rms|k_rope|k_swiglu|k_vit|k_clip")


def _tool():
    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(ROOT, "tools", "resource_usage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_demangler_names_kernels_and_integral_template_arguments():
    ru = _tool()
    assert ru.demangle(["_ZN3vft11k_gemm8p_tnILi0EEEvPKDF16_S2_PKfS2_PDF16_iiiNS_6LnFuseE", "_ZN2vf12k_scan_wide8ILi4EEEvNS_8ScanArgsE",
                        "_ZN3vft19k_attention_stream2ILi256ELb1EEEvPKDF16_PKiiiiifPDF16_", "_ZN2vf6k_scanILi2ELi4ELi0ELi0EEEvNS_8ScanArgsE"]) == \
        ["k_gemm8p_tn<0>", "k_scan_wide8<4>", "k_attention_stream2<256,true>", "k_scan<2,4,0,0>"]


def test_no_hot_kernel_uses_scratch_unless_allow_listed():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    ru = _tool()
    from veritasfi_amd import build as vf_build
    srcs = [os.path.join(vf_build.CSRC, s) for s in vf_build.SOURCES]
    with concurrent.futures.ThreadPoolExecutor(len(srcs)) as ex:
        kernels = [k for ks in ex.map(ru.usage, srcs) for k in ks]
    assert len(kernels) > 60, "the remarks were not parsed"
    names = {k["pretty"] for k in kernels}
    for name in ("k_attention_stream2<256,true>", "k_gemm9_tn<2,0>"):   # once allow-listed, now clean: keep them clean
        assert next(k for k in kernels if k["pretty"] == name).get("scratch", 0) == 0, name
    for must in ("k_scan_wide8<8>", "k_scan2<2,0>", "k_gemm9_tn<0,0>", "k_attention2<0>", "k_final"):
        assert must in names, must
    # measured-and-rejected variants are not in the product build (round 6: -DVF_EXPERIMENTS / libvf_test.so carries them)
    for gone in ("k_scan_wide8<4>", "k_gemm9_tn<0,2>", "k_gemm9_tn<1,2>", "k_gemm8p_tn<7>", "k_gemm8p_tn<8>", "k_gemm8p_tn<10>"):
        assert gone not in names, gone
    bad, used = [], set()
    for k in kernels:
        if not HOT.match(k["pretty"]):
            continue
        if k.get("scratch", 0) > 0 or k.get("vgpr_spill", 0) > 0 or str(k.get("dynamic_stack", "False")) == "True":
            pat = next((p for p in ALLOW if re.fullmatch(p, k["pretty"])), None)
            if pat is None:
                bad.append((k["pretty"], k.get("scratch"), k.get("vgpr_spill")))
            else:
                used.add(pat)
    assert not bad, f"kernels with scratch that are not allow-listed: {bad}"
    assert used == set(ALLOW), f"allow-list entries that no longer apply (remove them): {set(ALLOW) - used}"
    w8 = next(k for k in kernels if k["pretty"] == "k_scan_wide8<8>")   # one 8-wave workgroup per CU: two waves per SIMD
    assert w8["scratch"] == 0 and w8["vgpr_spill"] == 0 and w8["occupancy"] == 2, w8
