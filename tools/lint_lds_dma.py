#!/usr/bin/env python3
"""ISA lint for the kernels that refill LDS by LDS-DMA (global_load_lds_*): compiles the two translation units to assembly and walks
every kernel that contains such an instruction, counting (conservatively, in program order) the ds_reads that can still be outstanding
 (i) where an LDS-DMA is issued and (ii) at every s_barrier.  A DMA overwrites LDS asynchronously: reads of the bytes it replaces must
have RETURNED (s_waitcnt lgkmcnt) before it is issued, or before the barrier that hands the bytes back to another wave -- "the
instructions that use the data have issued" is not a guarantee the compiler keeps (round 4: it sank them below a refill in a variant
of k_scan2, which then lost rows intermittently).
Expected: zero everywhere, except the two kernels whose design reads across the hand-over point on purpose:
  k_scan_wide8  issues the DMAs of the NEXT stage between its two asm pieces while the CURRENT stage's fragment reads are in flight
                (two stages; the refilled one was drained by the explicit lgkmcnt(0) at the end of the previous tile body + a barrier);
  k_gemm8p_tn   issues a phase's fragment reads, takes the phase barrier, waits lgkmcnt(0) first thing behind it; the slot is refilled
                a further barrier later.
    python tools/lint_lds_dma.py        (exit code 1 if any other kernel shows a non-zero count)"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "veritasfi_amd", "csrc")
ALLOW = ("k_scan_wide8", "k_gemm8p_tn")


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("vf_kernels.hip", "vf_transformer.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call([shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-w", "--cuda-device-only",
                                   "-S", "-o", out, os.path.join(CSRC, src)], cwd=CSRC)
            s = open(out).read()
            for m in re.finditer(r"^(_Z[^:\n]*):\s*; @", s, re.M):
                name, i = m.group(1), m.start()
                body = s[i:s.find(".Lfunc_end", i)]
                if "global_load_lds" not in body:
                    continue
                outst = dma_n = bar_n = 0
                for l in (x.strip() for x in body.split("\n")):
                    if l.startswith("ds_read"):
                        outst += 1
                    w = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", l)
                    if w:
                        outst = min(outst, int(w.group(1)))
                    if l.startswith("global_load_lds") and outst:
                        dma_n += 1
                    if l.startswith("s_barrier") and outst:
                        bar_n += 1
                allowed = any(a in name for a in ALLOW)
                flag = "" if not (dma_n or bar_n) else ("  (by design)" if allowed else "  <-- CHECK")
                if (dma_n or bar_n) and not allowed:
                    bad += 1
                print(f"{name[:60]:60s} DMA issued with reads outstanding: {dma_n:3d}   barriers with reads outstanding: {bar_n:3d}{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
