#!/usr/bin/env python3
"""ISA lint for the kernels that refill LDS by LDS-DMA (global_load_lds_*): compiles the two translation units to assembly and walks
every kernel that contains such an instruction, counting (conservatively) the ds_reads that can still be outstanding (i) where an
LDS-DMA is issued and (ii) at every s_barrier.  The count is a forward data-flow over the kernel's basic blocks (.LBB labels, s_branch /
s_cbranch edges), joined with max() at a label and iterated to a fixed point -- so reads still outstanding at the BOTTOM of a loop
body reach the DMA or barrier at its TOP (the ring-refill hazard; a single pass in program order missed exactly that edge).  The
compile uses the product's own flags (veritasfi_amd.build.FLAGS + VF_BUILD_FLAGS): the ISA linted is the ISA shipped.  A DMA overwrites LDS asynchronously: reads of the bytes it replaces must
have RETURNED (s_waitcnt lgkmcnt) before it is issued, or before the barrier that hands the bytes back to another wave -- "the
instructions that use the data have issued" is not a guarantee the compiler keeps (round 4: it sank them below a refill in a variant
of k_scan2, which then lost rows intermittently).
Expected: zero everywhere, except the kernels whose design reads across the hand-over point on purpose:
  k_scan_wide8  issues the DMAs of the NEXT stage between its two asm pieces while the CURRENT stage's fragment reads are in flight
                (two stages; the refilled one was drained by the explicit lgkmcnt(0) at the end of the previous tile body + a barrier);
  k_gemm8p_tn   issues a phase's fragment reads, takes the phase barrier, waits lgkmcnt(0) first thing behind it; the slot is refilled
                a further barrier later;
  k_attention2  (found by the loop-aware walk, round 5) prefetches the K / V fragments of tiles tn, tn2 and THEN runs the chunk
                boundaries g < tn >> cts: the reads in flight at boundary g belong to chunks > g, the refill behind the barrier
                goes to chunk g -- disjoint rows by construction (vf_transformer.hip, "reads may run ahead of the boundaries").
                The walk counts reads, not addresses, so it cannot see that.
    python tools/lint_lds_dma.py        (exit code 1 if any other kernel shows a non-zero count)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veritasfi_amd import build as vf_build  # noqa: E402
CSRC = vf_build.CSRC
ALLOW = ("k_scan_wide8", "k_gemm8p_tn", "k_attention2")
CAP = 15   # lgkmcnt is a 4-bit counter: "15 or more" is one state, which also bounds the iteration


def blocks_of(body: str):
    """[(label, [instructions], [successor labels], falls_through)] of one function's assembly."""
    blocks, cur = [], ["<entry>", [], [], True]
    for raw in body.split("\n"):
        l = raw.split(";")[0].strip()
        if not l or l.startswith("."):
            m = re.match(r"(\.LBB\d+_\d+):", l)
            if m:
                blocks.append(tuple(cur))
                cur = [m.group(1), [], [], True]
            continue
        if not cur[3]:   # code behind an unconditional branch without a label: unreachable by fall-through, still its own block
            blocks.append(tuple(cur))
            cur = ["<anon>", [], [], True]
        cur[1].append(l)
        b = re.match(r"s_(c?)branch\S*\s+(\.LBB\d+_\d+)", l)
        if b:   # a branch ends its block (the state handed to the target is the state AT the branch)
            cur[2].append(b.group(2))
            falls = bool(b.group(1))
            cur[3] = falls
            if falls:
                blocks.append(tuple(cur))
                cur = ["<anon>", [], [], True]
        elif l.startswith("s_endpgm") or l.startswith("s_setpc"):
            cur[3] = False
    blocks.append(tuple(cur))
    return blocks


def transfer(ins, outst, counts=None, sites=None, label=""):
    for n, l in enumerate(ins):
        if l.startswith("ds_read") or l.startswith("ds_load"):
            outst = min(CAP, outst + 1)
        w = re.match(r"s_waitcnt.*lgkmcnt\((\d+)\)", l)
        if w:
            outst = min(outst, int(w.group(1)))
        elif re.match(r"s_waitcnt\s+(0|0x0)\s*$", l):
            outst = 0
        if counts is not None and outst:
            if l.startswith("global_load_lds"):
                counts[0] += 1
            if l.startswith("s_barrier"):
                counts[1] += 1
            if sites is not None and (l.startswith("global_load_lds") or l.startswith("s_barrier")):
                sites.append((label, n, outst, l))
    return outst


def lint_function(body: str, sites=None):
    """(DMAs issued with reads outstanding, barriers with reads outstanding) over all paths through the function; ``sites`` collects
    (block label, instruction index in the block, reads outstanding, instruction) of every flagged place."""
    bl = blocks_of(body)
    index = {b[0]: i for i, b in enumerate(bl) if b[0].startswith(".LBB")}
    entry = [0] * len(bl)
    changed = True
    while changed:
        changed = False
        for i, (label, ins, succ, falls) in enumerate(bl):
            out = transfer(ins, entry[i])
            for j in ([i + 1] if falls and i + 1 < len(bl) else []) + [index[t] for t in succ if t in index]:
                if out > entry[j]:
                    entry[j], changed = out, True
    counts = [0, 0]
    for i, (label, ins, succ, falls) in enumerate(bl):
        transfer(ins, entry[i], counts, sites, label)
    return counts[0], counts[1]


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("vf_kernels.hip", "vf_transformer.hip"):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call([vf_build._hipcc()] + vf_build.FLAGS + vf_build.EXTRA + ["-w", "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, src)], cwd=CSRC)
            s = open(out).read()
            for m in re.finditer(r"^(_Z[^:\n]*):\s*; @", s, re.M):
                name, i = m.group(1), m.start()
                body = s[i:s.find(".Lfunc_end", i)]
                if "v_mfma" in body and "scratch_" in body and ("k_gemm9_tn" in name or "k_gemm8p_tn" in name):
                    # kernels whose scratch is tolerated because it sits outside the matrix loop (tests/test_kernel_resources.py): say so
                    lines = body.split("\n")
                    mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
                    inside = sum(1 for l in lines[mf[0]:mf[-1] + 1] if re.match(r"\s*scratch_(load|store)", l))
                    total = sum(1 for l in lines if re.match(r"\s*scratch_(load|store)", l))
                    print(f"{name[:60]:60s} scratch accesses between the first and the last matrix instruction: {inside:3d} (of {total}){'  <-- CHECK' if inside else ''}")
                    bad += 1 if inside else 0
                if "global_load_lds" not in body:
                    continue
                sites = [] if "-v" in sys.argv else None
                dma_n, bar_n = lint_function(body, sites)
                allowed = any(a in name for a in ALLOW)
                flag = "" if not (dma_n or bar_n) else ("  (by design)" if allowed else "  <-- CHECK")
                if (dma_n or bar_n) and not allowed:
                    bad += 1
                print(f"{name[:60]:60s} DMA issued with reads outstanding: {dma_n:3d}   barriers with reads outstanding: {bar_n:3d}{flag}")
                for site in sites or []:
                    print("      block %s instruction %d: %d outstanding at  %s" % site)
            if "--keep" in sys.argv:
                import shutil
                shutil.copy(out, os.path.join("/tmp", src + ".s"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
