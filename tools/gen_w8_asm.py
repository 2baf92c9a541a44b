#!/usr/bin/env python3
"""Generates the asm K-tile bodies of k_scan_wide8 (fixed registers; veritasfi_amd/csrc/vf_kernels.hip carries the output verbatim
between '#define VF8_ASM_E0_A' and the kernel; tests/test_abi_and_host.py checks that they agree).  usage: gen_w8_asm.py > block.inc"""
ACC = {(m, nt): f"v[{64*m+16*nt}:{64*m+16*nt+15}]" for m in range(2) for nt in range(4)}
def slot(i):
    b = 128 + 16 * i
    return dict(q=[f"v[{b+4*k}:{b+4*k+3}]" for k in range(4)], lo=f"v[{b}:{b+7}]", hi=f"v[{b+8}:{b+15}]", all=f"v[{b}:{b+15}]")
S = [slot(i) for i in range(5)]
def read(slot_i, a0, a1, off, second):   # second: offset of the slot's second operand (A: row tile m = 1 at +2048; B: the lo codes at +16384)
    s = S[slot_i]
    return [f"ds_read_b128 {s['q'][0]}, {a0} offset:{off}", f"ds_read_b128 {s['q'][1]}, {a1} offset:{off}",
            f"ds_read_b128 {s['q'][2]}, {a0} offset:{off+second}", f"ds_read_b128 {s['q'][3]}, {a1} offset:{off+second}"]
def mm(nt, a_slot, b_slot, zero=False):   # zero: the first K-tile of a super-tile starts the sums (C = 0: no clearing of the accumulators)
    A, B = S[a_slot], S[b_slot]
    def one(m, asrc, bsrc, sc, c0):
        return f"v_mfma_scale_f32_32x32x64_f8f6f4 {ACC[(m, nt)]}, {asrc}, {bsrc}, {'0' if c0 else ACC[(m, nt)]}, %[sa], {sc} op_sel_hi:[0,0,0]"
    return [one(0, A['lo'], B['lo'], "%[sh]", zero), one(1, A['hi'], B['lo'], "%[sh]", zero), one(0, A['lo'], B['hi'], "%[sl]", False), one(1, A['hi'], B['hi'], "%[sl]", False)]
def body(even, has_prev, is_last, first=False, second=False):   # first / second: K-tile 0 / 1 of a super-tile
    cur, prv = (1, 0) if even else (0, 1)       # slot of this tile's A / of the previous tile's A
    L = []
    rA = read(cur, "%[pa0]", "%[pa1]", 0, 2048)
    rB0 = read(2, "%[pb0]", "%[pb1]", 0, 16384)
    rB1 = read(3, "%[pb0]", "%[pb1]", 2048, 16384)
    rB2 = read(prv, "%[pb0]", "%[pb1]", 4096, 16384)
    rB3 = read(4, "%[pb0]", "%[pb1]", 6144, 16384)
    if has_prev:
        d = mm(3, prv, 4, zero=second)
        L += [d[0]] + rA + [d[1]] + rB0 + ["@SPLIT"] + [d[2]] + rB1 + [d[3]]
    else:
        L += rA + rB0 + ["@SPLIT"] + rB1
    L += rB2 + rB3
    L += ["s_waitcnt lgkmcnt(12)"] + mm(0, cur, 2, first) + ["s_waitcnt lgkmcnt(8)"] + mm(1, cur, 3, first) + ["s_waitcnt lgkmcnt(4)"] + mm(2, cur, prv, first)
    L += ["s_waitcnt lgkmcnt(0)"]
    if is_last:
        # compiler-generated VALU code reads the accumulators right behind this body, and LLVM's hazard recognizer does not look
        # inside inline asm: the 16-pass matrix result -> VALU read wait states are spelled out (they cost 24 cycles per super-tile)
        L += mm(3, cur, 4) + ["s_nop 15", "s_nop 7"]
    return L
def cstr(lines):
    return "\n".join(f'        "{l}\\n\\t"' for l in lines)
out = []
for name, args in (("VF8_ASM_E0", (True, False, False, True, False)), ("VF8_ASM_O1", (False, True, False, False, True)), ("VF8_ASM_EM", (True, True, False)),
                   ("VF8_ASM_OM", (False, True, False)), ("VF8_ASM_OL", (False, True, True))):
    lines = body(*args)
    k = lines.index("@SPLIT")
    for suffix, part in (("_A", lines[:k]), ("_B", lines[k + 1:])):
        out.append(f"#define {name}{suffix} \\\n" + " \\\n".join(f'    "{l}\\n\\t"' for l in part) + "\n")
print("\n".join(out))

