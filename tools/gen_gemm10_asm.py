#!/usr/bin/env python3
"""Generates the asm K-tile bodies of k_gemm10_tn (fixed registers; veritasfi_amd/csrc/vf_transformer.hip carries the output verbatim
between '#define VFX_ASM_F_A' and the kernel; tests/test_abi_and_host.py checks that they agree).  usage: gen_gemm10_asm.py > block.inc

Registers: accumulators acc[ni][mi] = v[(2 ni + mi) 16 : +15] (v[0:127]); four fragment slots F0 .. F3 of 24 registers at v[128 + 24 k]:
W fragments of the four column tiles (4 x 4 registers), then A fragments of the two row tiles (2 x 4).  k-step ks of a K-tile (16 of
its 64 elements) reads into F[ks]; the eight MFMAs of k-step 3 are HELD BACK and issued behind the next K-tile's barrier, under that
tile's first fragment reads and the wave's DMA issue (piece _A | C++ DMA issue | piece _B).
LDS: A stage s at 32768 s, W stage s at 65536 + 32768 s: one address register per (operand, k-step), stage and tile by offset."""
def acc(ni, mi): b = (2 * ni + mi) * 16; return f"v[{b}:{b + 15}]"
def wf(k, ni): b = 128 + 24 * k + 4 * ni; return f"v[{b}:{b + 3}]"
def af(k, mi): b = 128 + 24 * k + 16 + 4 * mi; return f"v[{b}:{b + 3}]"
def reads(ks, stage):
    L = [f"ds_read_b128 {wf(ks, ni)}, %[pw{ks}] offset:{stage * 32768 + ni * 4096}" for ni in range(4)]
    L += [f"ds_read_b128 {af(ks, mi)}, %[pa{ks}] offset:{stage * 32768 + mi * 4096}" for mi in range(2)]
    return L
def mfmas(ks, zero=False):
    return [f"v_mfma_f32_32x32x16_f16 {acc(ni, mi)}, {wf(ks, ni)}, {af(ks, mi)}, {'0' if zero else acc(ni, mi)}" for ni in range(4) for mi in range(2)]
def body(stage, first, last):
    A, B = [], []
    if first:      # K-tile 0 of a tile: nothing held back in front of it; its first k-step STARTS the sums (C = 0)
        A += reads(0, stage) + reads(1, stage)
        B += ["s_waitcnt lgkmcnt(6)"] + mfmas(0, zero=True)
    else:
        hb = mfmas(3)
        A += [hb[0]] + reads(0, stage) + [hb[1]] + reads(1, stage)
        B += hb[2:] + ["s_waitcnt lgkmcnt(6)"] + mfmas(0)
    B += reads(2, stage) + ["s_waitcnt lgkmcnt(6)"] + mfmas(1) + reads(3, stage) + ["s_waitcnt lgkmcnt(6)"] + mfmas(2) + ["s_waitcnt lgkmcnt(0)"]
    if last:       # the tile's last K-tile runs its own k-step 3; the epilogue reads the accumulators: let the pipe drain
        B += mfmas(3) + ["s_nop 15"] * 5
    return A, B
out = []
for name, args in (("VFX_ASM_F", (0, True, False)), ("VFX_ASM_E", (0, False, False)), ("VFX_ASM_O", (1, False, False)), ("VFX_ASM_L", (1, False, True))):
    a, b = body(*args)
    for suffix, part in (("_A", a), ("_B", b)):
        out.append(f"#define {name}{suffix} \\\n" + " \\\n".join(f'    "{l}\\n\\t"' for l in part) + "\n")
print("\n".join(out))
