#!/usr/bin/env python3
"""Attention micro-bench / parity through the vf_debug_attention test hook (dh = 64).
qkv [B*T][3*64*heads] fp16 with Q already carrying log2(e)/8 (as the encoder's folded query projection produces it);
kind 1 = first-generation resident kernel, 2 = k_attention2, 3 = streaming kernel.  Reference: fp32 softmax in torch."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from veritasfi_amd import _ffi


def make_case(B, T, heads, dev, seed=1, scale=1.0, ragged=False, growing=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    H = heads * 64
    qkv = torch.randn(B * T, 3 * H, device=dev, generator=g) * scale
    if growing:
        # key norms grow along the sequence: the running maximum keeps moving (exercises the reference-move path)
        ramp = torch.linspace(0.2, 3.0, T, device=dev).repeat(B).unsqueeze(1)
        qkv[:, H:2 * H] *= ramp
    qkv[:, :H] *= 0.125 * 1.4426950408889634
    qkv = qkv.half()
    mask = torch.ones(B, T, dtype=torch.int32, device=dev)
    if ragged:
        lens = torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(seed))
        for i, n in enumerate(lens.tolist()):
            mask[i, n:] = 0
        if B > 2:
            mask[1, :] = 0          # a sequence without any valid key
            mask[2, :] = 0
            mask[2, T // 2] = 1     # a single valid key in the middle
    return qkv, mask.reshape(-1).contiguous()


def reference(qkv, mask, B, T, heads, rows=None):
    H = heads * 64
    x = qkv.float().reshape(B, T, 3, heads, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)   # [B][heads][T][64]
    if rows is not None:
        q, k, v = q[:rows], k[:rows], v[:rows]
        mask = mask.reshape(B, T)[:rows]
    else:
        mask = mask.reshape(B, T)
    s = q @ k.transpose(-1, -2) * 0.6931471805599453          # back to natural units
    s = s + (mask[:, None, None, :] == 0).float() * -30000.0
    p = torch.softmax(s, dim=-1)
    o = p @ v
    return o.transpose(1, 2).reshape(-1, H)


def run(L, kind, qkv, mask, B, T, heads, ctx):
    rc = L.vf_debug_attention(qkv.data_ptr(), mask.data_ptr(), B, T, heads, ctx.data_ptr(),
                              torch.cuda.current_stream().cuda_stream, kind)
    assert rc == 0, rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="100x512x12,100x256x12,256x128x12,64x512x16")
    ap.add_argument("--kinds", default="1,2,3")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ragged", type=int, default=0)
    a = ap.parse_args()
    L = _ffi.lib()
    L.vf_debug_attention.restype = ctypes.c_int
    L.vf_debug_attention.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for sh in a.shapes.split(","):
        B, T, heads = map(int, sh.split("x"))
        qkv, mask = make_case(B, T, heads, dev, ragged=bool(a.ragged))
        rows = min(B, 8)
        ref = reference(qkv, mask, B, T, heads, rows=rows)
        for kind in [int(x) for x in a.kinds.split(",")]:
            ctx = torch.zeros(B * T, heads * 64, dtype=torch.float16, device=dev)
            run(L, kind, qkv, mask, B, T, heads, ctx)
            torch.cuda.synchronize()
            valid = mask.reshape(B, T)[:rows].reshape(-1).bool()
            err = float((ctx[:rows * T].float() - ref)[valid].abs().max())
            e0.record()
            for _ in range(a.iters):
                run(L, kind, qkv, mask, B, T, heads, ctx)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            flops = 4.0 * B * heads * T * T * 64
            print(json.dumps({"shape": sh, "kind": kind, "ragged": a.ragged, "us": round(us, 1), "tflops": round(flops / us / 1e6, 1),
                              "max_err": err}), flush=True)


if __name__ == "__main__":
    main()
