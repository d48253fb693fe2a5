"""Micro-benchmark of the varlen attention / rope / rmsnorm kernels at the PaDT_Pro_3B REC batch-8 shapes.
usage: python tools/bench_attn.py     (PADT_HIP_LIB selects the library build → A/B inside one GPU session)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops
from tools.bench_kernels import timeit

BF = torch.bfloat16


def main():
    dev = "cuda"
    # ViT: 8 images x 2116 patches (grid 46x46), 16 heads x 80
    P, H, hd = 8 * 2116, 16, 80
    qkv = (torch.randn(P, 3 * H * hd, device=dev) * 0.5).to(BF)
    out = torch.empty(P, H * hd, device=dev, dtype=BF)
    cu_full = torch.arange(0, P + 1, 2116, dtype=torch.int32, device=dev)
    cu_win = torch.arange(0, P + 1, 64, dtype=torch.int32, device=dev)
    vh = H * hd
    for name, cu, mx in (("vit_full", cu_full, 2116), ("vit_win", cu_win, 64)):
        t = timeit(lambda: ops.attn_varlen(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], out, cu, cu, mx, H, H, hd), 50)
        nseg = cu.numel() - 1
        fl = 4.0 * nseg * mx * mx * H * hd
        print(f"{name:10s}: {t:8.1f} us  {fl / t * 1e-6:7.1f} TFLOP/s")
    # LLM prefill: 8 x 577 tokens, 16 q heads / 2 kv heads x 128, causal
    L, B, Hq, Hk, D = 577, 8, 16, 2, 128
    T = B * L
    q = (torch.randn(T, Hq * D, device=dev) * 0.5).to(BF)
    k = (torch.randn(T, Hk * D, device=dev) * 0.5).to(BF)
    v = (torch.randn(T, Hk * D, device=dev) * 0.5).to(BF)
    o = torch.empty(T, Hq * D, device=dev, dtype=BF)
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.attn_varlen(q, k, v, o, cu, cu, L, Hq, Hk, D, causal=True), 100)
    print(f"{'prefill':10s}: {t:8.1f} us  {2.0 * B * L * L * Hq * D / t * 1e-6:7.1f} TFLOP/s")
    cos = torch.randn(P, hd, device=dev)
    sin = torch.randn(P, hd, device=dev)
    t = timeit(lambda: ops.rope_half_(qkv, cos, sin, 2 * H, hd), 100)
    print(f"{'rope_half':10s}: {t:8.1f} us  {(2 * P * 2 * vh * 2 + 2 * P * hd * 4) / t * 1e-3:7.1f} GB/s")
    x = torch.randn(P, vh, device=dev).to(BF)
    w = torch.ones(vh, device=dev, dtype=BF)
    n = torch.empty_like(x)
    t = timeit(lambda: ops.rmsnorm(x, w, out=n), 100)
    print(f"{'rmsnorm':10s}: {t:8.1f} us  {2 * P * vh * 2 / t * 1e-3:7.1f} GB/s")


if __name__ == "__main__":
    main()
