"""Prefill attention kernels at the model's shapes: ViT full (8 x 2116, 16 x 80), ViT window segments, prompt causal (GQA 16:2 x 128)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops
BF = torch.bfloat16
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
a = torch.randn(8192, 8192, device="cuda").to(BF); ops.gemm(a, a); ops.gemm(a, a)
win = ([64] * 25 + [48] * 10 + [36]) * 8
for name, D, H, Hkv, lens, causal in (("vit full", 80, 16, 16, [2116] * 8, False), ("vit window", 80, 16, 16, win, False), ("prompt 3B", 128, 16, 2, [577] * 8, True),
                                      ("prompt 7B", 128, 28, 4, [577] * 8, True)):
    T = sum(lens); cu = [0]
    for l in lens: cu.append(cu[-1] + l)
    qkv = (torch.randn(T, (H + 2 * Hkv) * D, device="cuda") * 0.5).to(BF)
    q, k, v = qkv[:, : H * D], qkv[:, H * D: (H + Hkv) * D], qkv[:, (H + Hkv) * D:]
    out = torch.zeros(T, H * D, device="cuda", dtype=BF)
    cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
    us = t(lambda: ops.attn_varlen(q, k, v, out, cu_t, cu_t, max(lens), H, Hkv, D, causal=causal))
    fl = sum(4.0 * l * l * D * H / (2 if causal else 1) for l in lens)
    print(f"{name:11s}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
