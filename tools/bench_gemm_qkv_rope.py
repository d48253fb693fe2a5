"""The ViT qkv GEMM (16928 x 3840 x 1280, fp16 operands) with its fused epilogue pieces switched on one by one: plain 16-bit output, + bias, + per-row
scale (folded RMSNorm rstd), + RoPE of the q / k columns — what each costs next to the SwiGLU gate/up GEMM of the same K.  python tools/bench_gemm_qkv_rope.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from padt_amd import ops  # noqa: E402

H = torch.float16


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


M, N, K, hd = 16928, 3840, 1280, 80
a = [torch.randn(M, K, device="cuda").to(H) for _ in range(4)]                       # rotate A: 43 MB each
w = (torch.randn(N, K, device="cuda") * 0.02).to(H)
b = (torch.randn(N, device="cuda") * 0.1).to(H)
out = torch.empty(M, N, device="cuda", dtype=H)
rs = torch.rand(M, device="cuda") + 0.5
cos, sin = torch.randn(M, hd // 2, device="cuda"), torch.randn(M, hd // 2, device="cuda")
i = [0]


def A():
    i[0] += 1
    return a[i[0] % 4]


fl = 2.0 * M * N * K
for name, fn in (("plain", lambda: ops.gemm(A(), w, None, out=out)),
                 ("+ bias", lambda: ops.gemm(A(), w, b, out=out)),
                 ("+ bias + row scale", lambda: ops.gemm(A(), w, b, out=out, row_scale=rs)),
                 ("+ bias + row scale + rope", lambda: ops.gemm_rope(A(), w, b, out, cos, sin, 2 * 1280, hd, row_scale=rs))):
    u = t(fn)
    print(f"qkv {name:28s}: {u:7.1f} us  {fl / u / 1e6:7.1f} TFLOP/s")
wg = (torch.randn(6912, K, device="cuda") * 0.02).to(H)
og = torch.empty(M, 3456, device="cuda", dtype=H)
u = t(lambda: ops.gemm(A(), wg, None, out=og, epilogue=ops.EPI_SWIGLU, row_scale=rs))
print(f"gate/up SwiGLU + row scale      : {u:7.1f} us  {2.0 * M * 6912 * K / u / 1e6:7.1f} TFLOP/s")
