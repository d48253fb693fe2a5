"""fp8 x fp8 MFMA GEMM (padt_gemm_fp8) against the bf16 tile GEMM on the prompt-length shapes of PaDT_Pro_7B / 3B, + the activation
quantisation pass.  python tools/bench_gemm_fp8.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from padt_amd import ops  # noqa: E402

BF = torch.bfloat16


def t(fn, reps=10):
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


M = 4616
for (N, K, epi, name) in [(4608, 3584, 0, "7B qkv"), (37888, 3584, 3, "7B gate/up"), (3584, 3584, 2, "7B o"), (3584, 18944, 2, "7B down"),
                          (2560, 2048, 0, "3B qkv"), (22016, 2048, 3, "3B gate/up"), (2048, 11008, 2, "3B down")]:
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    a8, rs = ops.quant_rows_fp8(a)
    w8, ws, _ = ops.quantize_fp8_rows(w)
    fl = 2.0 * M * N * K
    x32 = torch.randn(M, N if epi == 2 else 8, device="cuda")
    xb = x32.to(BF)
    if epi == 2:
        u8 = t(lambda: ops.gemm_fp8(a8, w8, ws, rs, epilogue=2, x32=x32, xb=xb))
        ub = t(lambda: ops.gemm_resid32(a, w, None, x32, xb))
    else:
        out = torch.empty(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
        u8 = t(lambda: ops.gemm_fp8(a8, w8, ws, rs, out=out, epilogue=epi))
        ub = t(lambda: ops.gemm(a, w, out=out, epilogue=epi, row_scale=rs))
    uq = t(lambda: ops.quant_rows_fp8(a, norm_eps=1e-6, out=a8, rs=rs))
    print(f"{name:11s} {M}x{N}x{K}: fp8 {u8:7.1f} us {fl / u8 / 1e6:7.1f} TFLOP/s | bf16 {ub:7.1f} us {fl / ub / 1e6:7.1f} | quantise A {uq:6.1f} us "
          f"| fp8 + quantise vs bf16: {ub / (u8 + uq):.2f}x")
