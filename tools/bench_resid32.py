"""Residual GEMMs of the ViT / prefill: fp32 residual stream (padt_gemm_resid32: fp32 in place + bf16 mirror) against the bf16-stream
epilogue, per shape (us per call, TFLOP/s).  python tools/bench_resid32.py"""
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from padt_amd import ops  # noqa: E402

BF = torch.bfloat16


def t(fn, reps=20):
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


for (M, N, K, name) in [(16928, 1280, 1280, "vit proj"), (16928, 1280, 3456, "vit down"), (4616, 2048, 2048, "llm o"), (4616, 2048, 11008, "llm down")]:
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    b = torch.zeros(N, device="cuda", dtype=BF)
    x32 = torch.randn(M, N, device="cuda")
    xb = x32.to(BF)
    fl = 2.0 * M * N * K
    u32 = t(lambda: ops.gemm_resid32(a, w, b, x32, xb))
    ubf = t(lambda: ops.gemm(a, w, b, out=xb, epilogue=ops.EPI_RESID, residual=xb))
    upl = t(lambda: ops.gemm(a, w, b, out=xb))
    print(f"{name:9s} {M}x{N}x{K}: fp32 stream {u32:7.1f} us {fl / u32 / 1e6:7.1f} TFLOP/s | bf16 stream {ubf:7.1f} us {fl / ubf / 1e6:7.1f} | no residual {upl:7.1f} us {fl / upl / 1e6:7.1f}")
