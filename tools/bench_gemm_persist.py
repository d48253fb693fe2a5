"""SwiGLU tile GEMMs (ViT / LLM gate-up shapes + ragged ones) under the current PADT_GEMM_PERSIST setting: a digest of every output (the persistent launch must be
bit-identical to the one-tile-per-block launch) and us per launch with rotating A operands.  Run once per setting and compare.  python tools/bench_gemm_persist.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from padt_amd import ops  # noqa: E402

H = torch.float16


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("PADT_GEMM_PERSIST =", os.environ.get("PADT_GEMM_PERSIST", "0"))
g = torch.Generator(device="cpu").manual_seed(1)
warm = torch.randn(8192, 8192, device="cuda").to(H)
for _ in range(20):
    warm @ warm                                                     # clocks up before the first timed shape
for (M, N, K, bias, rs) in [(16928, 6912, 1280, True, True), (4616, 22016, 2048, False, True), (4616, 22016, 2048, False, False), (2116, 6912, 1280, True, True),
                            (1000, 1344, 640, True, False), (16928 + 40, 6912, 1280, True, True), (5000, 37888, 3584, False, True)]:
    a = [(torch.randn(M, K, generator=g) * 0.5).cuda().to(H) for _ in range(3)]
    w = (torch.randn(N, K, generator=g) * 0.03).cuda().to(H)
    b = (torch.randn(N, generator=g) * 0.1).cuda().to(H) if bias else None
    r = (torch.rand(M, generator=g) + 0.5).cuda() if rs else None
    out = torch.zeros(M, N // 2, device="cuda", dtype=H)
    ops.gemm(a[0], w, b, out=out, epilogue=ops.EPI_SWIGLU, row_scale=r)
    torch.cuda.synchronize()
    dig = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    i = [0]

    def f():
        i[0] += 1
        ops.gemm(a[i[0] % 3], w, b, out=out, epilogue=ops.EPI_SWIGLU, row_scale=r)
    u = t(f)
    print(f"{M:6d} x {N:6d} x {K:5d} bias={int(bias)} rs={int(rs)}: {u:8.1f} us {2.0 * M * N * K / u / 1e6:7.1f} TFLOP/s  sha1 {dig}")
