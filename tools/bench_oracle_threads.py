"""How many host threads should the fp32 CPU oracle use on the GPU box?  (tests/test_real_shape_gpu.py runs it on 32; the box has 256 cores;
the oracle is ≈ 85 % of the GPU suite's 9.2 minutes.)  Times the two GEMM shapes that dominate a PaDT_Pro_3B oracle pass.

Measured at the end of round 4 (256 cores, torch default 128 threads): 32 threads 1.56 / 1.40 TFLOP/s (gate/up / ViT qkv), 64 threads 1.27 / 1.00,
128 threads 0.84 / 0.62, 192 threads 0.62 / 0.42; an 8-row decode-step GEMM 2.4 ms at 32-64 threads, 33 ms at 128, 168 ms at 192.  More
threads are SLOWER: 32 stays (fewer was not measured)."""
import os
import time

import torch

print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
g = torch.Generator().manual_seed(0)
shapes = (("llm gate/up 4616x2048 @ 22016", 4616, 2048, 22016), ("vit qkv 16928x1280 @ 3840", 16928, 1280, 3840), ("decode step 8x2048 @ 22016", 8, 2048, 22016))
mats = [(n, torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)) for n, M, K, N in shapes]
for nt in (32, 64, 128, 192):
    torch.set_num_threads(nt)
    line = [f"{nt:3d} threads:"]
    for name, a, w in mats:
        torch.nn.functional.linear(a, w)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            torch.nn.functional.linear(a, w)
        dt = (time.perf_counter() - t0) / reps
        line.append(f"{name} {dt * 1e3:8.1f} ms ({2 * a.shape[0] * a.shape[1] * w.shape[0] / dt / 1e12:5.2f} TFLOP/s)")
    print("  ".join(line), flush=True)
