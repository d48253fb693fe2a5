import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
BF = torch.bfloat16
for (M, N, K, epi) in [(8192, 8192, 8192, 0), (4616, 22016, 2048, 3), (16928, 3840, 1280, 0)]:
    a = torch.randn(M, K, device="cuda").to(BF)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    out = torch.zeros(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
    for _ in range(3):
        ops.gemm(a, w, out=out, epilogue=epi)
    torch.cuda.synchronize()
