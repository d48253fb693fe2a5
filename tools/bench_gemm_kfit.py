import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
BF = torch.bfloat16
def t(M, N, K, epi, reps=20):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    out = torch.zeros(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
    res = out if epi == 2 else None
    for _ in range(3): ops.gemm(a, w, out=out, epilogue=epi, residual=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, w, out=out, epilogue=epi, residual=res)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# warm clocks
t(8192, 8192, 8192, 0, 10)
ops.gemm_knobs(mf=4, peel=0, colsplit=0)
for (M, N, epi, name) in [(16896, 6912, 3, "vit gate/up 66x27=1782 tiles (6.96 rounds)"), (16896, 3840, 0, "vit qkv 66x15=990 (3.87)"), (4096, 4096, 0, "exactly 256 tiles (1 round)"), (8192, 8192, 0, "1024 tiles (4 rounds)")]:
    line = name + ": "
    ts = []
    for K in (640, 1280, 2560, 5120):
        us = t(M, N, K, epi); ts.append(us)
        line += f" K={K}: {us:7.1f}us"
    tiles = (M // 256) * (N // 256); rounds = -(-tiles // 256)
    slope = (ts[3] - ts[1]) / (60 * rounds)      # us per K-tile per round
    icpt = ts[1] / rounds - slope * 20
    print(line + f" | per K-tile {slope:.3f} us, fixed per round {icpt:.2f} us", flush=True)
