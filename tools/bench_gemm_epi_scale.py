import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
BF = torch.bfloat16
def t(M, N, K, epi, reps=30):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    out = torch.zeros(M, N // 2 if epi == 3 else N, device="cuda", dtype=BF)
    for _ in range(3): ops.gemm(a, w, out=out, epilogue=epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(a, w, out=out, epilogue=epi)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t(8192, 8192, 8192, 0, 10)
ops.gemm_knobs(mode256=2, mf=4, peel=0, colsplit=0)
for tiles, (M, N) in [(8, (512, 1024)), (32, (1024, 2048)), (64, (2048, 2048)), (128, (2048, 4096)), (256, (4096, 4096))]:
    print(f"{tiles:4d} tiles: K=1280 {t(M, N, 1280, 0):6.1f} us   K=2560 {t(M, N, 2560, 0):6.1f} us", flush=True)
