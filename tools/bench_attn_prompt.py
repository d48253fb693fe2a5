import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
BF = torch.bfloat16
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
a = torch.randn(8192, 8192, device="cuda").to(BF); ops.gemm(a, a); ops.gemm(a, a)
for name, D, H, Hkv, lens in (("3B prompt", 128, 16, 2, [577] * 8), ("7B prompt", 128, 28, 4, [577] * 8), ("3B ovd prompt", 128, 16, 2, [890] * 8)):
    T = sum(lens); cu = [0]
    for l in lens: cu.append(cu[-1] + l)
    qkv = (torch.randn(T, (H + 2 * Hkv) * D, device="cuda") * 0.5).to(BF)
    q, k, v = qkv[:, : H * D], qkv[:, H * D: (H + Hkv) * D], qkv[:, (H + Hkv) * D:]
    out = torch.zeros(T, H * D, device="cuda", dtype=BF)
    cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
    us = t(lambda: ops.attn_varlen(q, k, v, out, cu_t, cu_t, max(lens), H, Hkv, D, causal=True))
    fl = sum(4.0 * l * l * D * H / 2 for l in lens)
    print(f"{name}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (causal half)", flush=True)
