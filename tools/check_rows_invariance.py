"""Are the decode projections' per-row results independent of how many rows share the launch (8 vs 64 rows: MT = 1 vs 4 kernels)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from padt_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
for name, N, K, epi in (("qkv", 2560, 2048, 0), ("o", 2048, 2048, 2), ("gu", 22016, 2048, 3), ("down", 2048, 11008, 2)):
    w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    wp = ops.pack_weight(w)
    x64 = torch.randn(64, K, device="cuda").to(BF)
    outs = {}
    for B in (8, 16, 32, 64):
        xp = torch.zeros((B + 15) // 16 * 16, K, device="cuda", dtype=BF)
        ops.pack_rows(x64[:B].contiguous(), xp, B, to_packed=True)
        n_out = N // 2 if epi == 3 else N
        o = torch.zeros((B + 15) // 16 * 16, n_out, device="cuda", dtype=BF)
        r = torch.zeros_like(o)
        split = 2 if name == "down" else 1
        ws = ops.new_splitk_workspace(N, 2, "cuda")
        if epi == 2:
            ops.gemm_packed(xp, wp, N, out=o, epilogue=2, residual=r, split_k=split, workspace=ws, a_packed=True, c_packed=True, rows=B)
        else:
            ops.gemm_packed(xp, wp, N, out=o, epilogue=epi, norm_eps=1e-6, a_packed=True, c_packed=True, rows=B)
        un = torch.zeros(B, n_out, device="cuda", dtype=BF)
        ops.pack_rows(o, un, B, to_packed=False)
        outs[B] = un[:8].clone()
    print(name, {B: bool(torch.equal(outs[8], outs[B])) for B in outs}, "max |d| 8 vs 64:", (outs[8].float() - outs[64].float()).abs().max().item())
