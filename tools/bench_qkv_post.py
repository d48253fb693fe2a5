import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from padt_amd import ops
BF = torch.bfloat16
Hq, Hkv, D, S_max, B, L = 16, 2, 128, 640, 8, 577
T = B * L
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device="cuda").to(BF)
pos = torch.arange(L, device="cuda", dtype=torch.int32).repeat(B)[None].repeat(3, 1).contiguous()
sample = torch.arange(B, device="cuda", dtype=torch.int32).repeat_interleave(L)
slot = torch.arange(L, device="cuda", dtype=torch.int32).repeat(B)
inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
q = torch.zeros(T, Hq * D, device="cuda", dtype=BF); kp = torch.zeros(T, Hkv * D, device="cuda", dtype=BF)
kc = torch.zeros(B, Hkv, S_max, D, device="cuda", dtype=BF); vt = torch.zeros(B, Hkv, D, S_max, device="cuda", dtype=BF)
def f(): ops.llm_qkv_post(qkv, pos, inv, q, kc, vt, Hq, Hkv, D, S_max, (16, 24, 24), sample=sample, slot=slot, k_pack=kp)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f()
e1.record(); torch.cuda.synchronize()
print("llm_qkv_post (8 x 577 tokens): %.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
v = qkv[:, (Hq + Hkv) * D:].view(B, L, Hkv, D).permute(0, 2, 3, 1)
print("V^T cache correct:", bool(torch.equal(vt[:, :, :, :L], v)), "untouched tail:", bool((vt[:, :, :, L:] == 0).all()))
