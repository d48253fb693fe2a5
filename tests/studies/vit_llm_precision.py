"""Hand-run study (not collected by pytest): which bf16 storage sites set the end-to-end distance of the ViT / LLM stack to the fp32
oracle?  Replays the ORACLE's ViT (real PaDT_Pro_3B shape: 32 blocks, 2116 x 1280, seeded weights as in
test_full_depth_3b_teacher_forced_against_oracle) on the CPU with a rounding hook at each class of stored tensor:

  resid    the residual stream is rounded to bf16 after every residual add (what rounds 1-2 stored)
  operand  every GEMM A operand (normed x, attention output, SwiGLU hidden) and q / k / v / P are rounded to bf16 — the floor ANY
           bf16-MFMA implementation has, whatever it keeps in fp32 between kernels
  both     round-2 arithmetic

    python tests/studies/vit_llm_precision.py [depth]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import parity_util as U  # noqa: E402

O = U.O
bf = lambda t: t.to(torch.bfloat16).float()


def vit(w, cfg, pix, grid, resid, operand, depth):
    r_res = bf if resid else (lambda t: t)
    r_op = bf if operand else (lambda t: t)
    H = cfg.vit_heads
    pw = w["visual.patch_embed.proj.weight"].reshape(cfg.vit_hidden, -1)
    x = r_res(F.linear(r_op(pix), pw))
    win_idx, cu_win = O.window_index(grid, cfg.spatial_merge_size, cfg.window_size, cfg.patch_size)
    P = x.shape[0]
    x = x.reshape(P // 4, 4, -1)[win_idx].reshape(P, -1)
    cos, sin = O.vit_rotary(cfg, grid, win_idx)
    cu_full = [0, P]
    outs = []
    for i in range(depth):
        p = f"visual.blocks.{i}."
        cu = cu_full if i in cfg.fullatt_block_indexes else cu_win
        T = x.shape[0]
        # the HIP path rounds x itself (rstd and the norm weight are applied to the accumulator / folded into W)
        xf = r_op(x)
        rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
        n = xf * rstd * w[p + "norm1.weight"]
        qkv = F.linear(n, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]).reshape(T, 3, H, -1)
        q, k, v = qkv.permute(1, 0, 2, 3).unbind(0)
        c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
        q = r_op(q * c + O.rotate_half(q) * s)
        k = r_op(k * c + O.rotate_half(k) * s)
        v = r_op(v)
        a = torch.empty_like(q)
        for sgi in range(len(cu) - 1):
            a0, a1 = int(cu[sgi]), int(cu[sgi + 1])
            qs, ks, vs = (t[a0:a1].transpose(0, 1) for t in (q, k, v))
            sc = torch.matmul(qs, ks.transpose(1, 2)) * (q.shape[-1] ** -0.5)
            mx = sc.max(-1, keepdim=True).values
            e = torch.exp(sc - mx)
            l = e.sum(-1, keepdim=True)                       # flash attention: P is rounded for the MFMA, l is summed in fp32
            a[a0:a1] = (torch.matmul(r_op(e), vs) / l).transpose(0, 1)
        a = r_op(a.reshape(T, -1))
        x = r_res(x + F.linear(a, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"]))
        xf = r_op(x)
        rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6)
        n = xf * rstd * w[p + "norm2.weight"]
        g = F.linear(n, w[p + "mlp.gate_proj.weight"], w[p + "mlp.gate_proj.bias"])
        u = F.linear(n, w[p + "mlp.up_proj.weight"], w[p + "mlp.up_proj.bias"])
        x = r_res(x + F.linear(r_op(F.silu(g) * u), w[p + "mlp.down_proj.weight"], w[p + "mlp.down_proj.bias"]))
        outs.append(x)
    return outs


def main():
    import padt_amd
    from padt_amd.weights import synthetic_state_dict
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    cfg = padt_amd.padt_pro_3b()
    sd = synthetic_state_dict(cfg, seed=3, std=0.02, bias_std=0.02, norm_jitter=0.1, device="cpu", dtype=torch.bfloat16)
    w = {k: v.float() for k, v in sd.items() if k.startswith("visual.")}
    oc = U.oracle_config(cfg)
    grid, pix, ids, am = U.synthetic_batch(cfg, [[1, 46, 46]], n_pre=15, n_post=33, seed=77)
    torch.set_num_threads(os.cpu_count() or 8)
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = vit(w, oc, pix.float(), grid, False, False, depth)
        print(f"fp32 reference: {time.perf_counter() - t0:.1f} s", flush=True)
        for name, (rs, op) in {"resid": (True, False), "operand": (False, True), "both": (True, True)}.items():
            got = vit(w, oc, pix.float(), grid, rs, op, depth)
            line = []
            for i in (0, 3, 7, 15, 23, 31):
                if i < depth:
                    d = (got[i] - ref[i]).pow(2).mean().sqrt() / ref[i].pow(2).mean().sqrt()
                    line.append(f"blk{i}: {d:.2e}")
            print(f"{name:8s} rel rms  " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
