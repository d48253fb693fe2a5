"""Where do the fused decode attention (rope + append + split attention) and its unfused test reference (llm_qkv_post + decode_attn) part
ways in the fp16 instantiation at head_dim 128?  Compares the rotated K / V appends, the per-split partials (m, l, O) and the merged outputs
of the two paths on the inputs of tests/test_kernels_f16_gpu.py::test_decode_attn_rope_f16_matches_unfused_pipeline.

    python tools/diag/decode_attn_paths.py        (GPU box)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from padt_amd import ops  # noqa: E402


def rnd(*shape, seed, dt):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda().to(dt)


def main():
    for dt in (torch.float16, torch.bfloat16):
        D, Hq, Hkv, sec = 128, 16, 2, (16, 24, 24)
        B, S_max = 3, 1344
        slots = [577, 63, 1290]
        qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=33, dt=dt)
        kc = rnd(B, Hkv, S_max, D, seed=34, dt=dt)
        vt = rnd(B, Hkv, S_max, D, seed=35, dt=dt).transpose(2, 3).contiguous()
        slot_t = torch.tensor(slots, dtype=torch.int32, device="cuda")
        inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
        nsplit = S_max // 64
        for seed in (1, 2):
            gpos = torch.randint(0, 4000, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(seed)).cuda()
            qa, qb = torch.zeros(B, Hq * D, device="cuda", dtype=dt), torch.zeros(B, Hq * D, device="cuda", dtype=dt)
            ka, va, kb, vb = kc.clone(), vt.clone(), kc.clone(), vt.clone()
            csx = torch.zeros(B, D // 2, 2, device="cuda")
            ops.rope_table(gpos, inv, csx, D, sec)
            wa = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
            wb = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
            wa.zero_(); wb.zero_()
            ops.decode_attn_rope(qkv, csx, slot_t, ka, va, qa, wa, Hq, Hkv, D, S_max, S_max)
            ops.llm_qkv_post(qkv, gpos, inv, qb, kb, vb, Hq, Hkv, D, S_max, sec, slot=slot_t)
            ob = torch.zeros_like(qa)
            ops.decode_attn(qb, kb, vb, slot_t + 1, ob, wb, Hq, Hkv, D, S_max, S_max)
            torch.cuda.synchronize()
            # the rotation the fused kernel performs, restated from its table (fp32 product, then one fused multiply-add: evaluated in fp64 and
            # rounded once), against the q the unfused path stored
            x = qkv.float().view(B, Hq + 2 * Hkv, D)[:, :Hq]
            x1, x2 = x[..., : D // 2], x[..., D // 2:]
            c, sn = csx[:, None, :, 0], csx[:, None, :, 1]
            lo = ((x1 * c).double() - x2.double() * sn.double()).float()
            hi = ((x2 * c).double() + x1.double() * sn.double()).float()
            q_tab = torch.cat([lo, hi], -1).to(dt).view(B, Hq * D)
            dq = (q_tab != qb).nonzero()
            print(f"    q rotated from the step table vs q stored by llm_qkv_post: {dq.shape[0]} of {qb.numel()} differ"
                  + (f"; first (b, head, d): {[(int(b), int(n) // D, int(n) % D) for b, n in dq[:4].tolist()]}" if dq.shape[0] else ""))
            sub = ((qb.float().abs() < 2.0 ** -14) & (qb.float() != 0)).nonzero()
            print(f"    subnormal rotated q elements (|q| < 2^-14): {[(int(b), int(n) // D, int(n) % D, float(qb[b, n])) for b, n in sub.tolist()]}; "
                  f"subnormal raw qkv elements (b, head [q 0..15 | k 16, 17 | v 18, 19], d, value): {[(int(b), int(n) // D, int(n) % D, float(qkv[b, n])) for b, n in ((qkv.float().abs() < 2.0 ** -14) & (qkv.float() != 0)).nonzero().tolist()]}; "
                  f"subnormal appended K: {sum(int(((kb[b, :, slots[b]].float().abs() < 2.0 ** -14) & (kb[b, :, slots[b]].float() != 0)).sum()) for b in range(B))}")
            ang_axis = torch.tensor([0 if d < sec[0] else (1 if d < sec[0] + sec[1] else 2) for d in range(D // 2)], device="cuda")
            ang = gpos.float()[ang_axis, :].T * inv[None, :]                       # [B][D/2]
            print(f"    step table vs torch.cos / torch.sin of the same fp32 angles: cos differs in {(csx[..., 0] != ang.cos()).sum().item()}, "
                  f"sin in {(csx[..., 1] != ang.sin()).sum().item()} of {ang.numel()}")
            n_o = B * Hkv * nsplit * 16 * D
            fa, fb = wa.view(torch.float32).flatten(), wb.view(torch.float32).flatten()
            oa_p, ob_p = fa[:n_o].view(B, Hkv, nsplit, 16, D), fb[:n_o].view(B, Hkv, nsplit, 16, D)
            ml_a, ml_b = fa[n_o:n_o + B * Hkv * nsplit * 32].view(B, Hkv, nsplit, 16, 2), fb[n_o:n_o + B * Hkv * nsplit * 32].view(B, Hkv, nsplit, 16, 2)
            print(f"[{dt}, seed {seed}] K equal {torch.equal(ka, kb)}, V equal {torch.equal(va, vb)}; outputs differing {(qa != ob).sum().item()} of {qa.numel()}")
            live = torch.zeros(B, Hkv, nsplit, 16, dtype=torch.bool, device="cuda")
            for b in range(B):
                live[b, :, : slots[b] // 64 + 1, : Hq // Hkv] = True
            dm = (ml_a[..., 0] != ml_b[..., 0]) & live
            dl = (ml_a[..., 1] != ml_b[..., 1]) & live
            do = (oa_p != ob_p).any(-1) & live
            print(f"    live (b, g, split, head) cells {live.sum().item()}: m differs in {dm.sum().item()}, l in {dl.sum().item()}, O in {do.sum().item()}")
            if dm.any() or dl.any() or do.any():
                idx = (dm | dl | do).nonzero()
                own = sum(1 for b, g, s, h in idx.tolist() if s == slots[b] // 64)
                print(f"    of the {idx.shape[0]} differing cells {own} are in the split that holds the appended token; first few (b, g, split, head): {idx[:8].tolist()}")
                # which path disagrees with an fp64 evaluation of max_j q . k_j from the STORED q and K?  and which one-ulp change of one
                # element of that head's q explains the other path's m over all splits?
                b, g, _, h = idx[0].tolist()
                hq = g * (Hq // Hkv) + h
                L = slots[b] + 1
                qv = qb[b].view(Hq, D)[hq].double()
                Kd = kb[b, g, :L].double()
                sl2 = D ** -0.5 * 1.4426950408889634
                def m_of(qvec):
                    sc = (Kd @ qvec) * sl2
                    pad = torch.full((nsplit * 64 - L,), -1e30, dtype=torch.float64, device="cuda")
                    return torch.cat([sc, pad]).view(nsplit, 64).max(-1).values
                m_ref = m_of(qv)
                ns = slots[b] // 64 + 1
                ea = (ml_a[b, g, :ns, h, 0].double() - m_ref[:ns]).abs().max().item()
                eb = (ml_b[b, g, :ns, h, 0].double() - m_ref[:ns]).abs().max().item()
                print(f"    head (b {b}, q head {hq}): max |m - fp64 m from the stored q, K| over its splits: fused {ea:.3g}, unfused {eb:.3g}")
                best = (1e9, None)
                q16 = qb[b].view(Hq, D)[hq]
                for d in range(D):
                    for sgn in (-1, 1):
                        qq = q16.clone()
                        bits = qq[d:d + 1].view(torch.int16)
                        bits += sgn
                        err = (ml_a[b, g, :ns, h, 0].double() - m_of(qq.double())[:ns]).abs().max().item()
                        if err < best[0]:
                            best = (err, (d, sgn, float(q16[d]), float(qq[d])))
                print(f"    best single one-ulp change of that head's stored q explaining the FUSED m: residual {best[0]:.3g} with (d, direction, stored, changed) = {best[1]}")
                x = qkv[b].view(Hq + 2 * Hkv, D)[hq].float()
                d0 = best[1][0] % (D // 2)
                cc, ssn = csx[b, d0, 0].item(), csx[b, d0, 1].item()
                print(f"    raw pair of that element: x1 {x[d0].item()!r} x2 {x[d0 + D // 2].item()!r} cos {cc!r} sin {ssn!r}; "
                      f"x1 cos - x2 sin = {x[d0].item() * cc - x[d0 + D // 2].item() * ssn!r}, x2 cos + x1 sin = {x[d0 + D // 2].item() * cc + x[d0].item() * ssn!r}")
                b, g, s, h = idx[0].tolist()
                print(f"    cell {b, g, s, h}: m {ml_a[b, g, s, h, 0].item():.9g} vs {ml_b[b, g, s, h, 0].item():.9g}; l {ml_a[b, g, s, h, 1].item():.9g} vs {ml_b[b, g, s, h, 1].item():.9g}; "
                      f"max |dO| {(oa_p[b, g, s, h] - ob_p[b, g, s, h]).abs().max().item():.3g}")


if __name__ == "__main__":
    main()
