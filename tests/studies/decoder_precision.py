"""CPU study (not a test; run by hand):  which bf16 roundings of the HIP PaDT-decoder path cost how much box / mask
accuracy at the REAL decoder shape (98 M parameters, 3 objects over 2 images — the inputs of
test_real_shape_gpu.py::test_padt_decoder_real_shape_against_reference_output).

The oracle's decoder (oracle/padt_oracle.py:510-622 ≙ padt_decoder.py:20-276) is re-run with a rounding hook R(cat, x) at
every place the HIP path stores a tensor; categories can be switched to bf16 / fp32 / "split" (bf16 hi + bf16 lo =
16 mantissa bits, what a 2-MFMA split-operand GEMM sees) independently.  Output: |Δ| of boxes / scores / mask logits
against the unrounded fp32 run on the same bf16-representable weights and inputs.

    python tests/study_decoder_precision.py
"""
import itertools
import os
import sys
import zlib

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import padt_oracle as O  # noqa: E402

bf = torch.bfloat16


def rb(x):
    return x.to(bf).float()


def rsplit(x):                     # hi + lo, both bf16
    hi = x.to(bf).float()
    return hi + (x - hi).to(bf).float()


MODES = {"f32": lambda x: x, "bf16": rb, "split": rsplit}


class Emu:
    def __init__(self, cats):
        self.cats = cats           # category → mode name

    def R(self, cat, x):
        return MODES[self.cats.get(cat, "f32")](x)

    # ---- decoder with hooks
    def attention(self, w, pfx, heads, query, key, cu_q, cu_k, q_pos, k_pos, rotary, side):
        R = self.R
        q_in = query if rotary[0] else R("norm_" + side[0], query + q_pos)
        k_in = key if rotary[1] else R("norm_" + side[1], key + k_pos)
        q = R("proj_" + side[0], O.linear(q_in, w[pfx + "q_proj.weight"], w[pfx + "q_proj.bias"]))
        k = R("proj_" + side[1], O.linear(k_in, w[pfx + "k_proj.weight"], w[pfx + "k_proj.bias"]))
        v = R("proj_" + side[1], O.linear(key, w[pfx + "v_proj.weight"], w[pfx + "v_proj.bias"]))
        q = q.reshape(query.shape[0], heads, -1)
        k = k.reshape(key.shape[0], heads, -1)
        v = v.reshape(key.shape[0], heads, -1)
        if rotary[0]:
            q = R("proj_" + side[0], O._apply_rotary_half(q, q_pos[0].chunk(2, -1)[0], q_pos[1].chunk(2, -1)[0]))
        if rotary[1]:
            k = R("proj_" + side[1], O._apply_rotary_half(k, k_pos[0].chunk(2, -1)[0], k_pos[1].chunk(2, -1)[0]))
        outs = []
        sc = q.shape[-1] ** -0.5
        for i in range(len(cu_q) - 1):
            qs, ks, vs = q[cu_q[i]:cu_q[i + 1]], k[cu_k[i]:cu_k[i + 1]], v[cu_k[i]:cu_k[i + 1]]
            s = torch.einsum("qhd,khd->hqk", qs, ks) * sc
            m = s.max(-1, keepdim=True).values
            p = torch.exp(s - m)
            l = p.sum(-1, keepdim=True)
            p = R("p_" + side[0], p)                                  # flash-attn-2 style: P rounded before P·V, l from fp32 P
            o = torch.einsum("hqk,khd->qhd", p, vs) / l.permute(1, 0, 2)
            outs.append(o)
        a = R("attn_" + side[0], torch.cat(outs).reshape(query.shape[0], -1))
        return O.linear(a, w[pfx + "proj.weight"], w[pfx + "proj.bias"])

    def block(self, w, pfx, heads, query, memory, cu_q, cu_m, query_pos, memory_pos):
        R = self.R
        qn = R("norm_q", O.rms_norm(query, w[pfx + "norm1.weight"]))
        query = R("resid_q", query + self.attention(w, pfx + "self_attn.", heads, qn, qn, cu_q, cu_q, query_pos, query_pos, (False, False), "qq"))
        qn = R("norm_q", O.rms_norm(query, w[pfx + "norm2.weight"]))
        mn = R("norm_m", O.rms_norm(memory, w[pfx + "norm3.weight"]))
        query = R("resid_q", query + self.attention(w, pfx + "cross_attn_query_to_image.", heads, qn, mn, cu_q, cu_m, query_pos, memory_pos, (False, True), "qm"))
        n4 = R("norm_q", O.rms_norm(query, w[pfx + "norm4.weight"]))
        h = R("mlp_q", F.gelu(O.linear(n4, w[pfx + "mlp.0.weight"], w[pfx + "mlp.0.bias"])))
        query = R("resid_q", query + O.linear(h, w[pfx + "mlp.2.weight"], w[pfx + "mlp.2.bias"]))
        qn = R("norm_q", O.rms_norm(query, w[pfx + "norm5.weight"]))
        mn = R("norm_m", O.rms_norm(memory, w[pfx + "norm6.weight"]))
        memory = R("resid_m", memory + self.attention(w, pfx + "cross_attn_image_to_query.", heads, mn, qn, cu_m, cu_q, memory_pos, query_pos, (True, False), "mq"))
        return query, memory

    def mlp3(self, w, pfx, x, cat):
        R = self.R
        x = R(cat, F.gelu(O.linear(x, w[pfx + "0.weight"], w[pfx + "0.bias"])))
        x = R(cat, F.gelu(O.linear(x, w[pfx + "2.weight"], w[pfx + "2.bias"])))
        return O.linear(x, w[pfx + "4.weight"], w[pfx + "4.bias"])

    def decoder(self, w, cfg, object_vp_feat, cu_low, cu_high, visual_pe, cu_patch, obj_grids):
        R = self.R
        p = "vl_decoder."
        heads, mu = cfg.dec_heads, cfg.merge_unit
        n_vp = [f.shape[0] for f in object_vp_feat]
        n_obj = len(n_vp)

        def in_proj(x, cat):
            x = R(cat, O.rms_norm(x, w[p + "input_projection.0.weight"]))
            x = R(cat, F.gelu(O.linear(x, w[p + "input_projection.1.weight"], w[p + "input_projection.1.bias"])))
            return O.linear(x, w[p + "input_projection.3.weight"], w[p + "input_projection.3.bias"])
        feats = R("resid_q", R("inproj_q", in_proj(torch.cat(object_vp_feat), "inproj_q")) + w[p + "vp_embedding.weight"])
        cu_query, acc = [], 0
        for n in n_vp:
            cu_query += [w[p + "bbox_score_mask_tokens.weight"], feats[acc:acc + n]]
            acc += n
        cu_query = torch.cat(cu_query)
        cu_q = [0]
        for n in n_vp:
            cu_q.append(cu_q[-1] + 3 + n)
        cu_p = cu_patch.tolist()
        cu_l = [c // mu for c in cu_p]
        low = R("resid_m", in_proj(cu_low, "inproj_m"))
        D = visual_pe[0].shape[-1]
        low_pe = (visual_pe[0].reshape(-1, mu, D)[:, 0, :], visual_pe[1].reshape(-1, mu, D)[:, 0, :])
        out, low = self.block(w, p + "low_res_transformer.", heads, cu_query, low, cu_q, cu_l, cu_query, low_pe)
        high = R("resid_m", O.rms_norm(low.unsqueeze(1).repeat_interleave(mu, dim=1).flatten(0, 1) + cu_high, w[p + "high_res_norm.weight"]))
        out, high = self.block(w, p + "high_res_transformer1.", heads, out, high, cu_q, cu_p, cu_query, visual_pe)
        out, high = self.block(w, p + "high_res_transformer2.", heads, out, high, cu_q, cu_p, cu_query, visual_pe)
        tok = torch.stack([out[cu_q[i]: cu_q[i] + 3] for i in range(n_obj)])
        tok = R("head_in", tok)
        bbox = torch.sigmoid(self.mlp3(w, p + "bbox_prediction.", tok[:, 0], "head"))
        score = O.linear(tok[:, 1], w[p + "score_prediction.weight"], w[p + "score_prediction.bias"])
        mask_tok = R("mask_tok", self.mlp3(w, p + "mask_output_mlp.", tok[:, 2], "head"))
        N, Dd = high.shape
        hi_in = R("mask_in", high)
        up1 = R("mask_up", O.linear(hi_in, w[p + "mask_output_upscaling1.0.weight"], w[p + "mask_output_upscaling1.0.bias"]))
        up1 = R("mask_up", F.gelu(O.rms_norm(up1, w[p + "mask_output_upscaling1.1.weight"])))
        e = up1.reshape(N, 2, 2, Dd // 4).permute(1, 2, 0, 3)
        e = R("mask_e2", F.gelu(O.linear(e, w[p + "mask_output_upscaling2.0.weight"], w[p + "mask_output_upscaling2.0.bias"])))
        e = e.reshape(2, 2, N, 2, 2, Dd // 16).permute(0, 3, 1, 4, 2, 5).flatten(0, 1).flatten(1, 2)
        per_patch = e.permute(2, 0, 1, 3).contiguous()
        pn = cu_patch[1:] - cu_patch[:-1]
        obj_of = torch.repeat_interleave(torch.arange(n_obj), pn.long())
        logit = (per_patch * mask_tok.index_select(0, obj_of)[:, None, None, :]).sum(-1)
        return bbox, score, logit


def seeded(shape, name, scale, jitter_one=False):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = scale * torch.randn(shape, generator=g)
    return 1 + t if jitter_one else t


def setup():
    ocfg = O.OracleConfig()
    w = {}
    for k, shp in O.weight_shapes(ocfg).items():
        if k.startswith("vl_decoder."):
            w[k] = rb(seeded(shp, k, 0.1, True) if O._is_norm_weight(k) else seeded(shp, k, 0.02 if k.endswith("bias") else 0.03))
    g = torch.Generator().manual_seed(123)
    _ = torch.randn(2116, 1280, generator=g)
    _ = torch.randint(0, 2116, (48,), generator=g)
    grids = torch.tensor([[1, 46, 30], [1, 8, 8]])
    Ps = [46 * 30, 64]
    low = rb(torch.randn(sum(Ps) // 4, 2048, generator=g))
    high = rb(torch.randn(sum(Ps), 1280, generator=g))
    wi, _ = O.window_index(grids, 2, 112, 14)
    c, s = O.vit_rotary(ocfg, grids, wi)
    feats = [[rb(torch.randn(5, 2048, generator=g)), rb(torch.randn(2, 2048, generator=g))], [rb(torch.randn(4, 2048, generator=g))]]
    # vl_decode's replication (padt.py:362-376)
    flat = sum(feats, [])
    off, lows, highs, pc, ps, cu, gl = 0, [], [], [], [], [], []
    for fs, gr in zip(feats, grids):
        n = int(gr[0] * gr[1] * gr[2])
        k = len(fs)
        lows.append(low[off // 4:(off + n) // 4].repeat(k, 1))
        highs.append(high[off:off + n].repeat(k, 1))
        pc.append(c[off:off + n].repeat(k, 1))
        ps.append(s[off:off + n].repeat(k, 1))
        cu += [n] * k
        gl += [gr] * k
        off += n
    cu_patch = F.pad(torch.tensor(cu, dtype=torch.float32).cumsum(0), (1, 0)).to(torch.int32)
    return ocfg, w, flat, torch.cat(lows), torch.cat(highs), (torch.cat(pc), torch.cat(ps)), cu_patch, torch.stack(gl)


CATS = ["resid_q", "resid_m", "norm_q", "norm_m", "proj_q", "proj_m", "p_q", "p_m", "attn_q", "attn_m", "mlp_q", "inproj_q",
        "inproj_m", "head_in", "head", "mask_tok", "mask_in", "mask_up", "mask_e2"]


def main():
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    args = setup()
    with torch.no_grad():
        ref = Emu({}).decoder(*[args[1], args[0]] + list(args[2:]))

        def run(cats, label):
            b, s, m = Emu(cats).decoder(*[args[1], args[0]] + list(args[2:]))
            db = (b - ref[0]).abs().max().item()
            ds = (s - ref[1]).abs().max().item()
            dm = (m - ref[2]).abs().max().item()
            rms = ((m - ref[2]).pow(2).mean().sqrt() / ref[2].pow(2).mean().sqrt()).item()
            print(f"{label:58s} box {db:.2e}  score {ds:.2e} (|s|max {ref[1].abs().max():.2f})  mask max {dm:.2e} rel-rms {rms:.2e} (|m|max {ref[2].abs().max():.2f})")
        run({c: "bf16" for c in CATS}, "all bf16 (= current HIP path)")
        for c in CATS:
            run({c: "bf16"}, f"only {c} bf16")
        base = {c: "bf16" for c in CATS}
        for off in (["resid_q", "resid_m"], ["resid_q", "resid_m", "norm_q", "proj_q", "attn_q", "mlp_q", "inproj_q", "head_in", "head", "p_q"],):
            d = dict(base)
            for c in off:
                d[c] = "f32"
            run(d, "bf16 except f32: " + ",".join(off)[:40])
        if len(sys.argv) > 1:
            for spec in sys.argv[1:]:                 # e.g. resid_q=f32,norm_q=split
                d = dict(base)
                for kv in spec.split(","):
                    k, v = kv.split("=")
                    for c in CATS:
                        if c == k or (k.endswith("*") and c.startswith(k[:-1])):
                            d[c] = v
                run(d, spec[:58])


if __name__ == "__main__":
    main()
