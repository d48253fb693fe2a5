"""Shared helpers for the GPU parity tests / smoke / bench cpu_baseline: build matching product + oracle models from the
same seeded bf16-representable weights, synthetic prompts, a fake tokenizer for the parser."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import padt_oracle as O  # noqa: E402  (test infrastructure — never imported by padt_amd/)
from padt_amd.config import PaDTConfig  # noqa: E402
from padt_amd.weights import synthetic_state_dict  # noqa: E402


def oracle_config(cfg: PaDTConfig) -> O.OracleConfig:
    v = cfg.vision_config
    d = cfg.vl_decoder
    return O.OracleConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_layers=cfg.num_hidden_layers,
        num_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
        intermediate_size=cfg.intermediate_size, rms_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        mrope_section=tuple(cfg.mrope_section), tie_word_embeddings=cfg.tie_word_embeddings, vit_hidden=v.hidden_size,
        vit_depth=v.depth, vit_heads=v.num_heads, vit_intermediate=v.intermediate_size, patch_size=v.patch_size,
        temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels, spatial_merge_size=v.spatial_merge_size,
        window_size=v.window_size, fullatt_block_indexes=tuple(v.fullatt_block_indexes),
        use_visual_prototype_projection=cfg.use_visual_prototype_projection, lora_r=cfg.lora_r,
        dec_hidden=d["hidden_size"], dec_heads=d["num_heads"], dec_intermediate=d["intermediate_size"],
        use_mask_loss=d.get("use_mask_loss", True), image_token_id=cfg.image_token_id,
        vision_start_token_id=cfg.vision_start_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)


def bf16_weights(cfg: PaDTConfig, seed=0, std=0.02, bias_std=0.02, norm_jitter=0.1):
    """fp32 tensors whose values are exactly bf16-representable: the HIP path and the fp32 oracle see the same numbers."""
    sd = synthetic_state_dict(cfg, seed=seed, std=std, bias_std=bias_std, norm_jitter=norm_jitter, device="cpu")
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


from padt_amd.synthetic import FakeProcessor, FakeTokenizer, rec_schedule, synthetic_batch  # noqa: E402,F401


def effective_llm_weights(model, w):
    """Oracle weights for a model built with llm_weights="fp8": the reference-layout fp32 dict `w` with the LLM projections replaced by
    the DEQUANTISED matrices the HIP path multiplies with (prepare_weights keeps them as bf16 images: scale * q is exact), un-fused
    (q | k | v), un-interleaved and un-padded (gate / up), with the RMSNorm weights they were folded with set to one — the same
    function of the input, parameterised the way the reference's modules are."""
    cfg, W = model.config, model.W
    hd, Hq, Hkv, I = cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    out = dict(w)
    for i in range(cfg.num_hidden_layers):
        s, d = f"model.layers.{i}.", f"llm.{i}."
        qkv = W[d + "qkv.w"].float().cpu()
        out[s + "self_attn.q_proj.weight"] = qkv[: Hq * hd]
        out[s + "self_attn.k_proj.weight"] = qkv[Hq * hd: (Hq + Hkv) * hd]
        out[s + "self_attn.v_proj.weight"] = qkv[(Hq + Hkv) * hd:]
        out[s + "self_attn.o_proj.weight"] = W[d + "o.w"].float().cpu()
        gu = W[d + "gu.w"].float().cpu()
        gu = gu.view(gu.shape[0] // 32, 2, 16, gu.shape[1])
        out[s + "mlp.gate_proj.weight"] = gu[:, 0].reshape(-1, gu.shape[-1])[:I]
        out[s + "mlp.up_proj.weight"] = gu[:, 1].reshape(-1, gu.shape[-1])[:I]
        out[s + "mlp.down_proj.weight"] = W[d + "down.w"].float().cpu()[:, :I]
        out[s + "input_layernorm.weight"] = torch.ones_like(w[s + "input_layernorm.weight"])
        out[s + "post_attention_layernorm.weight"] = torch.ones_like(w[s + "post_attention_layernorm.weight"])
    return out


# ------------------------------------------------------------------------------------------------ operand floors
OPERAND_CLASSES = ("vit.rows", "vit.qkv", "vit.p", "vit.ao", "vit.hid", "vit.misc",
                   "llm.rows", "llm.qkv", "llm.p", "llm.ao", "llm.hid", "head")


class operand_floor:
    """Context manager: inside it the ORACLE's ViT / LLM arithmetic rounds matrix-multiply operands on the ACTIVATION side to a 16-bit
    type and keeps EVERYTHING else in fp32 (residual streams, accumulators, softmax statistics, norms, weights as given).  With every
    class on and dtype = bf16 that is the smallest distance to the fp32 reference any implementation on bf16 MFMA operands can have,
    whatever it stores between kernels; the full-depth parity tests measure it on their own inputs and bound the HIP path by a multiple
    of it.  `classes` selects WHICH operands are rounded (the attribution study, tests/studies/operand_attribution.py):

      {vit,llm}.rows  the normalised rows entering qkv / gate-up     {vit,llm}.qkv  q, k, v after the rotation (and the KV cache)
      {vit,llm}.p     the un-normalised probabilities                {vit,llm}.ao   the attention output entering proj / o_proj
      {vit,llm}.hid   the SwiGLU hidden entering down_proj           vit.misc       pixel rows, merger and prototype-projection inputs
      head            last-layer hidden rows and prototype rows entering the logit head

    dtype: torch.bfloat16 (8 mantissa bits) or torch.float16 (11).  The PaDT decoder is left alone (call vl_decode outside the context;
    the HIP decoder runs split-precision operands)."""

    def __init__(self, dtype=torch.bfloat16, classes=OPERAND_CLASSES, gemm=None):
        """gemm (round 6, tests/studies/split_fp8_floor.py): optional callable (x, w, b, cls) → F.linear-like result that REPLACES the plain
        "round the activation operand, multiply" of the projection sites (classes *.rows / *.ao / *.hid / vit.misc) — a model of a GEMM with a
        different operand scheme (e.g. an fp16 hi term plus an fp8 lo term against an fp8 weight image); the attention-internal classes
        (*.qkv, *.p) and the head keep the rounding of `dtype`."""
        self.dtype = dtype
        self.gemm = gemm
        self.classes = frozenset(classes)
        unknown = self.classes - set(OPERAND_CLASSES)
        if unknown:
            raise ValueError("unknown operand classes: %s" % sorted(unknown))

    def __enter__(self):
        import torch.nn.functional as F
        dt, on = self.dtype, self.classes
        self._saved = {k: getattr(O, k) for k in ("vit_block", "llm_layer", "linear", "vrt_logits")}

        def rd(t, cls):
            return t.to(dt).to(torch.float32) if cls in on else t
        gemm = self.gemm

        def mm(x, w, b, cls):                                        # one projection site
            if gemm is not None and cls in on:
                return gemm(x, w, b, cls)
            return F.linear(rd(x, cls), w, b)

        def linear(x, w, b=None):                                    # the sites outside the blocks: patch embed, merger, prototypes
            return mm(x, w, b, "vit.misc")

        def vit_block(w, pfx, cfg, x, cu, cos, sin):
            H, T = cfg.vit_heads, x.shape[0]
            n = O.rms_norm(x, w[pfx + "norm1.weight"], 1e-6)
            qkv = mm(n, w[pfx + "attn.qkv.weight"], w[pfx + "attn.qkv.bias"], "vit.rows").reshape(T, 3, H, -1)
            q, k, v = qkv.permute(1, 0, 2, 3).unbind(0)
            c, s = cos.unsqueeze(-2).float(), sin.unsqueeze(-2).float()
            q, k, v = rd(q * c + O.rotate_half(q) * s, "vit.qkv"), rd(k * c + O.rotate_half(k) * s, "vit.qkv"), rd(v, "vit.qkv")
            a = torch.empty_like(q)
            for i in range(len(cu) - 1):
                a0, a1 = int(cu[i]), int(cu[i + 1])
                qs, ks, vs = (t[a0:a1].transpose(0, 1) for t in (q, k, v))
                sc = torch.matmul(qs, ks.transpose(1, 2)) * (q.shape[-1] ** -0.5)
                e = torch.exp(sc - sc.max(-1, keepdim=True).values)
                a[a0:a1] = (torch.matmul(rd(e, "vit.p"), vs) / e.sum(-1, keepdim=True)).transpose(0, 1)
            x = x + mm(a.reshape(T, -1), w[pfx + "attn.proj.weight"], w[pfx + "attn.proj.bias"], "vit.ao")
            n = O.rms_norm(x, w[pfx + "norm2.weight"], 1e-6)
            g = mm(n, w[pfx + "mlp.gate_proj.weight"], w[pfx + "mlp.gate_proj.bias"], "vit.rows")
            u = mm(n, w[pfx + "mlp.up_proj.weight"], w[pfx + "mlp.up_proj.bias"], "vit.rows")
            return x + mm(F.silu(g) * u, w[pfx + "mlp.down_proj.weight"], w[pfx + "mlp.down_proj.bias"], "vit.hid")

        def llm_layer(w, pfx, cfg, h, cos, sin, attn_bias, cache, li):
            B, Lq, _ = h.shape
            n = O.rms_norm(h, w[pfx + "input_layernorm.weight"], cfg.rms_eps)
            q = mm(n, w[pfx + "self_attn.q_proj.weight"], w[pfx + "self_attn.q_proj.bias"], "llm.rows").view(B, Lq, cfg.num_heads, cfg.head_dim)
            k = mm(n, w[pfx + "self_attn.k_proj.weight"], w[pfx + "self_attn.k_proj.bias"], "llm.rows").view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
            v = mm(n, w[pfx + "self_attn.v_proj.weight"], w[pfx + "self_attn.v_proj.bias"], "llm.rows").view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
            c, s = cos.unsqueeze(2), sin.unsqueeze(2)
            q, k, v = rd(q * c + O.rotate_half(q) * s, "llm.qkv"), rd(k * c + O.rotate_half(k) * s, "llm.qkv"), rd(v, "llm.qkv")
            if cache is not None:
                k, v = cache.update(li, k, v)
            rep = cfg.num_heads // cfg.num_kv_heads
            qh = q.transpose(1, 2)
            kh = k.transpose(1, 2).repeat_interleave(rep, 1)
            vh = v.transpose(1, 2).repeat_interleave(rep, 1)
            sc = torch.matmul(qh, kh.transpose(2, 3)) * (cfg.head_dim ** -0.5) + attn_bias
            e = torch.exp(sc - sc.max(-1, keepdim=True).values)
            a = (torch.matmul(rd(e, "llm.p"), vh) / e.sum(-1, keepdim=True)).transpose(1, 2).reshape(B, Lq, -1)
            h = h + mm(a, w[pfx + "self_attn.o_proj.weight"], None, "llm.ao")
            n = O.rms_norm(h, w[pfx + "post_attention_layernorm.weight"], cfg.rms_eps)
            g = mm(n, w[pfx + "mlp.gate_proj.weight"], None, "llm.rows")
            u = mm(n, w[pfx + "mlp.up_proj.weight"], None, "llm.rows")
            return h + mm(F.silu(g) * u, w[pfx + "mlp.down_proj.weight"], None, "llm.hid")

        def vrt_logits(w, cfg, hidden, proto, lmask):
            return self._saved["vrt_logits"](w, cfg, rd(hidden, "head"), rd(proto, "head"), lmask)

        O.linear, O.vit_block, O.llm_layer, O.vrt_logits = linear, vit_block, llm_layer, vrt_logits
        return self

    def __exit__(self, *exc):
        for k, v in self._saved.items():
            setattr(O, k, v)
        return False


def folded_weight_images(w, cfg: PaDTConfig, dtype):
    """The reference-layout dict with q / k / v / gate / up (ViT and LLM) replaced by round_dtype(w_norm * W) / w_norm: the same function of
    the input, carrying the ONE weight-side rounding the HIP path adds — padt_amd/weights.py folds each RMSNorm weight into the rows of the
    projection that consumes the normalised activations and rounds the product to the operand type."""
    out = dict(w)

    def fold(names, nw):
        for nm in names:
            out[nm] = (w[nm] * nw).to(dtype).float() / nw
    for i in range(cfg.num_hidden_layers):
        s = f"model.layers.{i}."
        fold([s + "self_attn.q_proj.weight", s + "self_attn.k_proj.weight", s + "self_attn.v_proj.weight"], w[s + "input_layernorm.weight"])
        fold([s + "mlp.gate_proj.weight", s + "mlp.up_proj.weight"], w[s + "post_attention_layernorm.weight"])
    for i in range(cfg.vision_config.depth):
        s = f"visual.blocks.{i}."
        fold([s + "attn.qkv.weight"], w[s + "norm1.weight"])
        fold([s + "mlp.gate_proj.weight", s + "mlp.up_proj.weight"], w[s + "norm2.weight"])
    return out


def bf16_operand_floor():
    """Every operand class rounded to bf16 (what rounds 2-3 measured and asserted against)."""
    return operand_floor(torch.bfloat16)


def fp16_operand_floor():
    """Every operand class rounded to fp16: the floor of an implementation on `v_mfma_f32_16x16x32_f16` operands."""
    return operand_floor(torch.float16)


# ------------------------------------------------------------------------------------------------ fp8 x fp8 prompt pass
def fake_quant_rows_e4m3(t):
    """What padt_quant_rows_fp8 + the fp8 MFMA see of a row: bf16 storage first, then e4m3 codes under the power-of-two row scale
    2^ceil(log2(amax / 448)) — returned de-quantised (fp32)."""
    t = t.to(torch.bfloat16).to(torch.float32)
    amax = t.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, torch.exp2(torch.ceil(torch.log2(amax.clamp_min(1e-30) / 448.0))), torch.ones_like(amax))
    return (t / scale).to(torch.float8_e4m3fn).to(torch.float32) * scale


class fp8_prefill_hooks:
    """Context manager for models built with llm_weights="fp8" and the fp8 x fp8 MFMA prompt pass (model.W.fp8_prefill): inside it the
    oracle's LLM layer quantises, AT PROMPT LENGTH ONLY (Lq > 1 — decode steps multiply bf16 activations with the fp8 weights), the input
    rows of exactly those projections the HIP path runs through padt_gemm_fp8, the way padt_quant_rows_fp8 does.  Use together with
    effective_llm_weights (dequantised matrices, norm weights folded): the oracle then computes the same function of the input as the HIP
    path, up to accumulation order and the occasional e4m3 code that flips because the two sides' bf16 roundings of an activation differ."""

    def __init__(self, model):
        W = model.W
        self.on = bool(getattr(W, "fp8_prefill", False))
        self.use = {nm: ("llm.0.%s.w8" % nm) in W for nm in ("qkv", "o", "gu", "down")}

    def __enter__(self):
        import torch.nn.functional as F
        self._saved = O.llm_layer
        if not self.on:
            return self
        use = self.use
        fq = fake_quant_rows_e4m3

        def llm_layer(w, pfx, cfg, h, cos, sin, attn_bias, cache, li):
            B, Lq, _ = h.shape
            if Lq == 1:
                return self._saved(w, pfx, cfg, h, cos, sin, attn_bias, cache, li)

            def normed(x, wn, on):
                if not on:
                    return O.rms_norm(x, wn, cfg.rms_eps)
                xb = x.to(torch.bfloat16).to(torch.float32)          # the bf16 mirror the quantiser reads (statistics from it too)
                return wn * (fq(x) * torch.rsqrt(xb.pow(2).mean(-1, keepdim=True) + cfg.rms_eps))
            n = normed(h, w[pfx + "input_layernorm.weight"], use["qkv"])
            q = O.linear(n, w[pfx + "self_attn.q_proj.weight"], w[pfx + "self_attn.q_proj.bias"]).view(B, Lq, cfg.num_heads, cfg.head_dim)
            k = O.linear(n, w[pfx + "self_attn.k_proj.weight"], w[pfx + "self_attn.k_proj.bias"]).view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
            v = O.linear(n, w[pfx + "self_attn.v_proj.weight"], w[pfx + "self_attn.v_proj.bias"]).view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
            c, s = cos.unsqueeze(2), sin.unsqueeze(2)
            q = q * c + O.rotate_half(q) * s
            k = k * c + O.rotate_half(k) * s
            if cache is not None:
                k, v = cache.update(li, k, v)
            rep = cfg.num_heads // cfg.num_kv_heads
            qh = q.transpose(1, 2).float()
            kh = k.transpose(1, 2).repeat_interleave(rep, 1).float()
            vh = v.transpose(1, 2).repeat_interleave(rep, 1).float()
            sc = torch.matmul(qh, kh.transpose(2, 3)) * (cfg.head_dim ** -0.5) + attn_bias
            a = torch.matmul(torch.softmax(sc, dim=-1), vh).transpose(1, 2).reshape(B, Lq, -1)
            h = h + O.linear(fq(a) if use["o"] else a, w[pfx + "self_attn.o_proj.weight"])
            n = normed(h, w[pfx + "post_attention_layernorm.weight"], use["gu"])
            g = O.linear(n, w[pfx + "mlp.gate_proj.weight"])
            u = O.linear(n, w[pfx + "mlp.up_proj.weight"])
            m = F.silu(g) * u
            return h + O.linear(fq(m) if use["down"] else m, w[pfx + "mlp.down_proj.weight"])

        O.llm_layer = llm_layer
        return self

    def __exit__(self, *exc):
        O.llm_layer = self._saved
        return False
