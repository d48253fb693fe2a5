"""Checkpoint → MI355X weight layout.

Input: a state dict with the checkpoint's (HF-4.50) key names — ``visual.*``, ``model.*``, ``lm_head.weight``,
``vis_norm.*``, ``vis_proj.*``, ``vl_decoder.*`` — from safetensors shards or from
``synthetic_state_dict`` (no checkpoints exist offline).  Output: ``PreparedWeights``, 16-bit device tensors (fp16 by default for
ViT / LLM, bf16 for the PaDT decoder: ``prepare_weights``) laid out for the kernels in libpadt_hip.so:
  * every Linear stays [out][in] (K-contiguous = MFMA operand order), conv3d patch-embed flattened to [hidden][C*T*p*p];
  * LLM q/k/v fused into one [Hq*D + 2*Hkv*D][hidden] matrix (+ fused bias); the LLM's input / post-attention RMSNorm
    weights folded into the q/k/v and gate/up matrices (W·diag(g));
  * SwiGLU gate/up interleaved in 16-row blocks ([gate16 | up16] ...) so one GEMM tile holds matching gate/up columns
    and the activation is fused into the epilogue; MLP intermediates zero-padded to a multiple of 64 (3420 → 3456).
"""
import os
import zlib
from typing import Dict, Optional

import torch

from .config import PaDTConfig

BF16 = torch.bfloat16


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# ------------------------------------------------------------------------------------------------- key → shape table
def weight_shapes(cfg: PaDTConfig) -> Dict[str, tuple]:
    v = cfg.vision_config
    s: Dict[str, tuple] = {}
    vh, vi = v.hidden_size, v.intermediate_size
    s["visual.patch_embed.proj.weight"] = (vh, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
    for i in range(v.depth):
        p = f"visual.blocks.{i}."
        s[p + "norm1.weight"] = (vh,)
        s[p + "norm2.weight"] = (vh,)
        s[p + "attn.qkv.weight"] = (3 * vh, vh)
        s[p + "attn.qkv.bias"] = (3 * vh,)
        s[p + "attn.proj.weight"] = (vh, vh)
        s[p + "attn.proj.bias"] = (vh,)
        for n_, a, b in (("gate_proj", vi, vh), ("up_proj", vi, vh), ("down_proj", vh, vi)):
            s[p + f"mlp.{n_}.weight"] = (a, b)
            s[p + f"mlp.{n_}.bias"] = (a,)
    mh = vh * cfg.merge_unit
    s["visual.merger.ln_q.weight"] = (vh,)
    s["visual.merger.mlp.0.weight"] = (mh, mh)
    s["visual.merger.mlp.0.bias"] = (mh,)
    s["visual.merger.mlp.2.weight"] = (cfg.hidden_size, mh)
    s["visual.merger.mlp.2.bias"] = (cfg.hidden_size,)
    D, I, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    s["model.embed_tokens.weight"] = (cfg.vocab_size, D)
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (D,)
        s[p + "post_attention_layernorm.weight"] = (D,)
        s[p + "self_attn.q_proj.weight"] = (cfg.num_attention_heads * hd, D)
        s[p + "self_attn.q_proj.bias"] = (cfg.num_attention_heads * hd,)
        for n_ in ("k_proj", "v_proj"):
            s[p + f"self_attn.{n_}.weight"] = (cfg.num_key_value_heads * hd, D)
            s[p + f"self_attn.{n_}.bias"] = (cfg.num_key_value_heads * hd,)
        s[p + "self_attn.o_proj.weight"] = (D, cfg.num_attention_heads * hd)
        s[p + "mlp.gate_proj.weight"] = (I, D)
        s[p + "mlp.up_proj.weight"] = (I, D)
        s[p + "mlp.down_proj.weight"] = (D, I)
    s["model.norm.weight"] = (D,)
    if not cfg.tie_word_embeddings:
        s["lm_head.weight"] = (cfg.vocab_size, D)
    if cfg.use_visual_prototype_projection:
        s["vis_norm.weight"] = (D,)
        s["vis_norm.bias"] = (D,)
        s["vis_proj.0.weight"] = (cfg.lora_r, D)
        s["vis_proj.1.weight"] = (D, cfg.lora_r)
    dh, di = cfg.vl_decoder["hidden_size"], cfg.vl_decoder["intermediate_size"]
    p = "vl_decoder."
    s[p + "vp_embedding.weight"] = (1, dh)
    s[p + "bbox_score_mask_tokens.weight"] = (3, dh)
    s[p + "input_projection.0.weight"] = (D,)
    s[p + "input_projection.1.weight"] = (dh, D)
    s[p + "input_projection.1.bias"] = (dh,)
    s[p + "input_projection.3.weight"] = (dh, dh)
    s[p + "input_projection.3.bias"] = (dh,)
    for blk in ("low_res_transformer", "high_res_transformer1", "high_res_transformer2"):
        b = p + blk + "."
        for k in range(1, 7):
            s[b + f"norm{k}.weight"] = (dh,)
        for att in ("self_attn", "cross_attn_query_to_image", "cross_attn_image_to_query"):
            for pr in ("q_proj", "k_proj", "v_proj", "proj"):
                s[b + f"{att}.{pr}.weight"] = (dh, dh)
                s[b + f"{att}.{pr}.bias"] = (dh,)
        s[b + "mlp.0.weight"] = (di, dh)
        s[b + "mlp.0.bias"] = (di,)
        s[b + "mlp.2.weight"] = (dh, di)
        s[b + "mlp.2.bias"] = (dh,)
    s[p + "high_res_norm.weight"] = (dh,)
    for name, last in (("bbox_prediction", 4), ("mask_output_mlp", dh // 16)):
        s[p + name + ".0.weight"] = (dh, dh)
        s[p + name + ".0.bias"] = (dh,)
        s[p + name + ".2.weight"] = (dh, dh)
        s[p + name + ".2.bias"] = (dh,)
        s[p + name + ".4.weight"] = (last, dh)
        s[p + name + ".4.bias"] = (last,)
    s[p + "score_prediction.weight"] = (1, dh)
    s[p + "score_prediction.bias"] = (1,)
    s[p + "mask_output_upscaling1.0.weight"] = (dh // 4 * 4, dh)
    s[p + "mask_output_upscaling1.0.bias"] = (dh // 4 * 4,)
    s[p + "mask_output_upscaling1.1.weight"] = (dh // 4 * 4,)
    s[p + "mask_output_upscaling2.0.weight"] = (dh // 16 * 4, dh // 4)
    s[p + "mask_output_upscaling2.0.bias"] = (dh // 16 * 4,)
    return s


def is_norm_weight(name: str) -> bool:
    return (name.endswith("norm.weight") or "layernorm.weight" in name or ".ln_q.weight" in name
            or any(name.endswith(f"norm{k}.weight") for k in range(1, 7))
            or name.endswith("input_projection.0.weight") or name.endswith("mask_output_upscaling1.1.weight"))


def synthetic_state_dict(cfg: PaDTConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.0,
                         norm_jitter: float = 0.0, device="cpu", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights: N(0,std²) matrices, norm weights 1 (+jitter), biases 0 (or N(0,bias_std²)).

    On CPU every tensor has its own generator seeded from (seed, crc32(key)) — the rule the test oracle uses, so both
    sides can build identical weights independently.  On a CUDA device the same rule seeds a device generator (fast
    path for the 3.85 B-parameter bench model; values differ from the CPU stream, distribution identical).
    """
    out = {}
    dev = torch.device(device)
    for name, shape in weight_shapes(cfg).items():
        g = torch.Generator(device=dev).manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        if is_norm_weight(name):
            t = torch.ones(shape, device=dev)
            if norm_jitter > 0:
                t = t + torch.randn(shape, generator=g, device=dev) * norm_jitter
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g, device=dev) * bias_std if bias_std > 0 else torch.zeros(shape, device=dev)
        else:
            t = torch.randn(shape, generator=g, device=dev) * std
        out[name] = t.to(dtype)
    return out


def load_checkpoint_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Read every ``*.safetensors`` shard under ``path`` (HF layout, optional index json)."""
    from safetensors.torch import load_file
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        sd.update(load_file(os.path.join(path, f)))
    # 5.x-style nesting ("model.language_model.", "model.visual.") → 4.50 names
    ren = {}
    for k, v in sd.items():
        k2 = k
        if k2.startswith("model.language_model."):
            k2 = "model." + k2[len("model.language_model."):]
        elif k2.startswith("model.visual."):
            k2 = k2[len("model."):]
        ren[k2] = v
    return ren


# ------------------------------------------------------------------------------------------------- prepared layout
def interleave_rope_rows(w, n_heads, head_dim):
    """Row permutation of a projection's leading n_heads*head_dim rows (weight (N, K) or bias (N,)): within each head, row
    2i <- d = i and row 2i + 1 <- d = i + head_dim/2, i.e. rotation pairs become adjacent output columns."""
    half = head_dim // 2
    idx = torch.arange(n_heads * head_dim, device=w.device).view(n_heads, 2, half).transpose(1, 2).reshape(-1)
    out = w.clone()
    out[: n_heads * head_dim] = w[: n_heads * head_dim][idx]
    return out


def interleave16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[gate16 | up16 | gate16 | up16 ...] along dim 0 (rows already padded to a multiple of 16)."""
    n = a.shape[0]
    rest = a.shape[1:]
    return torch.stack([a.reshape(n // 16, 16, *rest), b.reshape(n // 16, 16, *rest)], dim=1).reshape(2 * n, *rest)


def _pad_rows(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[0] == n:
        return t
    out = t.new_zeros((n,) + tuple(t.shape[1:]))
    out[: t.shape[0]] = t
    return out


def _pad_cols(t: torch.Tensor, n: int) -> torch.Tensor:
    if t.shape[1] == n:
        return t
    out = t.new_zeros((t.shape[0], n))
    out[:, : t.shape[1]] = t
    return out


def fp8_gemm_ok(n: int, k: int) -> bool:
    """Shapes padt_gemm_fp8 takes: whole 256-column tiles and whole 128-element K-tiles."""
    return n % 256 == 0 and k % 128 == 0


class Fp16RangeError(ValueError):
    """A checkpoint value (or a norm-folded product) does not fit fp16: the model must multiply bf16 operands."""


class PreparedWeights(dict):
    """name → device tensor in kernel layout (ViT / LLM / prototype tensors in the operand type `op16`, the PaDT decoder's in bf16).
    Plain dict plus a few derived sizes."""
    vit_ipad: int
    llm_ipad: int
    dec_ipad: int
    dec_hp: bool = False
    llm_weights: str = "bf16"
    resid_f32: bool = True
    fp8_prefill: bool = False
    op16: torch.dtype = torch.bfloat16

    def eps_m(self, eps: float) -> float:
        """eps for the consumers of a residual-stream mirror (ops.mirror_eps)."""
        from .ops import mirror_eps
        return mirror_eps(eps, self.op16)


def prepare_weights(sd: Dict[str, torch.Tensor], cfg: PaDTConfig, device="cuda", llm_weights: str = "bf16",
                    operands: Optional[str] = None) -> PreparedWeights:
    """operands = "fp16" (default; env PADT_OPERANDS) | "bf16": the 16-bit MFMA operand type of ViT / LLM / prototypes — activations between
    kernels, KV caches and the weight images.  Checkpoints are bf16: a bf16 weight is exact in fp16 unless |w| < 2^-14 (then within 3e-8), and
    the norm-folded matrices w_norm * W are ROUNDED to the operand type, which fp16 does with 11 instead of 8 mantissa bits.  fp16 operands sit
    8x closer to the fp32 reference at the bf16 MFMA rate (tests/studies/operand_attribution.py) and need the fp32 residual streams (an
    un-normalised stream does not fit fp16's range; its 16-bit mirror is stored scaled, ops.stream_scale).
    llm_weights = "fp8": the LLM's projection matrices (after the norm folding) are quantised per output row to OCP e4m3 with
    power-of-two scales (ops.quantize_fp8_rows); the decode step streams the fp8 images (half the bytes), prefill multiplies with the
    16-bit image of the SAME numbers (scale * q is exact) — BASELINE configs[4], the 7B "fp8 MFMA weight path".  "fp8+act" additionally
    runs the prompt pass as fp8 x fp8 MFMA GEMMs over e4m3 ACTIVATION rows quantised on the fly (1.6x the 16-bit tile GEMM on the 7B shapes;
    costs precision: DESIGN.md §4 numerics) — opt-in since round 4 (it was implied by "fp8" in round 3)."""
    if llm_weights not in ("bf16", "fp8", "fp8+act"):
        raise ValueError("llm_weights must be 'bf16', 'fp8' or 'fp8+act'")
    operands = operands or os.environ.get("PADT_OPERANDS", "fp16")
    if operands not in ("bf16", "fp16"):
        raise ValueError("operands must be 'fp16' or 'bf16'")
    dev = torch.device(device)
    W = PreparedWeights()
    fp8_act = llm_weights == "fp8+act" or (llm_weights == "fp8" and os.environ.get("PADT_FP8_PREFILL", "0") == "1")
    llm_weights = "fp8" if llm_weights == "fp8+act" else llm_weights
    W.llm_weights = llm_weights
    # fp32 residual streams in the ViT and the LLM (default; PADT_RESID_F32=0 keeps the round-2 bf16 streams for A/B runs): the
    # residual GEMMs' epilogues update an fp32 stream in place and emit its bf16 mirror for the next projection
    W.resid_f32 = os.environ.get("PADT_RESID_F32", "1") != "0"
    if operands == "fp16" and not W.resid_f32:
        raise ValueError("fp16 operands need the fp32 residual streams (PADT_RESID_F32=0 is a bf16-only A/B switch)")
    W.op16 = torch.float16 if operands == "fp16" else BF16
    # "fp8+act": at prompt length the LLM projections run as fp8 x fp8 MFMA GEMMs (activation rows quantised to e4m3 on the fly,
    # v_mfma_f32_16x16x128_f8f6f4) wherever the shape allows.  Needs the fp32 residual streams.
    W.fp8_prefill = llm_weights == "fp8" and W.resid_f32 and fp8_act
    op16 = W.op16

    out_of_range = []                                       # (name, device bool): fp16 images holding inf / NaN, read back once at the end

    def put(name, t):
        # everything under "dec." belongs to the split-precision PaDT decoder, whose (hi, lo) operand pairs are bf16
        W[name] = t.to(device=dev, dtype=BF16 if name.startswith("dec.") else op16).contiguous()
        if W[name].dtype == torch.float16:
            out_of_range.append((name, (~torch.isfinite(W[name])).any()))

    def get(name):
        if name not in sd:
            raise KeyError(f"checkpoint is missing '{name}'")
        return sd[name]

    v = cfg.vision_config
    vi_pad = _pad_to(v.intermediate_size, 64)
    W.vit_ipad = vi_pad
    put("vit.patch_embed", get("visual.patch_embed.proj.weight").reshape(v.hidden_size, -1))
    for i in range(v.depth):
        s, d = f"visual.blocks.{i}.", f"vit.{i}."
        # norm1 / norm2 weights folded into qkv / gate-up like the LLM's (below): the GEMM reads x itself and scales its
        # accumulator with rstd[row] (ops.row_rstd + gemm(row_scale=)) — the normalised activations are never materialised
        n1 = get(s + "norm1.weight").float().to(dev)[None, :]
        n2 = get(s + "norm2.weight").float().to(dev)[None, :]
        # q and k rows pair-interleaved per head ((d, d + hd/2) adjacent) so the qkv GEMM's epilogue can apply the rotary
        # embedding lane-locally (ops.gemm_rope); q·k scores are invariant under the common permutation, v is untouched
        hd_v = v.hidden_size // v.num_heads
        W.vit_rope_fused = hd_v % 4 == 0                               # else: plain qkv GEMM + rope_half pass
        il = (lambda t: interleave_rope_rows(t, 2 * v.num_heads, hd_v)) if W.vit_rope_fused else (lambda t: t)
        put(d + "qkv.w", il(get(s + "attn.qkv.weight").to(dev).float() * n1))
        put(d + "qkv.b", il(get(s + "attn.qkv.bias").to(dev)))
        put(d + "proj.w", get(s + "attn.proj.weight"))
        put(d + "proj.b", get(s + "attn.proj.bias"))
        put(d + "gu.w", interleave16(_pad_rows(get(s + "mlp.gate_proj.weight").to(dev).float() * n2, vi_pad),
                                     _pad_rows(get(s + "mlp.up_proj.weight").to(dev).float() * n2, vi_pad)))
        put(d + "gu.b", interleave16(_pad_rows(get(s + "mlp.gate_proj.bias"), vi_pad), _pad_rows(get(s + "mlp.up_proj.bias"), vi_pad)))
        put(d + "down.w", _pad_cols(get(s + "mlp.down_proj.weight"), vi_pad))
        put(d + "down.b", get(s + "mlp.down_proj.bias"))
    put("vit.merger.ln_q", get("visual.merger.ln_q.weight"))
    put("vit.merger.0.w", get("visual.merger.mlp.0.weight"))
    put("vit.merger.0.b", get("visual.merger.mlp.0.bias"))
    put("vit.merger.2.w", get("visual.merger.mlp.2.weight"))
    put("vit.merger.2.b", get("visual.merger.mlp.2.bias"))

    li_pad = _pad_to(cfg.intermediate_size, 64)
    W.llm_ipad = li_pad
    put("llm.embed", get("model.embed_tokens.weight"))
    put("llm.head", get("model.embed_tokens.weight") if cfg.tie_word_embeddings else get("lm_head.weight"))
    if cfg.tie_word_embeddings:
        W["llm.head"] = W["llm.embed"]
    for i in range(cfg.num_hidden_layers):
        s, d = f"model.layers.{i}.", f"llm.{i}."
        # RMSNorm weights are folded into the projection that consumes the normalised activations:
        #   (x·rstd ⊙ g) Wᵀ = rstd · x (W·diag(g))ᵀ  — one rounding of W·g to bf16 at load time, so the decode GEMV can
        #   fuse the norm (rstd only) and prefill scales the GEMM accumulator with rstd[row] (ops.row_rstd + row_scale).
        g1 = get(s + "input_layernorm.weight").float().to(dev)[None, :]
        g2 = get(s + "post_attention_layernorm.weight").float().to(dev)[None, :]
        qkv = torch.cat([get(s + "self_attn.q_proj.weight"), get(s + "self_attn.k_proj.weight"), get(s + "self_attn.v_proj.weight")], 0)
        put(d + "qkv.w", qkv.to(dev).float() * g1)
        put(d + "qkv.b", torch.cat([get(s + "self_attn.q_proj.bias"), get(s + "self_attn.k_proj.bias"), get(s + "self_attn.v_proj.bias")], 0))
        put(d + "o.w", get(s + "self_attn.o_proj.weight"))
        put(d + "gu.w", interleave16(_pad_rows(get(s + "mlp.gate_proj.weight").to(dev).float() * g2, li_pad),
                                     _pad_rows(get(s + "mlp.up_proj.weight").to(dev).float() * g2, li_pad)))
        put(d + "down.w", _pad_cols(get(s + "mlp.down_proj.weight"), li_pad))
    # decode-step copies of the LLM matrices in MFMA-fragment order (ops.pack_weight): the single-token GEMVs stream
    # them with fully coalesced 1 KiB wave loads.  +5.5 GB for PaDT_Pro_3B — HBM capacity is not the constraint here.
    from .ops import pack_weight, pack_weight_fp8, quantize_fp8_rows
    for i in range(cfg.num_hidden_layers):
        d = f"llm.{i}."
        for nm in ("qkv", "o", "gu", "down"):
            if llm_weights == "fp8":
                q, sc, deq = quantize_fp8_rows(W[d + nm + ".w"], deq_dtype=op16)
                W[d + nm + ".w"] = deq                           # prefill: 16-bit image of the quantised matrix (exact)
                W[d + nm + ".wq"] = pack_weight_fp8(q)           # decode: fp8 fragment-packed image + per-row scales
                W[d + nm + ".ws"] = sc
                if W.fp8_prefill and fp8_gemm_ok(q.shape[0], q.shape[1]):
                    W[d + nm + ".w8"] = q.contiguous()           # prefill: row-major e4m3 image for the fp8 x fp8 MFMA GEMM (padt_gemm_fp8)
            else:
                W[d + nm + ".wp"] = pack_weight(W[d + nm + ".w"])
    # fragment-packed copy of the head table for the decode-step logit head (+0.62 GB at 3B; the row-major table stays: it is
    # the embedding table too, and the prefill-side gathers read rows)
    if cfg.vocab_size % 16 == 0 and cfg.hidden_size % 32 == 0:
        W["llm.head.wp"] = pack_weight(W["llm.head"])
    put("llm.norm", get("model.norm.weight"))
    W["llm.ones"] = torch.ones(cfg.hidden_size, device=dev, dtype=op16)
    if cfg.use_visual_prototype_projection:
        put("proto.norm.w", get("vis_norm.weight"))
        put("proto.norm.b", get("vis_norm.bias"))
        put("proto.0.w", get("vis_proj.0.weight"))
        put("proto.1.w", get("vis_proj.1.weight"))

    if out_of_range:
        bad = torch.stack([b for _, b in out_of_range]).cpu().tolist()
        names = [n for (n, _), b in zip(out_of_range, bad) if b]
        if names:
            raise Fp16RangeError("values outside fp16's range (65504) in the fp16 images of %s%s: build the model with operands='bf16' "
                                 "(operands='auto' does so by itself)" % (", ".join(names[:4]), " ..." if len(names) > 4 else ""))

    dh, di = cfg.vl_decoder["hidden_size"], cfg.vl_decoder["intermediate_size"]
    di_pad = _pad_to(di, 64)
    W.dec_ipad = di_pad
    p = "vl_decoder."
    for name in ("vp_embedding.weight", "bbox_score_mask_tokens.weight", "input_projection.0.weight",
                 "input_projection.1.weight", "input_projection.1.bias", "input_projection.3.weight",
                 "input_projection.3.bias", "high_res_norm.weight", "score_prediction.weight", "score_prediction.bias",
                 "mask_output_upscaling1.0.weight", "mask_output_upscaling1.0.bias", "mask_output_upscaling1.1.weight",
                 "mask_output_upscaling2.0.weight", "mask_output_upscaling2.0.bias"):
        put("dec." + name, get(p + name))
    for head in ("bbox_prediction", "mask_output_mlp"):
        for k in (0, 2, 4):
            put(f"dec.{head}.{k}.weight", get(p + f"{head}.{k}.weight"))
            put(f"dec.{head}.{k}.bias", get(p + f"{head}.{k}.bias"))
    for blk in ("low_res_transformer", "high_res_transformer1", "high_res_transformer2"):
        s, d = p + blk + ".", "dec." + blk + "."
        for k in range(1, 7):
            put(d + f"norm{k}", get(s + f"norm{k}.weight"))
        for att in ("self_attn", "cross_attn_query_to_image", "cross_attn_image_to_query"):
            for pr in ("q_proj", "k_proj", "v_proj", "proj"):
                put(d + f"{att}.{pr}.w", get(s + f"{att}.{pr}.weight"))
                put(d + f"{att}.{pr}.b", get(s + f"{att}.{pr}.bias"))
        put(d + "mlp.0.w", _pad_rows(get(s + "mlp.0.weight"), di_pad))
        put(d + "mlp.0.b", _pad_rows(get(s + "mlp.0.bias"), di_pad))
        put(d + "mlp.2.w", _pad_cols(get(s + "mlp.2.weight"), di_pad))
        put(d + "mlp.2.b", get(s + "mlp.2.bias"))
    # Split-precision decoder (csrc/decoder_hp.hip, default): every decoder Linear also as the image [W | W] along K, so that the
    # bf16 MFMA GEMM over an activation stored as [hi | lo] pairs accumulates hi·W + lo·W (16-bit-mantissa activations) with
    # unchanged kernels at K' = 2K; projections that share their input are stacked along N (self-attn q|k, query→image k|v).
    # +0.4 GB for the 98 M-parameter decoder.  PADT_DECODER_HP=0 keeps the plain bf16 decoder only.
    W.dec_hp = os.environ.get("PADT_DECODER_HP", "1") != "0"
    if W.dec_hp:
        def dbl(*names):
            t = torch.cat([W[n] for n in names], dim=0) if len(names) > 1 else W[names[0]]
            return torch.cat([t, t], dim=1).contiguous()
        W["dec.tokens.f32"] = W["dec.bbox_score_mask_tokens.weight"].float()
        W["dec.vp.f32"] = W["dec.vp_embedding.weight"].float()
        for name in ("input_projection.1.weight", "input_projection.3.weight", "score_prediction.weight",
                     "mask_output_upscaling1.0.weight", "mask_output_upscaling2.0.weight"):
            W["dec." + name + ".hp"] = dbl("dec." + name)
        for head in ("bbox_prediction", "mask_output_mlp"):
            for k in (0, 2, 4):
                W[f"dec.{head}.{k}.weight.hp"] = dbl(f"dec.{head}.{k}.weight")
        for blk in ("low_res_transformer", "high_res_transformer1", "high_res_transformer2"):
            d = "dec." + blk + "."
            a = d + "self_attn."
            W[a + "qk.hp"] = dbl(a + "q_proj.w", a + "k_proj.w")
            W[a + "qk.b"] = torch.cat([W[a + "q_proj.b"], W[a + "k_proj.b"]]).contiguous()
            W[a + "v_proj.hp"] = dbl(a + "v_proj.w")
            W[a + "proj.hp"] = dbl(a + "proj.w")
            a = d + "cross_attn_query_to_image."
            W[a + "q_proj.hp"] = dbl(a + "q_proj.w")
            W[a + "kv.hp"] = dbl(a + "k_proj.w", a + "v_proj.w")
            W[a + "kv.b"] = torch.cat([W[a + "k_proj.b"], W[a + "v_proj.b"]]).contiguous()
            W[a + "proj.hp"] = dbl(a + "proj.w")
            a = d + "cross_attn_image_to_query."
            for pr in ("q_proj", "k_proj", "v_proj", "proj"):
                W[a + pr + ".hp"] = dbl(a + pr + ".w")
            W[d + "mlp.0.hp"] = dbl(d + "mlp.0.w")
            W[d + "mlp.2.hp"] = dbl(d + "mlp.2.w")
    return W
