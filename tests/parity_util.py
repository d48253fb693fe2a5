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
