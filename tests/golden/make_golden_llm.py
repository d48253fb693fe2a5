"""Real-width LLM fixture: ONE Qwen2.5-VL decoder layer at PaDT_Pro_3B width (hidden 2048, 16 q / 2 kv heads x 128, MLP 11008,
mRoPE sections [16, 24, 24]) + final norm, run with the INSTALLED transformers' `Qwen2_5_VLTextModel` (the module the reference
calls at padt.py:279-291; 5.15 here, same per-layer math as the pinned 4.50 — SURVEY.md §8c) on a real prompt layout
(15 text + 529 image + 33 text tokens, grid 46x46 → L = 577) with the 4.50 position ids, then two single-token decode steps
through its DynamicCache.  Weights and inputs are regenerated from seeds by the tests; only sampled outputs are stored.

run:  HF_HUB_OFFLINE=1 TRANSFORMERS_OFFLINE=1 python tests/golden/make_golden_llm.py      (writes tests/golden/real_llm_layer.npz)
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import padt_oracle as O  # noqa: E402

V = 1024                                                            # small vocabulary: the embedding table is not under test


def cfg_1layer():
    return O.OracleConfig(vocab_size=V, num_layers=1, image_token_id=V - 3, vision_start_token_id=V - 2, eos_token_id=V - 1,
                          pad_token_id=V - 4)


def seeded(shape, name, scale, one=False):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    t = scale * torch.randn(shape, generator=g)
    return 1 + t if one else t


def layer_weights(cfg):
    w = {}
    for k, shp in O.weight_shapes(cfg).items():
        if k.startswith("model.layers.0.") or k == "model.norm.weight":
            if k.endswith("layernorm.weight") or k == "model.norm.weight":
                w[k] = seeded(shp, k, 0.1, True)
            elif k.endswith("bias"):
                w[k] = seeded(shp, k, 0.05)
            else:
                w[k] = seeded(shp, k, 0.02)
    return w


def prompt(cfg):
    ids = torch.cat([torch.arange(10, 24), torch.tensor([cfg.vision_start_token_id]), torch.full((529,), cfg.image_token_id),
                     torch.arange(30, 63)])[None]
    assert ids.shape[1] == 577
    return ids, torch.tensor([[1, 46, 46]])


def main():
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as M
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig
    cfg = cfg_1layer()
    w = layer_weights(cfg)
    tcfg = Qwen2_5_VLTextConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=1,
        num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads, rms_norm_eps=cfg.rms_eps,
        tie_word_embeddings=True, use_sliding_window=False,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)},
        pad_token_id=None)
    tcfg._attn_implementation = "eager"
    text = M.Qwen2_5_VLTextModel(tcfg).eval()
    sd = {k[len("model."):]: v for k, v in w.items()}
    sd["embed_tokens.weight"] = torch.zeros(V, cfg.hidden_size)
    text.load_state_dict(sd, strict=True)
    ids, grid = prompt(cfg)
    pos, deltas = O.rope_index(cfg, ids, grid, torch.ones_like(ids))            # (3, 1, L), 4.50 rule
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(1, 577, cfg.hidden_size, generator=g)
    xd = torch.randn(2, 1, 1, cfg.hidden_size, generator=g)                     # two decode-step inputs
    with torch.no_grad():
        out = text(inputs_embeds=x, position_ids=pos, use_cache=True)
        h0, cache = out.last_hidden_state, out.past_key_values
        steps = []
        for t in range(2):
            p = (torch.tensor([[577 + t]]) + deltas).view(1, 1, 1).expand(3, 1, 1)
            o = text(inputs_embeds=xd[t], position_ids=p, past_key_values=cache, use_cache=True)
            steps.append(o.last_hidden_state[0, 0])
            cache = o.past_key_values
    # the oracle on the same tensors (must agree before anything is written)
    kv = O.KVCache(1)
    oh = O.llm_forward(w, cfg, x, pos, torch.ones(1, 577, dtype=torch.long), kv)
    d0 = (oh - h0).abs().max().item()
    print("oracle vs HF text model, prefill max|d| = %.3e" % d0)
    assert d0 < 5e-4
    rows = torch.randint(0, 577, (64,), generator=g)
    np.savez_compressed(os.path.join(HERE, "real_llm_layer.npz"), rows=rows.numpy(), h_rows=h0[0, rows].numpy(),
                        h_last=h0[0, -1].numpy(), step0=steps[0].numpy(), step1=steps[1].numpy(), x_seed=np.array(4242),
                        deltas=deltas.numpy())
    print("wrote real_llm_layer.npz", h0.shape, float(h0.abs().max()))


if __name__ == "__main__":
    main()
