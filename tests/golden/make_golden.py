"""Generate golden vectors by importing the REFERENCE (read-only, /root/reference) under shims.

Run in the build container only:   python tests/golden/make_golden.py
The reference cannot travel to the GPU box; what travels are the small ``.npz`` fixtures this
script writes next to itself, plus this script.  Nothing in tests/, bench.py or
smoke() imports this file or reads /root/reference at run time.

Shims, none of which restate PaDT logic:
  1. an empty ``deepspeed`` module (padt.py:4 / padt_decoder.py:3 import it, never use it on this path);
  2. three names padt_decoder.py:8 imports from HF's qwen2_5_vl module and 5.15 no longer exports:
     ``Qwen2RMSNorm`` (alias of the HF RMSNorm), ``apply_rotary_emb`` and ``flash_attn_varlen_func``
     (pure-torch statements of flash-attn 2.7.4's public semantics: non-interleaved rotary;
     per-segment fp32 softmax(QK^T d^-1/2)V, non-causal);
  3. ``sys.path += /root/reference/src``.
What runs UNMODIFIED from the reference: ``custom_visual_forward`` (installed on HF's ViT by padt.py:108),
``PaDTForConditionalGeneration.forward_main`` and ``.vl_decode`` (as unbound methods on an adapter object that
owns HF's 5.15 ViT / text model), ``PaDTDecoder``, ``VisonTextProcessingClass``, ``parseVRTintoCompletion``.
``get_rope_index`` is NOT reference/HF-5.15 code (5.15 changed its signature and delta rule); the adapter uses
the oracle's 4.50 restatement, so position ids are unpinned (stated in oracle/padt_oracle.py and DESIGN.md).
"""
import importlib.machinery
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import padt_oracle as O  # noqa: E402


# ------------------------------------------------------------------ shims
def install_shims():
    import transformers.generation.utils  # noqa: F401  (must precede the deepspeed stub)
    ds = types.ModuleType("deepspeed")
    ds.__spec__ = importlib.machinery.ModuleSpec("deepspeed", None)
    sys.modules["deepspeed"] = ds
    import transformers.models.qwen2_5_vl.modeling_qwen2_5_vl as M

    def apply_rotary_emb(x, cos, sin, interleaved=False, inplace=False):
        ro = cos.shape[-1] * 2
        c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
        x1, x2 = x[..., : ro // 2], x[..., ro // 2: ro]
        return torch.cat([x1 * c - x2 * s, x1 * s + x2 * c, x[..., ro:]], dim=-1)

    def flash_attn_varlen_func(q, k, v, cu_q, cu_k, max_q, max_k, dropout_p=0.0, softmax_scale=None, causal=False):
        assert not causal
        out = torch.empty_like(q)
        scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
        for i in range(len(cu_q) - 1):
            qs = q[cu_q[i]:cu_q[i + 1]].transpose(0, 1).float()
            ks = k[cu_k[i]:cu_k[i + 1]].transpose(0, 1).float()
            vs = v[cu_k[i]:cu_k[i + 1]].transpose(0, 1).float()
            p = torch.softmax(qs @ ks.transpose(1, 2) * scale, dim=-1)
            out[cu_q[i]:cu_q[i + 1]] = (p @ vs).transpose(0, 1).to(q.dtype)
        return out

    M.Qwen2RMSNorm = M.Qwen2_5_VLRMSNorm
    M.apply_rotary_emb = apply_rotary_emb
    M.flash_attn_varlen_func = flash_attn_varlen_func
    sys.path.insert(0, "/root/reference/src")
    import PaDT  # noqa: F401
    return M


# ------------------------------------------------------------------ reference-side model assembly
def build_reference(M, cfg: O.OracleConfig, w):
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig, Qwen2_5_VLVisionConfig
    from PaDT.models.padt import PaDTForConditionalGeneration
    from PaDT.models.padt_decoder import PaDTDecoder

    vcfg = Qwen2_5_VLVisionConfig(
        depth=cfg.vit_depth, hidden_size=cfg.vit_hidden, hidden_act="silu", intermediate_size=cfg.vit_intermediate,
        num_heads=cfg.vit_heads, in_channels=cfg.in_channels, patch_size=cfg.patch_size,
        spatial_merge_size=cfg.spatial_merge_size, temporal_patch_size=cfg.temporal_patch_size,
        window_size=cfg.window_size, fullatt_block_indexes=list(cfg.fullatt_block_indexes),
        out_hidden_size=cfg.hidden_size)
    vcfg._attn_implementation = "eager"
    tcfg = Qwen2_5_VLTextConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
        rms_norm_eps=cfg.rms_eps, tie_word_embeddings=cfg.tie_word_embeddings, use_sliding_window=False,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)},
        pad_token_id=None)
    tcfg._attn_implementation = "eager"
    visual = M.Qwen2_5_VisionTransformerPretrainedModel(vcfg).eval()
    text = M.Qwen2_5_VLTextModel(tcfg).eval()
    assert tcfg.head_dim == cfg.head_dim if hasattr(tcfg, "head_dim") and tcfg.head_dim else True

    missing = visual.load_state_dict({k[len("visual."):]: v for k, v in w.items() if k.startswith("visual.")}, strict=True)
    text.load_state_dict({k[len("model."):]: v for k, v in w.items() if k.startswith("model.")}, strict=True)
    dec_cfg = {"hidden_size": cfg.dec_hidden, "intermediate_size": cfg.dec_intermediate, "num_heads": cfg.dec_heads,
               "use_mask_loss": cfg.use_mask_loss, "attn_implementation": "flash_attention_2",
               "spatial_merge_size": cfg.spatial_merge_size, "llm_hidden_state": cfg.hidden_size}
    decoder = PaDTDecoder(dec_cfg, torch.float32).eval()
    decoder.load_state_dict({k[len("vl_decoder."):]: v for k, v in w.items() if k.startswith("vl_decoder.")}, strict=True)

    class TextAdapter:
        """`.model` of the reference class: has embed_tokens and is callable like HF's text model."""
        def __init__(self, m):
            self.m = m
            self.embed_tokens = m.embed_tokens

        def __call__(self, **kw):
            kw.pop("cache_position", None)
            kw.pop("output_attentions", None)
            kw.pop("return_dict", None)
            return self.m(**kw)

    class Adapter:
        forward_main = PaDTForConditionalGeneration.forward_main
        vl_decode = PaDTForConditionalGeneration.vl_decode

        def __init__(self):
            self.config = types.SimpleNamespace(
                vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, image_token_id=cfg.image_token_id,
                video_token_id=-1, tie_word_embeddings=cfg.tie_word_embeddings, output_attentions=False,
                output_hidden_states=False, use_return_dict=True,
                vision_config=types.SimpleNamespace(spatial_merge_size=cfg.spatial_merge_size,
                                                    hidden_size=cfg.vit_hidden, num_heads=cfg.vit_heads))
            self.visual = visual
            self.model = TextAdapter(text)
            self.rope_deltas = None
            self.use_visual_prototype_projection = cfg.use_visual_prototype_projection
            if cfg.use_visual_prototype_projection:
                self.vis_norm = torch.nn.LayerNorm(cfg.hidden_size)
                self.vis_norm.weight.data.copy_(w["vis_norm.weight"])
                self.vis_norm.bias.data.copy_(w["vis_norm.bias"])
                self.vis_proj = torch.nn.Sequential(torch.nn.Linear(cfg.hidden_size, cfg.lora_r, bias=False),
                                                    torch.nn.Linear(cfg.lora_r, cfg.hidden_size, bias=False))
                self.vis_proj[0].weight.data.copy_(w["vis_proj.0.weight"])
                self.vis_proj[1].weight.data.copy_(w["vis_proj.1.weight"])
            if not cfg.tie_word_embeddings:
                self.lm_head = torch.nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)
                self.lm_head.weight.data.copy_(w["lm_head.weight"])
            self.vl_decoder = decoder
            self.device = torch.device("cpu")
            self.dtype = torch.float32

        def get_rope_index(self, input_ids=None, image_grid_thw=None, video_grid_thw=None,
                           second_per_grid_ts=None, attention_mask=None):
            return O.rope_index(cfg, input_ids, image_grid_thw, attention_mask)   # 4.50 restatement (unpinned)

    return Adapter()


# ------------------------------------------------------------------ fake tokenizer for the processor/parser
class FakeTokenizer:
    """Minimal tokenizer surface used by padt_processor.py (get_vocab / add_tokens / vocab / eos_token)."""
    def __init__(self, words, eos_token):
        self.id2tok = list(words)
        self.eos_token = eos_token

    def get_vocab(self):
        return {t: i for i, t in enumerate(self.id2tok)}

    @property
    def vocab(self):
        return self.get_vocab()

    def add_tokens(self, toks):
        for t in toks:
            s = t.content if hasattr(t, "content") else str(t)
            if s not in self.id2tok:
                self.id2tok.append(s)
        return len(toks)


class FakeProcessor:
    def __init__(self, tok):
        self.tokenizer = tok

    def batch_decode(self, ids):
        return [self.tokenizer.id2tok[int(i)] for i in ids]

    def __call__(self, *a, **k):
        return {"image_grid_thw": k["image_grid_thw"]} if "image_grid_thw" in k else {}


BASE_WORDS = ["<|endoftext|>", "<|im_end|>", "The", " ", "\"", " \"", "\" ", "car", " car", "person", " person",
              " refers", " to", " in", " this", " image", ".", ",", " (", ")", "In", " there", " are", " 2", " 1",
              "<", "answer", ">", "</", " on the", " left", "think"]


def make_tiny_cfg():
    return O.OracleConfig(
        vocab_size=512, hidden_size=64, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=16, intermediate_size=96,
        mrope_section=(2, 3, 3), vit_hidden=32, vit_depth=4, vit_heads=2, vit_intermediate=48, patch_size=2,
        temporal_patch_size=2, in_channels=3, window_size=16, fullatt_block_indexes=(1, 3), lora_r=8,
        dec_hidden=32, dec_heads=2, dec_intermediate=48, image_token_id=500, vision_start_token_id=501,
        eos_token_id=1, pad_token_id=0)


def tiny_inputs(cfg, seed=7):
    """Two images of different size (10x12 and 6x8 patches) → left-padded batch."""
    g = torch.Generator().manual_seed(seed)
    grid = torch.tensor([[1, 10, 12], [1, 6, 8]])
    P = int((grid[:, 1] * grid[:, 2]).sum())
    pix = torch.randn(P, cfg.patch_dim, generator=g)
    n = O.merged_counts(cfg, grid).tolist()
    rows = []
    for b in range(2):
        pre = torch.randint(2, 400, (3 + b,), generator=g).tolist()
        post = torch.randint(2, 400, (5,), generator=g).tolist()
        # one prompt-side VRT reference per sample (LOCAL id; exercises assign_to_global_vrt_id)
        rows.append(pre + [cfg.vision_start_token_id] + [cfg.image_token_id] * n[b] + [502] + post + [cfg.vocab_size + 3 + b])
    L = max(len(r) for r in rows)
    ids = torch.full((2, L), cfg.pad_token_id, dtype=torch.long)
    am = torch.zeros((2, L), dtype=torch.long)
    for b, r in enumerate(rows):
        ids[b, L - len(r):] = torch.tensor(r)
        am[b, L - len(r):] = 1
    return grid, pix, ids, am


def npz_save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    M = install_shims()
    from PaDT.models.padt_processor import VisonTextProcessingClass, parseVRTintoCompletion

    # ============================================================ 1. index tables (int, bit-exact)
    tables = {}
    vis = M.Qwen2_5_VisionTransformerPretrainedModel
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    real_v = Qwen2_5_VLVisionConfig(depth=1, hidden_size=32, num_heads=2, intermediate_size=8, out_hidden_size=16)
    vm = vis(real_v)
    for name, g in (("46x46", [[1, 46, 46]]), ("46x30", [[1, 46, 30]]), ("10x12", [[1, 10, 12]]),
                    ("batch", [[1, 46, 46], [1, 30, 46], [1, 8, 8]])):
        gt = torch.tensor(g)
        wi, cu = vm.get_window_index(gt)
        cu = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
        from transformers.vision_utils import get_vision_position_ids
        tables[f"{name}.grid"] = gt
        tables[f"{name}.window_index"] = wi
        tables[f"{name}.cu_window"] = cu
        tables[f"{name}.pos_ids"] = get_vision_position_ids(gt, 2)
    npz_save("index_tables.npz", **tables)

    # ============================================================ 2. tiny end-to-end
    cfg = make_tiny_cfg()
    w = O.synthetic_weights(cfg, seed=11, std=0.08, bias_std=0.05, norm_jitter=0.1)
    w["vis_norm.bias"] = w["vis_norm.bias"] + 0.03
    ref = build_reference(M, cfg, w)
    grid, pix, ids_local, am = tiny_inputs(cfg)

    tok = FakeTokenizer(BASE_WORDS, "<|im_end|>")
    proc = VisonTextProcessingClass(FakeProcessor(tok), cfg.spatial_merge_size)
    proc.prepare(cfg.vocab_size)
    proc(image_grid_thw=grid)                                   # set_image_grid_thw → adds <|VRT_i|>
    vocab_after = len(tok.get_vocab())
    ids_global = proc.assign_to_global_vrt_id(ids_local.clone(), grid)
    ids_back = proc.assign_to_local_vrt_id(ids_global.clone(), grid)
    assert torch.equal(ids_back, ids_local)

    # --- ViT (custom_visual_forward, unmodified)
    low, high, (cos, sin) = ref.visual(pix, grid_thw=grid)

    # --- prefill (forward_main unmodified)
    cache_pos = torch.arange(ids_global.shape[1])
    from transformers.cache_utils import DynamicCache
    cache = DynamicCache()
    out = ref.forward_main(input_ids=ids_global, attention_mask=am, past_key_values=cache, use_cache=True,
                           output_hidden_states=True, return_dict=True, pixel_values=pix, image_grid_thw=grid,
                           cache_position=cache_pos)
    prefill_logits = out.logits
    prefill_hidden = out.hidden_states[-1]
    proto = out.past_image_embeds
    lmask = out.past_logit_mask
    rope_deltas = ref.rope_deltas.clone()

    # --- decode: teacher-forced completion (so every parser/decoder stage is exercised deterministically)
    V = cfg.vocab_size
    n_m = O.merged_counts(cfg, grid).tolist()
    t = {s: i for i, s in enumerate(tok.id2tok)}
    comp_local = [
        # sample 0 (REC-style): The "car" refers to <VRT..> in this image.<eos>
        [t["The"], t[" \""], t["car"], t["\" "], t[" refers"], t[" to"], V + 7, V + 2, V + 11, t[" in"], t[" this"],
         t[" image"], t["."], t["<|im_end|>"]],
        # sample 1 (OVD-style): In there are 2 "person" (<V><V>, <V>) .<eos>  (second run reuses the label)
        [t["In"], t[" there"], t[" are"], t[" 2"], t[" \""], t["person"], t["\" "], t[" ("], V + 1, V + 5, t[","],
         V + 9, t[")"], t["<|im_end|>"]],
    ]
    comp_local = torch.tensor(comp_local)
    comp_global = proc.assign_to_global_vrt_id(comp_local.clone(), grid)
    T = comp_local.shape[1]
    hidden_steps = [out.hidden_states]                      # step 0 = prefill tuple (predicts completion token 0)
    step_logits = [prefill_logits[:, -1]]
    attn = am
    kw = dict(past_image_embeds=out.past_image_embeds, past_logit_mask=out.past_logit_mask,
              past_high_res_image_embeds=out.past_high_res_image_embeds, past_visual_pe=out.past_visual_pe)
    L = ids_global.shape[1]
    for s in range(T - 1):
        attn = torch.cat([attn, attn.new_ones(2, 1)], dim=1)
        o = ref.forward_main(input_ids=comp_global[:, s:s + 1], attention_mask=attn, past_key_values=cache,
                             use_cache=True, output_hidden_states=True, return_dict=True,
                             cache_position=torch.tensor([L + s]), **kw)
        hidden_steps.append(o.hidden_states)
        step_logits.append(o.logits[:, -1])
    step_logits = torch.stack(step_logits, dim=1)            # (B,T,V+N)
    last_hidden = torch.stack([h[-1][:, -1] for h in hidden_steps], dim=1)   # (B,T,D)

    # --- parser (reference, unmodified)
    completions, feats, labels, vrts, _ = parseVRTintoCompletion(proc, comp_local, hidden_steps, torch.Tensor([False, False]))
    assert [len(f) for f in feats] == [1, 2], [len(f) for f in feats]

    # --- vl_decode (reference, unmodified)
    dec = ref.vl_decode(feats, out.past_image_embeds, out.past_high_res_image_embeds, grid, out.past_visual_pe)

    # parser edge cases: truncated VRT run (no EOS → whole sample dropped), thinking mask / <answer> tags, empty
    edge_ids = torch.tensor([
        [t["The"], t[" \""], t["car"], t["\" "], V + 1, V + 2, V + 3],                      # run hits the end → except → []
        [t["<"], t["answer"], t[">"], t[" \""], t["car"], t["\" "], V + 4, t["</"], t["answer"], t[">"]][:7] ,
    ])
    edge_hidden = [(torch.full((2, 1, cfg.hidden_size), float(i)),) for i in range(edge_ids.shape[1])]
    e_comp, e_feats, e_labels, e_vrts, _ = parseVRTintoCompletion(proc, edge_ids, edge_hidden, torch.Tensor([False, True]))

    pack = dict(
        grid=grid, pixel_values=pix, input_ids_local=ids_local, input_ids_global=ids_global, attention_mask=am,
        vocab_after=np.int64(vocab_after), image_embeds=low, high_res=high, cos=cos, sin=sin, proto=proto,
        logit_mask=lmask, prefill_logits_last=prefill_logits[:, -1], prefill_logits_pos3=prefill_logits[:, 3],
        prefill_hidden=prefill_hidden, rope_deltas=rope_deltas, comp_local=comp_local, comp_global=comp_global,
        step_logits=step_logits, last_hidden=last_hidden,
        pred_boxes=dec["pred_boxes"], pred_score=dec["pred_score"], pred_mask=dec["pred_mask"],
        valid_h=dec["pred_mask_valid_hw"][0], valid_w=dec["pred_mask_valid_hw"][1],
        sample_idx=np.array(dec["sample_idx"]),
        completions=np.array(completions), labels=np.array([";".join(l) for l in labels]),
        vrts=np.array(["|".join(v) for v in vrts]),
        feat_0_0=feats[0][0], feat_1_0=feats[1][0], feat_1_1=feats[1][1],
        edge_ids=edge_ids, edge_n_feats=np.array([len(f) for f in e_feats]),
        edge_labels=np.array([";".join(l) for l in e_labels]), edge_completions=np.array(e_comp),
        edge_feat_1_0=e_feats[1][0] if len(e_feats[1]) else np.zeros(0),
        tokenizer_words=np.array(tok.id2tok),
    )
    for k, v in w.items():
        pack["w::" + k] = v
    npz_save("tiny_e2e.npz", **pack)

    # ============================================================ 3. real-shape single pieces (sampled outputs)
    g = torch.Generator().manual_seed(123)
    rcfg = O.OracleConfig()
    # 3a. one ViT block at 2116x1280, window and full segmentation
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig as VC
    vc = VC(depth=32, hidden_size=1280, hidden_act="silu", intermediate_size=3420, num_heads=16, in_channels=3,
            patch_size=14, spatial_merge_size=2, temporal_patch_size=2, window_size=112,
            fullatt_block_indexes=[7, 15, 23, 31], out_hidden_size=2048)
    vc._attn_implementation = "eager"
    blk = M.Qwen2_5_VLVisionBlock(vc).eval()
    shapes = {k: v for k, v in O.weight_shapes(rcfg).items() if k.startswith("visual.blocks.0.")}
    bw = {}
    for k, shp in shapes.items():
        gg = torch.Generator().manual_seed(hash_name(k))
        if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            bw[k] = 1 + 0.1 * torch.randn(shp, generator=gg)
        elif k.endswith("bias"):
            bw[k] = 0.02 * torch.randn(shp, generator=gg)
        else:
            bw[k] = 0.02 * torch.randn(shp, generator=gg)
    blk.load_state_dict({k[len("visual.blocks.0."):]: v for k, v in bw.items()})
    grid1 = torch.tensor([[1, 46, 46]])
    x = torch.randn(2116, 1280, generator=g)
    wi, cu_win = O.window_index(grid1, 2, 112, 14)
    c1, s1 = O.vit_rotary(rcfg, grid1, wi)
    rows = torch.randint(0, 2116, (48,), generator=g)
    y_win = blk(x, cu_seqlens=torch.tensor(cu_win, dtype=torch.int32), position_embeddings=(c1, s1))
    y_full = blk(x, cu_seqlens=torch.tensor([0, 2116], dtype=torch.int32), position_embeddings=(c1, s1))
    o_win = O.vit_block(bw, "visual.blocks.0.", rcfg, x, cu_win, c1, s1)
    o_full = O.vit_block(bw, "visual.blocks.0.", rcfg, x, [0, 2116], c1, s1)
    print("real ViT block: oracle vs reference max|d| win %.3e full %.3e" % ((o_win - y_win).abs().max(), (o_full - y_full).abs().max()))
    npz_save("real_vit_block.npz", rows=rows, y_win=y_win[rows], y_full=y_full[rows], x_seed=np.int64(123))

    # 3b. PaDTDecoder real config (1280/16/3420), 2 objects on a 46x30 image + 1 object on 8x8
    from PaDT.models.padt_decoder import PaDTDecoder
    dcfg = {"hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "use_mask_loss": True,
            "attn_implementation": "flash_attention_2", "spatial_merge_size": 2, "llm_hidden_state": 2048}
    dec_m = PaDTDecoder(dcfg, torch.float32).eval()
    dshapes = {k: v for k, v in O.weight_shapes(rcfg).items() if k.startswith("vl_decoder.")}
    dw = {}
    for k, shp in dshapes.items():
        gg = torch.Generator().manual_seed(hash_name(k))
        if O._is_norm_weight(k):
            dw[k] = 1 + 0.1 * torch.randn(shp, generator=gg)
        elif k.endswith("bias"):
            dw[k] = 0.02 * torch.randn(shp, generator=gg)
        else:
            dw[k] = 0.03 * torch.randn(shp, generator=gg)
    dec_m.load_state_dict({k[len("vl_decoder."):]: v for k, v in dw.items()})
    grids = torch.tensor([[1, 46, 30], [1, 8, 8]])
    Ps = [46 * 30, 64]
    low_all = torch.randn(sum(Ps) // 4, 2048, generator=g)
    high_all = torch.randn(sum(Ps), 1280, generator=g)
    wi2, _ = O.window_index(grids, 2, 112, 14)
    c2, s2 = O.vit_rotary(rcfg, grids, wi2)
    feats_r = [[torch.randn(5, 2048, generator=g), torch.randn(2, 2048, generator=g)], [torch.randn(4, 2048, generator=g)]]
    rad = types.SimpleNamespace(vl_decoder=dec_m, device=torch.device("cpu"), dtype=torch.float32,
                                config=types.SimpleNamespace(hidden_size=2048))
    from PaDT.models.padt import PaDTForConditionalGeneration
    rdec = PaDTForConditionalGeneration.vl_decode(rad, feats_r, low_all, high_all, grids, (c2, s2))
    odec = O.vl_decode(dw, rcfg, feats_r, low_all, high_all, grids, (c2, s2))
    print("real decoder: oracle vs reference max|d| box %.3e score %.3e mask %.3e" % (
        (odec["pred_boxes"] - rdec["pred_boxes"]).abs().max(), (odec["pred_score"] - rdec["pred_score"]).abs().max(),
        (odec["pred_mask"] - rdec["pred_mask"]).abs().max()))
    mi = torch.randint(0, rdec["pred_mask"].numel(), (4096,), generator=g)
    npz_save("real_decoder.npz", pred_boxes=rdec["pred_boxes"], pred_score=rdec["pred_score"],
             mask_idx=mi, mask_vals=rdec["pred_mask"].flatten()[mi], mask_shape=np.array(rdec["pred_mask"].shape),
             valid_h=rdec["pred_mask_valid_hw"][0], valid_w=rdec["pred_mask_valid_hw"][1],
             sample_idx=np.array(rdec["sample_idx"]))


def hash_name(k: str) -> int:
    import zlib
    return zlib.crc32(k.encode()) & 0x7FFFFFFF


if __name__ == "__main__":
    main()
