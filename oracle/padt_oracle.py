"""CPU oracle for the PaDT generate-with-VRT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 by default) restatement of the reference
algorithm.  It exists so that the HIP path can be checked against something that
can travel to the GPU box (the reference itself cannot).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
Nothing under ``padt_amd/`` imports it; the product path fails loudly when the HIP
extension is missing.

Pinning: the reference has no tests and its only fixture needs real weights
(SURVEY.md §4).  The oracle is pinned against outputs of the reference itself,
imported in the build container under three shims (``tests/golden/make_golden.py``)
with seeded synthetic weights; the resulting vectors are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` replays them.  Not covered by
the reference import and therefore "parity unpinned" (restated from upstream
transformers==4.50.0, which is not in the container): the greedy loop in ``generate``
(padt.py:670-762 calls 4.50-only GenerationMixin helpers) and two 4.50 conventions of
``rope_index`` (filler 1 at padded positions, rope_deltas against the PADDED length).
The valid-token positions of ``rope_index`` (incl. left-padded ragged batches),
``warp_logits``, ``pil_resample`` and ``patchify_normalize`` are pinned against the
installed transformers 5.15 / Pillow (tests/test_host_logic_cpu.py, test_preprocess_cpu.py).

Citations are ``file:line`` into the reference repository (``src/PaDT/models/...``)
or ``HF:`` = transformers ``models/qwen2_5_vl/modeling_qwen2_5_vl.py`` /
``vision_utils.py`` as described in SURVEY.md Appendix A.

Weights are a flat ``dict[str, Tensor]`` using the checkpoint (HF-4.50) key names:
``visual.*``, ``model.*``, ``lm_head.weight``, ``vis_norm.*``, ``vis_proj.*``,
``vl_decoder.*``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class OracleConfig:
    # LLM
    vocab_size: int = 151936
    hidden_size: int = 2048
    num_layers: int = 36
    num_heads: int = 16
    num_kv_heads: int = 2
    head_dim: int = 128
    intermediate_size: int = 11008
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    tie_word_embeddings: bool = True
    # ViT
    vit_hidden: int = 1280
    vit_depth: int = 32
    vit_heads: int = 16
    vit_intermediate: int = 3420
    patch_size: int = 14
    temporal_patch_size: int = 2
    in_channels: int = 3
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    # PaDT
    use_visual_prototype_projection: bool = True
    lora_r: int = 64
    dec_hidden: int = 1280
    dec_heads: int = 16
    dec_intermediate: int = 3420
    use_mask_loss: bool = True
    # special ids
    image_token_id: int = 151655
    vision_start_token_id: int = 151652
    eos_token_id: int = 151645
    pad_token_id: int = 151643

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size ** 2

    @property
    def merge_unit(self) -> int:
        return self.spatial_merge_size ** 2


def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """HF:74-79 — upcast, x*rsqrt(mean(x^2)+eps), cast back, then *weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    return F.linear(x, w, b)


def rotate_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def varlen_attention(q: Tensor, k: Tensor, v: Tensor, cu_q: Sequence[int], cu_k: Sequence[int],
                     causal: bool = False) -> Tensor:
    """Per-segment softmax(QK^T/sqrt(d))V in fp32; q (Tq,H,d), k/v (Tk,Hkv,d).

    Stands in for flash_attn_varlen_func (padt_decoder.py:55; HF ViT/LLM attention).
    With ``causal`` the mask is bottom-right aligned (flash-attn convention), which is
    what prefill (Tq==Tk) and decode (Tq==1) both need.
    """
    H, d = q.shape[1], q.shape[2]
    Hkv = k.shape[1]
    rep = H // Hkv
    out = torch.empty_like(q)
    scale = d ** -0.5
    for s in range(len(cu_q) - 1):
        q0, q1 = int(cu_q[s]), int(cu_q[s + 1])
        k0, k1 = int(cu_k[s]), int(cu_k[s + 1])
        if q1 == q0:
            continue
        qs = q[q0:q1].transpose(0, 1).float()                       # H,Lq,d
        ks = k[k0:k1].transpose(0, 1).float().repeat_interleave(rep, 0)
        vs = v[k0:k1].transpose(0, 1).float().repeat_interleave(rep, 0)
        sc = torch.matmul(qs, ks.transpose(1, 2)) * scale           # H,Lq,Lk
        if causal:
            Lq, Lk = q1 - q0, k1 - k0
            mask = torch.ones(Lq, Lk, dtype=torch.bool).tril(diagonal=Lk - Lq)
            sc = sc.masked_fill(~mask, float("-inf"))
        p = torch.softmax(sc, dim=-1)
        out[q0:q1] = torch.matmul(p, vs).transpose(0, 1).to(q.dtype)
    return out


# --------------------------------------------------------------------------- ViT index prep (int)
def vision_position_ids(grid_thw: Tensor, merge: int) -> Tensor:
    """HF:vision_utils.py:111-127 — (h,w) ids laid out in merge x merge block-major order."""
    out = []
    for t, h, w in grid_thw.tolist():
        hp = torch.arange(h).unsqueeze(1).expand(h, w)
        wp = torch.arange(w).unsqueeze(0).expand(h, w)
        shp = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shp).permute(0, 2, 1, 3).flatten()
        wp = wp.reshape(shp).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def window_index(grid_thw: Tensor, merge: int, window_size: int, patch_size: int) -> Tuple[Tensor, List[int]]:
    """HF:vision_utils.py:155-188 + padt.py:62-67 (unique_consecutive on the cu list).

    Returns (window_index over merged tokens, cu_window_seqlens in units of patches).
    """
    win = window_size // merge // patch_size
    unit = merge * merge
    idx_all, cu, base = [], [0], 0
    for t, h, w in grid_thw.tolist():
        lh, lw = h // merge, w // merge
        index = torch.arange(t * lh * lw).reshape(t, lh, lw)
        pad_h = win - lh % win
        pad_w = win - lw % win
        nh, nw = (lh + pad_h) // win, (lw + pad_w) // win
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        ip = ip.reshape(-1)
        idx_all.append(ip[ip != -100] + base)
        cs = seqlens.cumsum(0) * unit + cu[-1]
        cu.extend(cs.tolist())
        base += t * lh * lw
    cu_t = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
    return torch.cat(idx_all), cu_t.tolist()


# --------------------------------------------------------------------------- ViT
def vit_rotary(cfg: OracleConfig, grid_thw: Tensor, win_idx: Tensor) -> Tuple[Tensor, Tensor]:
    """padt.py:60,73-77 — rotary table permuted to window order; cos/sin (P, head_dim) fp32."""
    hd = cfg.vit_hidden // cfg.vit_heads
    dim = hd // 2
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    pos = vision_position_ids(grid_thw, cfg.spatial_merge_size)
    freqs = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)        # (P, hd/2)
    P = freqs.shape[0]
    freqs = freqs.reshape(P // cfg.merge_unit, cfg.merge_unit, -1)[win_idx].reshape(P, -1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def vit_block(w: Dict[str, Tensor], pfx: str, cfg: OracleConfig, x: Tensor, cu: Sequence[int],
              cos: Tensor, sin: Tensor) -> Tensor:
    """HF:211-321 — x += proj(attn(rope(qkv(norm1 x)))); x += down(silu(gate n)*up n)."""
    H = cfg.vit_heads
    T = x.shape[0]
    n = rms_norm(x, w[pfx + "norm1.weight"], 1e-6)
    qkv = linear(n, w[pfx + "attn.qkv.weight"], w[pfx + "attn.qkv.bias"]).reshape(T, 3, H, -1)
    q, k, v = qkv.permute(1, 0, 2, 3).unbind(0)
    c, s = cos.unsqueeze(-2).float(), sin.unsqueeze(-2).float()
    qf, kf = q.float(), k.float()
    q = (qf * c + rotate_half(qf) * s).to(x.dtype)
    k = (kf * c + rotate_half(kf) * s).to(x.dtype)
    a = varlen_attention(q, k, v, cu, cu, causal=False).reshape(T, -1)
    x = x + linear(a, w[pfx + "attn.proj.weight"], w[pfx + "attn.proj.bias"])
    n = rms_norm(x, w[pfx + "norm2.weight"], 1e-6)
    g = linear(n, w[pfx + "mlp.gate_proj.weight"], w[pfx + "mlp.gate_proj.bias"])
    u = linear(n, w[pfx + "mlp.up_proj.weight"], w[pfx + "mlp.up_proj.bias"])
    x = x + linear(F.silu(g) * u, w[pfx + "mlp.down_proj.weight"], w[pfx + "mlp.down_proj.bias"])
    return x


def vit_forward(w: Dict[str, Tensor], cfg: OracleConfig, pixel_values: Tensor, grid_thw: Tensor,
                collect: Optional[dict] = None):
    """custom_visual_forward, padt.py:48-106.

    Returns (image_embeds (N,D) raster order, high_res (P,vit_hidden) WINDOW order, (cos,sin) window order).
    """
    dt = w["visual.patch_embed.proj.weight"].dtype
    pw = w["visual.patch_embed.proj.weight"].reshape(cfg.vit_hidden, -1)
    x = linear(pixel_values.to(dt), pw)                                   # conv3d stride=kernel == GEMM (HF:116-122)
    win_idx, cu_win = window_index(grid_thw, cfg.spatial_merge_size, cfg.window_size, cfg.patch_size)
    P = x.shape[0]
    x = x.reshape(P // cfg.merge_unit, cfg.merge_unit, -1)[win_idx].reshape(P, -1)
    cos, sin = vit_rotary(cfg, grid_thw, win_idx)
    seg = torch.repeat_interleave(grid_thw[:, 1] * grid_thw[:, 2], grid_thw[:, 0]).cumsum(0)
    cu_full = [0] + seg.tolist()
    for i in range(cfg.vit_depth):
        cu = cu_full if i in cfg.fullatt_block_indexes else cu_win
        x = vit_block(w, f"visual.blocks.{i}.", cfg, x, cu, cos, sin)
        if collect is not None:
            collect.setdefault("vit_block_out", []).append(x)
    high = x
    n = rms_norm(x, w["visual.merger.ln_q.weight"], 1e-6).reshape(-1, cfg.vit_hidden * cfg.merge_unit)
    m = linear(n, w["visual.merger.mlp.0.weight"], w["visual.merger.mlp.0.bias"])
    m = linear(F.gelu(m), w["visual.merger.mlp.2.weight"], w["visual.merger.mlp.2.bias"])
    low = m[torch.argsort(win_idx)]
    return low, high, (cos, sin)


# --------------------------------------------------------------------------- VRT table
def prototypes(w: Dict[str, Tensor], cfg: OracleConfig, image_embeds: Tensor) -> Tensor:
    """padt.py:187-191 — LayerNorm(eps 1e-5) then + W2(W1 x)."""
    if not cfg.use_visual_prototype_projection:
        return image_embeds.clone()
    p = F.layer_norm(image_embeds, (image_embeds.shape[-1],), w["vis_norm.weight"], w["vis_norm.bias"], 1e-5)
    return p + linear(linear(p, w["vis_proj.0.weight"]), w["vis_proj.1.weight"])


def merged_counts(cfg: OracleConfig, grid_thw: Tensor) -> Tensor:
    return (grid_thw[:, 1] * grid_thw[:, 2]) // cfg.merge_unit


def logit_mask(cfg: OracleConfig, grid_thw: Tensor, table_rows: int) -> Tensor:
    """padt.py:196-201."""
    n_img = grid_thw.shape[0]
    m = torch.zeros((n_img, table_rows), dtype=torch.bool)
    m[:, :cfg.vocab_size] = True
    pn = F.pad(merged_counts(cfg, grid_thw).cumsum(0), (1, 0), value=0).tolist()
    for i in range(n_img):
        m[i, cfg.vocab_size + pn[i]: cfg.vocab_size + pn[i + 1]] = True
    return m


def embed_inputs(w: Dict[str, Tensor], cfg: OracleConfig, input_ids: Tensor, proto: Tensor,
                 image_embeds: Optional[Tensor]) -> Tensor:
    """padt.py:193-219 (prefill) / 226-229 (decode)."""
    table = torch.cat([w["model.embed_tokens.weight"], proto], dim=0)
    assert int(input_ids.max()) < table.shape[0]
    x = table[input_ids]
    if image_embeds is not None:
        n_tok = int((input_ids == cfg.image_token_id).sum())
        if n_tok != image_embeds.shape[0]:
            raise ValueError(
                f"Image features and image tokens do not match: tokens: {n_tok}, features {image_embeds.shape[0]}")
        mask = (input_ids == cfg.image_token_id).unsqueeze(-1).expand_as(x)
        x = x.masked_scatter(mask, image_embeds.to(x.dtype))
    return x


# --------------------------------------------------------------------------- positions (4.50 semantics; valid-token ids pinned vs 5.15)
def rope_index(cfg: OracleConfig, input_ids: Tensor, grid_thw: Tensor,
               attention_mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """transformers==4.50.0 ``get_rope_index`` restated (called at padt.py:263).

    text run: same running index on t/h/w; image: t=start, h=start+row, w=start+col over the merged grid;
    next start = previous max + 1; padded positions keep 1; rope_deltas = max+1 - PADDED length.
    """
    B, L = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    pos = torch.ones(3, B, L, dtype=input_ids.dtype)
    deltas = []
    img = 0
    m = cfg.spatial_merge_size
    for b in range(B):
        keep = attention_mask[b] == 1
        toks = input_ids[b][keep].tolist()
        chunks: List[Tensor] = []
        st = 0
        n_img = 0
        for i, tk in enumerate(toks[:-1]):
            if tk == cfg.vision_start_token_id and toks[i + 1] == cfg.image_token_id:
                n_img += 1
        for _ in range(n_img):
            ed = toks.index(cfg.image_token_id, st)
            t, h, wd = grid_thw[img].tolist()
            img += 1
            lt, lh, lw = t, h // m, wd // m
            text_len = ed - st
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            chunks.append(torch.arange(text_len).view(1, -1).expand(3, -1) + st_idx)
            ti = torch.zeros(lt * lh * lw, dtype=torch.long)          # images: second_per_grid_t == 0
            hi = torch.arange(lh).view(1, -1, 1).expand(lt, -1, lw).flatten()
            wi = torch.arange(lw).view(1, 1, -1).expand(lt, lh, -1).flatten()
            chunks.append(torch.stack([ti, hi, wi]) + text_len + st_idx)
            st = ed + lt * lh * lw
        if st < len(toks):
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            chunks.append(torch.arange(len(toks) - st).view(1, -1).expand(3, -1) + st_idx)
        llm_pos = torch.cat(chunks, dim=1).reshape(3, -1)
        pos[:, b, keep] = llm_pos.to(pos.dtype)
        deltas.append(int(llm_pos.max()) + 1 - L)
    return pos, torch.tensor(deltas, dtype=input_ids.dtype).unsqueeze(1)


def mrope_cos_sin(cfg: OracleConfig, position_ids: Tensor, dtype: torch.dtype) -> Tuple[Tensor, Tensor]:
    """HF:525-538 + 589-595 — per-axis cos/sin then section interleave; returns (B,L,head_dim)."""
    hd = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    freqs = position_ids[..., None].float() * inv                      # 3,B,L,hd/2
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos().to(dtype), emb.sin().to(dtype)
    sec = list(cfg.mrope_section) * 2
    cos = torch.cat([c[i % 3] for i, c in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([s[i % 3] for i, s in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


# --------------------------------------------------------------------------- LLM
class KVCache:
    def __init__(self, n_layers: int):
        self.k: List[Optional[Tensor]] = [None] * n_layers
        self.v: List[Optional[Tensor]] = [None] * n_layers

    def update(self, i: int, k: Tensor, v: Tensor):
        if self.k[i] is None:
            self.k[i], self.v[i] = k, v
        else:
            self.k[i] = torch.cat([self.k[i], k], dim=1)
            self.v[i] = torch.cat([self.v[i], v], dim=1)
        return self.k[i], self.v[i]

    def seq_len(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[1]


def llm_layer(w: Dict[str, Tensor], pfx: str, cfg: OracleConfig, h: Tensor, cos: Tensor, sin: Tensor,
              attn_bias: Tensor, cache: Optional[KVCache], li: int) -> Tensor:
    """HF:641-757 — pre-norm GQA layer with mRoPE and KV cache. h (B,Lq,D); attn_bias (B,1,Lq,Lk) additive."""
    B, Lq, _ = h.shape
    n = rms_norm(h, w[pfx + "input_layernorm.weight"], cfg.rms_eps)
    q = linear(n, w[pfx + "self_attn.q_proj.weight"], w[pfx + "self_attn.q_proj.bias"]).view(B, Lq, cfg.num_heads, cfg.head_dim)
    k = linear(n, w[pfx + "self_attn.k_proj.weight"], w[pfx + "self_attn.k_proj.bias"]).view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
    v = linear(n, w[pfx + "self_attn.v_proj.weight"], w[pfx + "self_attn.v_proj.bias"]).view(B, Lq, cfg.num_kv_heads, cfg.head_dim)
    c, s = cos.unsqueeze(2), sin.unsqueeze(2)
    q = q * c + rotate_half(q) * s
    k = k * c + rotate_half(k) * s
    if cache is not None:
        k, v = cache.update(li, k, v)
    rep = cfg.num_heads // cfg.num_kv_heads
    qh = q.transpose(1, 2).float()
    kh = k.transpose(1, 2).repeat_interleave(rep, 1).float()
    vh = v.transpose(1, 2).repeat_interleave(rep, 1).float()
    sc = torch.matmul(qh, kh.transpose(2, 3)) * (cfg.head_dim ** -0.5) + attn_bias
    p = torch.softmax(sc, dim=-1)
    a = torch.matmul(p, vh).transpose(1, 2).reshape(B, Lq, -1).to(h.dtype)
    h = h + linear(a, w[pfx + "self_attn.o_proj.weight"])
    n = rms_norm(h, w[pfx + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = linear(n, w[pfx + "mlp.gate_proj.weight"])
    u = linear(n, w[pfx + "mlp.up_proj.weight"])
    return h + linear(F.silu(g) * u, w[pfx + "mlp.down_proj.weight"])


def llm_forward(w: Dict[str, Tensor], cfg: OracleConfig, inputs_embeds: Tensor, position_ids: Tensor,
                attention_mask: Tensor, cache: Optional[KVCache], all_hidden: bool = False):
    """HF:790-872 — returns post-final-norm hidden (B,Lq,D) (+ optional per-layer tuple, HF convention:
    entry 0 = embeddings, entries 1..n-1 = layer outputs, last = post-norm)."""
    B, Lq, _ = inputs_embeds.shape
    past = cache.seq_len() if cache is not None else 0
    Lk = past + Lq
    cos, sin = mrope_cos_sin(cfg, position_ids, inputs_embeds.dtype)
    qpos = torch.arange(past, past + Lq).view(Lq, 1)
    kpos = torch.arange(Lk).view(1, Lk)
    allow = (kpos <= qpos).view(1, 1, Lq, Lk) & (attention_mask[:, None, None, :Lk] == 1)
    bias = torch.zeros(B, 1, Lq, Lk).masked_fill(~allow, float("-inf"))
    # rows that are fully masked (left padding) would give NaN; give them a harmless uniform row
    dead = ~allow.any(-1, keepdim=True)
    bias = bias.masked_fill(dead.expand_as(bias), 0.0)
    h = inputs_embeds
    hs = [h]
    for i in range(cfg.num_layers):
        h = llm_layer(w, f"model.layers.{i}.", cfg, h, cos, sin, bias, cache, i)
        hs.append(h)
    h = rms_norm(h, w["model.norm.weight"], cfg.rms_eps)
    hs[-1] = h
    return (h, tuple(hs)) if all_hidden else h


def vrt_logits(w: Dict[str, Tensor], cfg: OracleConfig, hidden: Tensor, proto: Tensor, lmask: Tensor) -> Tensor:
    """padt.py:292-301 — hidden @ [E|lm_head ‖ proto]^T, then -inf where ~logit_mask[b]."""
    head = w["model.embed_tokens.weight"] if cfg.tie_word_embeddings else w["lm_head.weight"]
    table = torch.cat([head, proto], dim=0)
    logits = hidden @ table.T
    return logits.masked_fill(~lmask[:, None, :].expand(-1, logits.shape[1], -1), float("-inf"))


@dataclass
class PrefillState:
    proto: Tensor
    lmask: Tensor
    high_res: Tensor
    visual_pe: Tuple[Tensor, Tensor]
    rope_deltas: Tensor
    cache: KVCache
    attention_mask: Tensor


def prefill(w, cfg: OracleConfig, input_ids: Tensor, attention_mask: Tensor, pixel_values: Tensor,
            grid_thw: Tensor, all_hidden: bool = False):
    """forward_main prefill branch, padt.py:183-219,256-301,330-339."""
    low, high, pe = vit_forward(w, cfg, pixel_values, grid_thw)
    proto = prototypes(w, cfg, low)
    rows = cfg.vocab_size + proto.shape[0]
    lmask = logit_mask(cfg, grid_thw, rows)
    x = embed_inputs(w, cfg, input_ids, proto, low)
    pos, deltas = rope_index(cfg, input_ids, grid_thw, attention_mask)
    cache = KVCache(cfg.num_layers)
    out = llm_forward(w, cfg, x, pos, attention_mask, cache, all_hidden)
    hidden = out[0] if all_hidden else out
    logits = vrt_logits(w, cfg, hidden, proto, lmask)
    st = PrefillState(proto, lmask, high, pe, deltas, cache, attention_mask)
    return logits, out, st


def decode_step(w, cfg: OracleConfig, st: PrefillState, input_ids: Tensor, all_hidden: bool = False):
    """forward_main decode branch, padt.py:221-229,268-277,279-301. input_ids (B,1) global VRT ids."""
    B = input_ids.shape[0]
    x = embed_inputs(w, cfg, input_ids, st.proto, None)
    cache_pos = st.cache.seq_len()
    delta = cache_pos + st.rope_deltas                                  # (B,1)
    pos = (torch.arange(1).view(1, -1).expand(B, -1) + delta).unsqueeze(0).expand(3, -1, -1)
    st.attention_mask = torch.cat([st.attention_mask, st.attention_mask.new_ones(B, 1)], dim=1)
    out = llm_forward(w, cfg, x, pos, st.attention_mask, st.cache, all_hidden)
    hidden = out[0] if all_hidden else out
    return vrt_logits(w, cfg, hidden, st.proto, st.lmask), out


def generate(w, cfg: OracleConfig, input_ids: Tensor, attention_mask: Tensor, pixel_values: Tensor,
             grid_thw: Tensor, max_new_tokens: int, schedule: Optional[Sequence[str]] = None,
             collect_logits: bool = False, force_tokens: Optional[Tensor] = None, repetition_penalty: float = 1.0,
             eos_token_ids: Optional[Sequence[int]] = None):
    """Greedy loop, padt.py:670-762 (essentials).

    ``schedule`` (synthetic-weights only): per step one of 't' (argmax restricted to text rows),
    'v' (restricted to the sample's VRT rows), 'e' (force EOS), None/'f' (free).  Implemented as an additive
    logits processor, i.e. where HF's ``logits_processor`` sits (padt.py:717).

    Returns dict(sequences (B,L+T), hidden (list over steps of last-layer (B,Lq_t,D)), state, logits).
    """
    B, L = input_ids.shape
    seq = input_ids.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    hiddens, all_logits = [], []
    st = None
    for t in range(max_new_tokens):
        if t == 0:
            logits, hidden, st = prefill(w, cfg, seq, attention_mask, pixel_values, grid_thw)
        else:
            logits, hidden = decode_step(w, cfg, st, seq[:, -1:])
        nl = logits[:, -1, :].clone().float()
        if repetition_penalty != 1.0:
            # HF RepetitionPenaltyLogitsProcessor (generation/logits_process.py), first in the processor list (padt.py:570-580,
            # applied at :717): every id already in the row — prompt, padding and generated tokens — is penalised
            score = torch.gather(nl, 1, seq)
            score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
            nl = nl.scatter(1, seq, score)
        mode = schedule[t] if schedule is not None and t < len(schedule) else None
        if mode == 't':
            nl[:, cfg.vocab_size:] = float("-inf")
        elif mode == 'v':
            nl[:, :cfg.vocab_size] = float("-inf")
        elif mode == 'e':
            nl[:] = float("-inf")
            nl[:, cfg.eos_token_id] = 0.0
        hiddens.append(hidden)
        if collect_logits:
            all_logits.append(nl)
        nxt = torch.argmax(nl, dim=-1)
        if force_tokens is not None and t < force_tokens.shape[1]:
            nxt = force_tokens[:, t].clone()                 # teacher forcing (tests: margin rule of SURVEY.md §7)
        nxt = nxt * unfinished + cfg.pad_token_id * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], dim=-1)
        eos_set = [cfg.eos_token_id] if eos_token_ids is None else list(eos_token_ids)
        unfinished = unfinished & (~torch.isin(nxt, torch.tensor(eos_set))).long()    # padt.py:756 (EosTokenCriteria over the id list)
        if int(unfinished.max()) == 0:
            break
    return {"sequences": seq, "hidden": hiddens, "state": st, "logits": all_logits}


def warp_logits(scores: Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, min_tokens_to_keep: int = 1) -> Tensor:
    """HF logits warpers in the order _get_logits_processor appends them (transformers 4.50 generation/utils.py, called at padt.py:570-580;
    classes in generation/logits_process.py): TemperatureLogitsWarper (scores / T), TopKLogitsWarper (-inf below the k-th largest: ties
    with it stay), TopPLogitsWarper (sort ascending, drop the tokens whose cumulative probability is <= 1 - top_p, keep the last
    min_tokens_to_keep).  The sampling branch then draws multinomial(softmax(result)) (padt.py:740-743)."""
    s = scores.float() / temperature
    if top_k and top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), s.shape[-1])
        kth = torch.topk(s, k)[0][..., -1, None]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p < 1.0:
        sorted_logits, sorted_indices = torch.sort(s, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = 0
        s = s.masked_fill(remove.scatter(-1, sorted_indices, remove), float("-inf"))
    return s


# --------------------------------------------------------------------------- PaDT decoder
def _apply_rotary_half(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """flash-attn non-interleaved rotary (padt_decoder.py:43,50): rot dim = 2*cos.shape[-1]; pairs (x_i, x_{i+rot/2})."""
    ro = cos.shape[-1] * 2
    x1, x2 = x[..., : ro // 2], x[..., ro // 2: ro]
    c, s = cos.unsqueeze(-2), sin.unsqueeze(-2)
    return torch.cat([x1 * c - x2 * s, x1 * s + x2 * c, x[..., ro:]], dim=-1)


def dec_attention(w, pfx: str, heads: int, query: Tensor, key: Tensor, cu_q, cu_k, q_pos, k_pos,
                  rotary: Tuple[bool, bool]) -> Tensor:
    """PaDTDecoderFlashAttention2.forward, padt_decoder.py:20-60."""
    q = linear(query if rotary[0] else query + q_pos, w[pfx + "q_proj.weight"], w[pfx + "q_proj.bias"])
    k = linear(key if rotary[1] else key + k_pos, w[pfx + "k_proj.weight"], w[pfx + "k_proj.bias"])
    v = linear(key, w[pfx + "v_proj.weight"], w[pfx + "v_proj.bias"])
    q = q.reshape(query.shape[0], heads, -1)
    k = k.reshape(key.shape[0], heads, -1)
    v = v.reshape(key.shape[0], heads, -1)
    if rotary[0]:
        cos, sin = q_pos
        q = _apply_rotary_half(q.float(), cos.chunk(2, -1)[0].float(), sin.chunk(2, -1)[0].float()).type_as(q)
    if rotary[1]:
        cos, sin = k_pos
        k = _apply_rotary_half(k.float(), cos.chunk(2, -1)[0].float(), sin.chunk(2, -1)[0].float()).type_as(k)
    a = varlen_attention(q, k, v, cu_q, cu_k, causal=False).reshape(query.shape[0], -1)
    return linear(a, w[pfx + "proj.weight"], w[pfx + "proj.bias"])


def dec_block(w, pfx: str, heads: int, query, memory, cu_q, cu_m, query_pos, memory_pos):
    """PaDTDecoderBlock.forward, padt_decoder.py:95-128 (update_memory=True for all three blocks)."""
    qn = rms_norm(query, w[pfx + "norm1.weight"])
    query = query + dec_attention(w, pfx + "self_attn.", heads, qn, qn, cu_q, cu_q, query_pos, query_pos, (False, False))
    qn = rms_norm(query, w[pfx + "norm2.weight"])
    mn = rms_norm(memory, w[pfx + "norm3.weight"])
    query = query + dec_attention(w, pfx + "cross_attn_query_to_image.", heads, qn, mn, cu_q, cu_m, query_pos, memory_pos, (False, True))
    n4 = rms_norm(query, w[pfx + "norm4.weight"])
    query = query + linear(F.gelu(linear(n4, w[pfx + "mlp.0.weight"], w[pfx + "mlp.0.bias"])),
                           w[pfx + "mlp.2.weight"], w[pfx + "mlp.2.bias"])
    qn = rms_norm(query, w[pfx + "norm5.weight"])
    mn = rms_norm(memory, w[pfx + "norm6.weight"])
    memory = memory + dec_attention(w, pfx + "cross_attn_image_to_query.", heads, mn, qn, cu_m, cu_q, memory_pos, query_pos, (True, False))
    return query, memory


def _mlp3(w, pfx, x):
    x = F.gelu(linear(x, w[pfx + "0.weight"], w[pfx + "0.bias"]))
    x = F.gelu(linear(x, w[pfx + "2.weight"], w[pfx + "2.bias"]))
    return linear(x, w[pfx + "4.weight"], w[pfx + "4.bias"])


def padt_decoder(w, cfg: OracleConfig, object_vp_feat: List[Tensor], cu_low: Tensor, cu_high: Tensor,
                 visual_pe: Tuple[Tensor, Tensor], cu_patch: Tensor, obj_grids: Tensor):
    """PaDTDecoder.forward, padt_decoder.py:187-276."""
    p = "vl_decoder."
    heads = cfg.dec_heads
    mu = cfg.merge_unit
    n_vp = [f.shape[0] for f in object_vp_feat]
    n_obj = len(object_vp_feat)

    def in_proj(x):
        x = rms_norm(x, w[p + "input_projection.0.weight"])
        x = F.gelu(linear(x, w[p + "input_projection.1.weight"], w[p + "input_projection.1.bias"]))
        return linear(x, w[p + "input_projection.3.weight"], w[p + "input_projection.3.bias"])

    feats = in_proj(torch.cat(object_vp_feat))
    cu_query, acc = [], 0
    for n in n_vp:
        cu_query.append(w[p + "bbox_score_mask_tokens.weight"])
        cu_query.append(feats[acc:acc + n] + w[p + "vp_embedding.weight"])
        acc += n
    cu_query = torch.cat(cu_query, dim=0)
    cu_q = [0]
    for n in n_vp:
        cu_q.append(cu_q[-1] + 3 + n)
    cu_p = cu_patch.tolist()
    cu_l = [c // mu for c in cu_p]
    low = in_proj(cu_low)
    D = visual_pe[0].shape[-1]
    low_pe = (visual_pe[0].reshape(-1, mu, D)[:, 0, :], visual_pe[1].reshape(-1, mu, D)[:, 0, :])

    out, low = dec_block(w, p + "low_res_transformer.", heads, cu_query, low, cu_q, cu_l, cu_query, low_pe)
    high = rms_norm(low.unsqueeze(1).repeat_interleave(mu, dim=1).flatten(0, 1) + cu_high, w[p + "high_res_norm.weight"])
    out, high = dec_block(w, p + "high_res_transformer1.", heads, out, high, cu_q, cu_p, cu_query, visual_pe)
    out, high = dec_block(w, p + "high_res_transformer2.", heads, out, high, cu_q, cu_p, cu_query, visual_pe)

    tok = torch.stack([out[cu_q[i]: cu_q[i] + 3] for i in range(n_obj)])          # (n_obj,3,D)
    bbox = torch.sigmoid(_mlp3(w, p + "bbox_prediction.", tok[:, 0]))
    score = linear(tok[:, 1], w[p + "score_prediction.weight"], w[p + "score_prediction.bias"])
    Hs = torch.tensor([int(g[1]) for g in obj_grids], dtype=torch.int64)
    Ws = torch.tensor([int(g[2]) for g in obj_grids], dtype=torch.int64)
    if not cfg.use_mask_loss:
        return bbox, score, None, ()
    mask_tok = _mlp3(w, p + "mask_output_mlp.", tok[:, 2])                         # (n_obj, D/16)

    N, Dd = high.shape
    up1 = linear(high, w[p + "mask_output_upscaling1.0.weight"], w[p + "mask_output_upscaling1.0.bias"])
    up1 = F.gelu(rms_norm(up1, w[p + "mask_output_upscaling1.1.weight"]))
    e = up1.reshape(N, 2, 2, Dd // 4).permute(1, 2, 0, 3)
    e = F.gelu(linear(e, w[p + "mask_output_upscaling2.0.weight"], w[p + "mask_output_upscaling2.0.bias"]))
    e = e.reshape(2, 2, N, 2, 2, Dd // 16).permute(0, 3, 1, 4, 2, 5).flatten(0, 1).flatten(1, 2)   # 4,4,N,D/16
    per_patch = e.permute(2, 0, 1, 3).contiguous()                                    # N,4,4,D/16

    pn = cu_patch[1:] - cu_patch[:-1]
    obj_of = torch.repeat_interleave(torch.arange(n_obj), pn.long())
    pos_in = torch.arange(int(cu_patch[-1])) - cu_patch[:-1].long()[obj_of]
    Wp = Ws[obj_of]
    row, col = pos_in // Wp, pos_in % Wp
    logit = (per_patch * mask_tok.index_select(0, obj_of)[:, None, None, :]).sum(-1)   # N,4,4
    Hm, Wm = int(Hs.max()), int(Ws.max())
    padded = torch.zeros((n_obj, 4, 4, Hm, Wm), dtype=logit.dtype)
    padded[obj_of, :, :, row, col] = logit
    masks = padded.permute(0, 3, 1, 4, 2).contiguous().reshape(n_obj, Hm * 4, Wm * 4)
    return bbox, score, masks, (Hs, Ws)


def vl_decode(w, cfg: OracleConfig, object_vp_feats: List[List[Tensor]], low_res: Tensor, high_res: Tensor,
              grid_thws: Tensor, visual_pes: Tuple[Tensor, Tensor]):
    """padt.py:342-412 (the empty-input dummy pass of 383-393 is skipped; outputs of 406-412 returned)."""
    flat = sum(object_vp_feats, [])
    dt = low_res.dtype
    if len(flat) == 0:
        return {"pred_boxes": torch.zeros((0, 4), dtype=dt), "pred_score": torch.zeros((0, 1), dtype=dt),
                "pred_mask": torch.zeros((0, 8, 8), dtype=dt), "pred_mask_valid_hw": (), "sample_idx": []}
    off = 0
    sidx, lows, highs, pc, ps, cu, grids = [], [], [], [], [], [], []
    for si, (feats, g) in enumerate(zip(object_vp_feats, grid_thws)):
        n = int(g[0] * g[1] * g[2])
        k = len(feats)
        lo = low_res[off // 4: (off + n) // 4]
        hi = high_res[off: off + n]
        sidx.extend([si] * k)
        lows.append(lo.unsqueeze(0).repeat_interleave(k, 0).flatten(0, 1))
        highs.append(hi.unsqueeze(0).repeat_interleave(k, 0).flatten(0, 1))
        pc.append(visual_pes[0][off: off + n].unsqueeze(0).repeat_interleave(k, 0).flatten(0, 1))
        ps.append(visual_pes[1][off: off + n].unsqueeze(0).repeat_interleave(k, 0).flatten(0, 1))
        cu.extend([n] * k)
        grids.extend([g] * k)
        off += n
    cu_patch = F.pad(torch.tensor(cu, dtype=torch.float32).cumsum(0), (1, 0)).to(torch.int32)
    bbox, score, masks, hw = padt_decoder(w, cfg, flat, torch.cat(lows), torch.cat(highs),
                                          (torch.cat(pc), torch.cat(ps)), cu_patch, torch.stack(grids))
    return {"pred_boxes": bbox, "pred_score": score, "pred_mask": masks, "pred_mask_valid_hw": hw, "sample_idx": sidx}


# --------------------------------------------------------------------------- synthetic weights
# ---------------------------------------------------------------------------------------------------------------------
# Caller-side post-processing: eval/evaluation_scripts/utils.py:252-266 (same expressions at
# eval/test_demo.py:145-161 with the resized image size).  The reference writes these inline in its eval loop; restated
# here expression by expression.  RLE: pycocotools (setup.py:31, unpinned, not in the container) — `cocomask.encode` of a
# Fortran-ordered uint8 mask = run lengths in column-major order starting with the zero run; the compressed `counts`
# string follows the published COCO maskApi `rleToString` (parity unpinned for the string form; the run lengths are exact).
def mask_rle_counts(mask_u8) -> List[int]:
    import numpy as np
    flat = np.asarray(mask_u8, dtype=np.uint8).flatten(order="F")
    if flat.size == 0:
        return []
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(bounds).tolist()
    return runs if flat[0] == 0 else [0] + runs


def rle_counts_to_string(counts: Sequence[int]) -> str:
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def box_iou_xywh(b1: Sequence[float], b2: Sequence[float]) -> float:
    """eval/evaluation_scripts/eval_refcoco.py:15-41 (calculate_iou on (x, y, w, h) boxes) — the "box IoU vs ref" of the metric."""
    x1, y1, w1, h1 = b1
    x2, y2, w2, h2 = b2
    iw = max(0, min(x1 + w1, x2 + w2) - max(x1, x2))
    ih = max(0, min(y1 + h1, y2 + h2) - max(y1, y2))
    inter = iw * ih
    union = w1 * h1 + w2 * h2 - inter
    return 0.0 if union == 0 else inter / union


def mask_ciou_parts(pred, gt) -> Tuple[int, int]:
    """eval_refcoco.py:44-47 (calculate_ciou): intersection and union pixel counts of two binary masks."""
    import numpy as np
    return int(np.logical_and(pred, gt).sum()), int(np.logical_or(pred, gt).sum())


def postprocess_results(decoded: Dict, labels: List[List[str]], image_sizes: Sequence[Tuple[int, int]]) -> List[Dict]:
    """utils.py:252-266: per object → score = sigmoid(logit); bbox = cxcywh → clamped xywh, scaled by the image's (w, h) and
    rounded with Python's round(); mask = bilinear up-sample of the valid 4H x 4W logits to (h, w), sigmoid > 0.5, uint8."""
    res = []
    if decoded["pred_boxes"].shape[0] == 0:
        return res
    hw = torch.stack([decoded["pred_mask_valid_hw"][0], decoded["pred_mask_valid_hw"][1]], dim=-1)
    flat_labels = sum(labels, [])
    for box, score, label, mask, mask_hw, si in zip(decoded["pred_boxes"], decoded["pred_score"].sigmoid(), flat_labels,
                                                    decoded["pred_mask"], hw, decoded["sample_idx"]):
        eval_box = (max(box[0].item() - box[2].item() / 2, 0), max(box[1].item() - box[3].item() / 2, 0),
                    min(box[2].item(), 1), min(box[3].item(), 1))
        w, h = image_sizes[si]
        eval_box = (round(eval_box[0] * w), round(eval_box[1] * h), round(eval_box[2] * w), round(eval_box[3] * h))
        up = F.interpolate(mask[None, None, :mask_hw[0] * 4, :mask_hw[1] * 4], size=(h, w), mode="bilinear")[0, 0]
        m = (up.sigmoid() > 0.5).cpu().numpy().astype("uint8")
        counts = mask_rle_counts(m)
        res.append({"sample_idx": int(si), "score": score.item(), "category": label, "bbox": eval_box, "mask": m,
                    "mask_logits_up": up, "rle": {"size": [h, w], "counts": rle_counts_to_string(counts)}, "rle_counts": counts})
    return res


# ---------------------------------------------------------------------------------------------------------------------
# Image front-end: the HF Qwen2-VL image processor the reference calls through AutoProcessor
# (transformers/models/qwen2_vl/image_processing_pil_qwen2_vl.py: smart_resize, _preprocess, patchify; arithmetic of
# image_transforms.rescale / normalize).  Pinned against the INSTALLED transformers 5.15 PIL processor
# (tests/golden/make_golden_pre.py); the reference pins 4.50.0, whose processor is not in the container (parity with 4.50
# unpinned).  The resize is Pillow's ImagingResample, restated below (pil_resample) and pinned against PIL.Image.resize itself.
def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


IMAGE_MEAN = (0.48145466, 0.4578275, 0.40821073)
IMAGE_STD = (0.26862954, 0.26130258, 0.27577711)
RESCALE = 0.00392156862745098


def patchify_normalize(image_u8, patch: int = 14, merge: int = 2, temporal: int = 2):
    """(H, W, 3) uint8, already resized to multiples of patch*merge → ((H/14)*(W/14), 3*2*14*14) float32 rows in
    (h/2, w/2, 2, 2) block-major patch order, + (grid_h, grid_w).  rescale: float32(float64(u8) * 1/255); normalize:
    (x - float32(mean)) / float32(std) in float32; frame duplicated along the temporal axis."""
    import numpy as np
    x = np.asarray(image_u8).transpose(2, 0, 1)
    r = (x.astype(np.float64) * RESCALE).astype(np.float32)
    mean = np.array(IMAGE_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(IMAGE_STD, dtype=np.float32)[:, None, None]
    v = (r - mean) / std
    C, H, W = v.shape
    gh, gw = H // patch, W // patch
    pt = v.reshape(C, gh // merge, merge, patch, gw // merge, merge, patch).transpose(1, 4, 2, 5, 0, 3, 6)
    pt = np.broadcast_to(pt[:, :, :, :, :, None, :, :], (*pt.shape[:5], temporal, *pt.shape[5:]))
    return pt.reshape(gh * gw, C * temporal * patch * patch), gh, gw


def pil_resample(image_u8, out_w: int, out_h: int, filter_name: str = "bicubic"):
    """Pillow's 8-bit ImagingResample (src/libImaging/Resample.c; Pillow is a dependency of the reference through
    transformers' image processor and eval/test_demo.py:73 `image.resize(..., Image.Resampling.LANCZOS)`; unpinned in setup.py) restated
    loop for loop: precompute_coeffs (double taps of the scaled filter, summed left to right, normalised), normalize_coeffs_8bpc
    (22-bit fixed point, truncation after +-0.5), horizontal pass then vertical pass over uint8 with
    clip8(((1 << 21) + sum) >> 22).  Pinned: byte-identical to PIL.Image.resize of the installed Pillow (tests/test_preprocess_cpu.py)."""
    import numpy as np
    PB = 32 - 8 - 2

    def bicubic(x):
        a = -0.5
        x = abs(x)
        if x < 1.0:
            return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
        if x < 2.0:
            return (((x - 5) * x + 8) * x - 4) * a
        return 0.0

    def sinc(x):
        if x == 0.0:
            return 1.0
        x = x * math.pi
        return math.sin(x) / x

    def lanczos(x):
        return sinc(x) * sinc(x / 3) if -3.0 <= x < 3.0 else 0.0
    f, sup = {"bicubic": (bicubic, 2.0), "lanczos": (lanczos, 3.0)}[filter_name]

    def coeffs(in_size, out_size):
        scale = filterscale = in_size / out_size
        if filterscale < 1.0:
            filterscale = 1.0
        support = sup * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        kk = np.zeros((out_size, ksize), np.int64)
        bounds = np.zeros((out_size, 2), np.int64)
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            ss = 1.0 / filterscale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), in_size) - xmin
            k = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
            ww = 0.0
            for v in k:
                ww += v
            if ww != 0.0:
                k = [v / ww for v in k]
            for x, v in enumerate(k):
                kk[xx, x] = int(-0.5 + v * (1 << PB)) if v < 0 else int(0.5 + v * (1 << PB))
            bounds[xx] = (xmin, xmax)
        return bounds, kk
    cur = np.asarray(image_u8)
    H, W, C = cur.shape
    if out_w != W:
        b, kk = coeffs(W, out_w)
        out = np.zeros((H, out_w, C), np.uint8)
        for x in range(out_w):
            lo, n = b[x]
            acc = (1 << (PB - 1)) + (cur[:, lo:lo + n, :].astype(np.int64) * kk[x, :n][None, :, None]).sum(1)
            out[:, x, :] = np.clip(acc >> PB, 0, 255)
        cur = out
    if out_h != H:
        b, kk = coeffs(H, out_h)
        out = np.zeros((out_h, cur.shape[1], C), np.uint8)
        for y in range(out_h):
            lo, n = b[y]
            acc = (1 << (PB - 1)) + (cur[lo:lo + n].astype(np.int64) * kk[y, :n][:, None, None]).sum(0)
            out[y] = np.clip(acc >> PB, 0, 255)
        cur = out
    return cur


def weight_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """Checkpoint key -> shape for the whole path (HF-4.50 layout + PaDT extras)."""
    s: Dict[str, Tuple[int, ...]] = {}
    vh, vi = cfg.vit_hidden, cfg.vit_intermediate
    s["visual.patch_embed.proj.weight"] = (vh, cfg.in_channels, cfg.temporal_patch_size, cfg.patch_size, cfg.patch_size)
    for i in range(cfg.vit_depth):
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
    D, I = cfg.hidden_size, cfg.intermediate_size
    s["model.embed_tokens.weight"] = (cfg.vocab_size, D)
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (D,)
        s[p + "post_attention_layernorm.weight"] = (D,)
        s[p + "self_attn.q_proj.weight"] = (cfg.num_heads * cfg.head_dim, D)
        s[p + "self_attn.q_proj.bias"] = (cfg.num_heads * cfg.head_dim,)
        for n_ in ("k_proj", "v_proj"):
            s[p + f"self_attn.{n_}.weight"] = (cfg.num_kv_heads * cfg.head_dim, D)
            s[p + f"self_attn.{n_}.bias"] = (cfg.num_kv_heads * cfg.head_dim,)
        s[p + "self_attn.o_proj.weight"] = (D, cfg.num_heads * cfg.head_dim)
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
    dh, di = cfg.dec_hidden, cfg.dec_intermediate
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


def _is_norm_weight(name: str) -> bool:
    return (name.endswith("norm.weight") or "layernorm.weight" in name or ".ln_q.weight" in name
            or any(name.endswith(f"norm{k}.weight") for k in range(1, 7))
            or name.endswith("input_projection.0.weight") or name.endswith("mask_output_upscaling1.1.weight")
            or name == "vis_norm.weight")


def synthetic_weights(cfg: OracleConfig, seed: int = 0, std: float = 0.02, dtype=torch.float32,
                      bias_std: float = 0.0, norm_jitter: float = 0.0) -> Dict[str, Tensor]:
    """Seeded N(0,std^2) matrices, norm weights 1 (+N(0,norm_jitter^2)), biases N(0,bias_std^2) (0 by default)
    — SURVEY.md §8d.

    Every tensor has its own generator seeded from (seed, key) so any subset can be regenerated
    independently (the HIP side uses the same rule to build identical weights on the GPU box).
    """
    import zlib
    out = {}
    for name, shape in weight_shapes(cfg).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        if _is_norm_weight(name):
            t = torch.ones(shape)
            if norm_jitter > 0:
                t = t + torch.randn(shape, generator=g) * norm_jitter
            out[name] = t.to(dtype)
            continue
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * bias_std if bias_std > 0 else torch.zeros(shape)
        else:
            t = torch.randn(shape, generator=g) * std
        out[name] = t.to(dtype)
    return out
