"""``precision="reference"`` — the split-precision machinery of the PaDT decoder (csrc/decoder_hp.hip) applied to the 68 upstream layers.

The default path multiplies 16-bit MFMA operands (fp16: 11 mantissa bits) and lands AT the attributed floor of that operand type — boxes
1.3e-4, mask logits 3.7e-3 of their range at full PaDT_Pro_3B depth: the north star's 1e-3 holds for the boxes and not for the mask logits.
This mode meets it on every float output, at 2x the MFMA work:

  * every GEMM A operand of ViT, merger, prototype projection and LLM is a bf16 (hi, lo) pair — hi = bf16(x), lo = bf16(x − hi): 16 mantissa
    bits — against the weight image [W | W] (``padt_gemm_bf16_ex`` at K' = 2K; checkpoints are bf16, so W itself is exact; NO norm folding:
    the normalised rows are produced explicitly by ``padt_norm_split``), fp32 accumulation, fp32 residual streams, fp32 SwiGLU (round 6: the
    gate/up GEMM's own epilogue — exact expf / division on the fp32 accumulators, (hi, lo) pair out — instead of fp32 gate / up rows + a
    ``padt_swiglu_split`` pass), fp32 LayerNorm of the prototypes (``padt_layernorm_f32``);
  * ALL attention in fp32 (``padt_attn_f32``, the decoder's varlen kernel): the ViT's 64-token windows and 2116 x 2116 full layers, the LLM's
    causal GQA prompt pass (kv_group / causal arguments, round 5) and its decode steps over an fp32 [K | V] cache per layer (one row per cached
    token, samples at a fixed stride; ``len_k`` = the valid keys; the step's new rows land by ``padt_scatter_rows_f32``); rotary in fp32
    (``padt_rope_half_f32`` on tables from ``padt_rope_table``).  (The first version of this mode kept the LLM's attention on the fp16 MFMA kernels —
    tests/studies/reference_mode_floor.py: 3.4e-4 on the 3B's mask logits, measured 3.3e-4 — which left PaDT_Pro_7B at 9.8e-4, inside 1e-3 with 2 % to
    spare; with fp32 attention nothing of the LLM is rounded below 16 mantissa bits.)  The logit head stays the fp16 ``padt_vrt_head`` (it selects
    tokens; no float output depends on it);
  * token / image / prototype embeddings are gathered from ONE fp32 table [E ‖ prototypes ‖ image rows] (no 16-bit rounding on the way in),
    the per-step hidden rows are kept in fp32 for ``parseVRTintoCompletion`` → ``vl_decode``.

Round 6: attention runs on the f32-input MFMA (``padt_attn_f32_mfma``), decode steps are captured hipGraphs over static session buffers, and
their projections read every weight ONCE (``padt_gemm_split_rows``: a weight fragment multiplies the hi and the lo fragment of a row block; the
K' = 2K form streamed the doubled image).  bench.py prints this mode's rate next to the headline (``reference_precision``).  Everything below is
kernel sequencing — no arithmetic in PyTorch.
"""
from typing import Dict

import torch

from . import ops
from .config import PaDTConfig
from .weights import _pad_cols, _pad_rows, _pad_to, interleave16

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


def _dbl(t: torch.Tensor) -> torch.Tensor:
    """[W | W] along K: the image a (hi | lo) row multiplies."""
    return torch.cat([t, t], dim=1).contiguous()


def prepare_reference_weights(sd: Dict[str, torch.Tensor], cfg: PaDTConfig, device) -> dict:
    """Checkpoint → bf16 weight images of the reference-precision path: plain nn.Linear matrices doubled along K, nothing folded, gate / up rows
    interleaved in 16-row blocks [gate16 | up16 | ...] (round 6: the SwiGLU is the gate/up GEMM's epilogue — exact expf / division on the fp32
    accumulators, (hi, lo) pair out; the fp32 gate / up rows of round 5 are never written), MLP intermediates zero-padded to a multiple of 64."""
    dev = torch.device(device)
    R = {}

    def g(name):
        if name not in sd:
            raise KeyError(f"checkpoint is missing '{name}'")
        return sd[name].to(device=dev, dtype=BF16)
    v = cfg.vision_config
    vi_pad = _pad_to(v.intermediate_size, 64)
    R["vi_pad"] = vi_pad
    R["vit.patch.hp"] = _dbl(g("visual.patch_embed.proj.weight").reshape(v.hidden_size, -1))
    for i in range(v.depth):
        s, d = f"visual.blocks.{i}.", f"vit.{i}."
        R[d + "norm1"], R[d + "norm2"] = g(s + "norm1.weight").contiguous(), g(s + "norm2.weight").contiguous()
        R[d + "qkv.hp"], R[d + "qkv.b"] = _dbl(g(s + "attn.qkv.weight")), g(s + "attn.qkv.bias").contiguous()
        R[d + "proj.hp"], R[d + "proj.b"] = _dbl(g(s + "attn.proj.weight")), g(s + "attn.proj.bias").contiguous()
        R[d + "gu.hp"] = _dbl(interleave16(_pad_rows(g(s + "mlp.gate_proj.weight"), vi_pad), _pad_rows(g(s + "mlp.up_proj.weight"), vi_pad)))
        R[d + "gu.b"] = interleave16(_pad_rows(g(s + "mlp.gate_proj.bias"), vi_pad), _pad_rows(g(s + "mlp.up_proj.bias"), vi_pad)).contiguous()
        R[d + "down.hp"], R[d + "down.b"] = _dbl(_pad_cols(g(s + "mlp.down_proj.weight"), vi_pad)), g(s + "mlp.down_proj.bias").contiguous()
    R["vit.ln_q"] = g("visual.merger.ln_q.weight").contiguous()
    mu, vh = cfg.merge_unit, v.hidden_size
    m0 = g("visual.merger.mlp.0.weight")                              # (mu*vh, mu*vh): its input row is mu ViT rows side by side,
    R["vit.m0.hp"] = torch.stack([m0.view(-1, mu, vh)] * 2, dim=2).reshape(m0.shape[0], 2 * mu * vh).contiguous()   # each as its own [hi | lo] pair
    R["vit.m0.b"] = g("visual.merger.mlp.0.bias").contiguous()
    R["vit.m2.hp"], R["vit.m2.b"] = _dbl(g("visual.merger.mlp.2.weight")), g("visual.merger.mlp.2.bias").contiguous()
    li_pad = _pad_to(cfg.intermediate_size, 64)
    R["li_pad"] = li_pad
    for i in range(cfg.num_hidden_layers):
        s, d = f"model.layers.{i}.", f"llm.{i}."
        R[d + "in_norm"], R[d + "post_norm"] = g(s + "input_layernorm.weight").contiguous(), g(s + "post_attention_layernorm.weight").contiguous()
        R[d + "qkv.hp"] = _dbl(torch.cat([g(s + "self_attn.q_proj.weight"), g(s + "self_attn.k_proj.weight"), g(s + "self_attn.v_proj.weight")], 0))
        R[d + "qkv.b"] = torch.cat([g(s + "self_attn.q_proj.bias"), g(s + "self_attn.k_proj.bias"), g(s + "self_attn.v_proj.bias")], 0).contiguous()
        R[d + "o.hp"] = _dbl(g(s + "self_attn.o_proj.weight"))
        R[d + "gu.hp"] = _dbl(interleave16(_pad_rows(g(s + "mlp.gate_proj.weight"), li_pad), _pad_rows(g(s + "mlp.up_proj.weight"), li_pad)))
        R[d + "down.hp"] = _dbl(_pad_cols(g(s + "mlp.down_proj.weight"), li_pad))
        # decode steps (<= 64 rows): the fragment-packed image of each projection (the K' = 2K images above are the prompt pass's) — 1-KiB wave loads
        for nm in ("qkv", "o", "gu", "down"):
            w2 = R[d + nm + ".hp"]
            R[d + nm + ".pk"] = ops.pack_weight(w2[:, : w2.shape[1] // 2].contiguous())
    R["llm.norm"] = g("model.norm.weight").contiguous()
    R["llm.embed32"] = sd["model.embed_tokens.weight"].to(device=dev, dtype=F32).contiguous()
    if cfg.use_visual_prototype_projection:
        R["proto.norm.w"], R["proto.norm.b"] = g("vis_norm.weight").contiguous(), g("vis_norm.bias").contiguous()
        R["proto.0.hp"], R["proto.1.hp"] = _dbl(g("vis_proj.0.weight")), _dbl(g("vis_proj.1.weight"))
    return R


class ReferencePath:
    """ViT, prototypes, prompt pass and decode step of a ``precision="reference"`` model (modeling.PaDTForConditionalGeneration routes its
    generate_launch / decode steps here; sessions, greedy bookkeeping, VRT head and parse / vl_decode are the default path's)."""

    def __init__(self, cfg: PaDTConfig, state_dict, model):
        self.cfg, self.m, self.device = cfg, model, model.device
        assert model.dtype == F16, "the reference-precision path runs its LLM attention on the fp16 instantiation"
        self.R = prepare_reference_weights(state_dict, cfg, self.device)

    # ------------------------------------------------------------------ ViT (padt.py:48-106)
    def visual(self, pixel_values, grid_thw, nf=None):
        cfg, R = self.cfg, self.R
        v = cfg.vision_config
        plan = self.m.visual.plan(grid_thw)
        P, vh, H = plan.P, v.hidden_size, v.num_heads
        hd = vh // H
        if pixel_values.shape[0] != P:
            raise ValueError(f"pixel_values has {pixel_values.shape[0]} rows, image_grid_thw implies {P}")
        pix = pixel_values.contiguous()
        if pix.dtype != F32:
            pix = ops.cast_x16_f32(pix)
        S, F = ops.OUT_SPLIT, ops.OUT_F32
        ps, _ = ops.norm_split(pix)                                            # pixel rows as (hi, lo) pairs
        x0 = ops.gemm_hp(ps, R["vit.patch.hp"])                              # conv3d-as-GEMM (HF:116-122), fp32 out
        x32 = ops.gather_rows(x0, plan.patch_perm, D=vh)                       # window order
        for i in range(v.depth):
            d = f"vit.{i}."
            full = i in v.fullatt_block_indexes
            cu, mx = (plan.cu_full, plan.max_full) if full else (plan.cu_win, plan.max_win)
            n, _ = ops.norm_split(x32, R[d + "norm1"], eps=1e-6)
            qkv = ops.gemm_hp(n, R[d + "qkv.hp"], R[d + "qkv.b"])             # (P, 3 vh) fp32
            ops.rope_half_f32_(qkv, plan.cos, plan.sin, 2 * H, hd)             # q and k are adjacent in the fused row
            a = ops.attn_f32(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:3 * vh], cu, cu, mx, mx, H, hd, mfma=hd in (80, 128))
            ops.gemm_hp(a, R[d + "proj.hp"], R[d + "proj.b"], out=x32, epilogue=ops.EPI_RESID, residual=x32)
            n, _ = ops.norm_split(x32, R[d + "norm2"], eps=1e-6)
            h = ops.gemm_hp(n, R[d + "gu.hp"], R[d + "gu.b"], epilogue=ops.EPI_SWIGLU, out_mode=S)      # silu(gate) * up as (hi, lo) rows
            ops.gemm_hp(h, R[d + "down.hp"], R[d + "down.b"], out=x32, epilogue=ops.EPI_RESID, residual=x32)
        if nf is not None:
            ops.check_finite(x32, nf)
        n, _ = ops.norm_split(x32, R["vit.ln_q"], eps=1e-6)                    # (P, 2 vh): mu consecutive rows = one merger input row of mu pairs
        mu = cfg.merge_unit
        m = ops.gemm_hp(n.view(plan.N, 2 * vh * mu), R["vit.m0.hp"], R["vit.m0.b"], epilogue=ops.EPI_GELU, out_mode=S)
        low_win = ops.gemm_hp(m, R["vit.m2.hp"], R["vit.m2.b"])
        low = ops.gather_rows(low_win, plan.reverse, D=cfg.hidden_size)        # raster order (padt.py:103-104)
        return low, x32, (plan.cos.clone(), plan.sin.clone())

    # ------------------------------------------------------------------ the fp32 embedding table of a session
    def table(self, sess):
        """[E ‖ prototypes (np_max rows) ‖ image rows (np_max rows: a batch has as many merged image tokens as prototypes)] in fp32, one per
        decode session and owned by it (the captured decode graph holds its address): every embedding the LLM consumes is a row of it."""
        cfg = self.cfg
        t = getattr(sess, "ref_table", None)
        if t is None:
            t = torch.empty((cfg.vocab_size + 2 * sess.np_max, cfg.hidden_size), device=self.device, dtype=F32)
            t[: cfg.vocab_size].copy_(self.R["llm.embed32"])
            sess.ref_table = t
        return t

    def prototypes(self, low32, sess, proto_row0, nf=None):
        """padt.py:187-191 in fp32 / split precision → the session's fp32 table rows AND its fp16 prototype table (head, range checks)."""
        cfg, R = self.cfg, self.R
        n = low32.shape[0]
        tab = self.table(sess)
        dst = tab[cfg.vocab_size + proto_row0: cfg.vocab_size + proto_row0 + n]
        if not cfg.use_visual_prototype_projection:
            dst.copy_(low32)
        else:
            p = ops.layernorm_f32(low32, R["proto.norm.w"], R["proto.norm.b"], eps=1e-5)
            ps, _ = ops.norm_split(p)
            t = ops.gemm_hp(ps, R["proto.0.hp"], out_mode=ops.OUT_SPLIT)
            ops.gemm_hp(t, R["proto.1.hp"], out=dst, epilogue=ops.EPI_RESID, residual=p)
        ops.cast_f32_x16(dst, out=sess.proto[proto_row0: proto_row0 + n])
        if nf is not None:
            ops.check_finite(dst, nf)
        return dst

    # ------------------------------------------------------------------ fp32 KV cache + rotary tables of a session
    def _kv(self, sess):
        """Per layer one fp32 tensor (B * S_max, 2 * Hkv * hd): row b * S_max + s = [K | V] of sample b's token s."""
        cfg = self.cfg
        if getattr(sess, "kv32", None) is None:
            w = 2 * cfg.num_key_value_heads * cfg.head_dim
            sess.kv32 = [torch.zeros((sess.B * sess.s_max, w), device=self.device, dtype=F32) for _ in range(cfg.num_hidden_layers)]
            sess.cu_q1 = torch.arange(sess.B + 1, dtype=torch.int32, device=self.device)
            sess.cu_k32 = (torch.arange(sess.B + 1, dtype=torch.int32, device=self.device) * sess.s_max).contiguous()
            sess.row_base = sess.cu_k32[: sess.B].contiguous()
        return sess.kv32

    def _rope_tables(self, pos3, sess):
        """mRoPE cos / sin [tokens][hd / 2] in fp32 for padt_rope_half_f32 (HF:557-599), from the decode path's own table kernel."""
        cfg = self.cfg
        cs = torch.empty((pos3.shape[1], cfg.head_dim // 2, 2), device=self.device, dtype=F32)
        ops.rope_table(pos3, sess.inv_freq, cs, cfg.head_dim, cfg.mrope_section)
        return cs[..., 0].contiguous(), cs[..., 1].contiguous()          # planar copies (no arithmetic)

    # ------------------------------------------------------------------ one LLM layer on fp32 rows x32 (in place)
    def _layer(self, i, x32, attention):
        cfg, R = self.cfg, self.R
        d = f"llm.{i}."
        few = x32.shape[0] <= 64                                               # a decode step: packed operands (ops.gemm_hp, w_packed)
        pk = (lambda nm: R[d + nm + ".pk"]) if few else (lambda nm: None)
        n, _ = ops.norm_split(x32, R[d + "in_norm"], eps=cfg.rms_norm_eps)
        qkv = ops.gemm_hp(n, R[d + "qkv.hp"], R[d + "qkv.b"], w_packed=pk("qkv"))
        a = attention(qkv)                                                     # fp32 rotary + fp32 attention → (hi, lo) rows
        ops.gemm_hp(a, R[d + "o.hp"], out=x32, epilogue=ops.EPI_RESID, residual=x32, w_packed=pk("o"))
        n, _ = ops.norm_split(x32, R[d + "post_norm"], eps=cfg.rms_norm_eps)
        h = ops.gemm_hp(n, R[d + "gu.hp"], epilogue=ops.EPI_SWIGLU, out_mode=ops.OUT_SPLIT, w_packed=pk("gu"))   # silu(gate) * up as (hi, lo) rows
        ops.gemm_hp(h, R[d + "down.hp"], out=x32, epilogue=ops.EPI_RESID, residual=x32, w_packed=pk("down"))

    def prefill(self, plan, low32, sess, nf=None):
        """Packed prompt pass; fills the session's fp32 KV cache; → post-norm hidden rows of all prompt tokens, fp32 (T, D)."""
        cfg = self.cfg
        Hq, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        n_img = low32.shape[0]
        tab = self.table(sess)
        img0 = cfg.vocab_size + sess.np_max
        assert n_img <= sess.np_max, (n_img, sess.np_max)
        tab[img0: img0 + n_img].copy_(low32)
        ops.embed_tokens(plan.ids, plan.img_index, self.m.W["llm.embed"], sess.proto, None if n_img == 0 else ops.cast_f32_x16(low32, dtype=F16),
                         err_flag=sess.err)                                   # the table-range assert of padt.py:203 (its rows are not used)
        idx = torch.where(plan.img_index >= 0, plan.img_index + img0, plan.ids.to(torch.int32)).to(torch.int32).contiguous()
        x32 = ops.gather_rows(tab, idx, D=cfg.hidden_size)
        kvc = self._kv(sess)
        cos, sin = self._rope_tables(plan.pos3, sess)
        mx = max(plan.lens)
        S, B = sess.s_max, plan.B
        # prompt K | V rows → the cache: cache row (row0 + b) * S + s  <-  packed row cu[b] + min(s, len_b - 1)  (rows past len_b are never read)
        cu = [0]
        for l in plan.lens:
            cu.append(cu[-1] + l)
        src = torch.cat([torch.clamp(torch.arange(S), max=l - 1) + c for l, c in zip(plan.lens, cu[:-1])]).to(torch.int32).to(self.device)
        first_row = plan.first_row
        for i in range(cfg.num_hidden_layers):
            def attention(qkv, i=i):
                ops.rope_half_f32_(qkv, cos, sin, Hq + Hkv, hd)                # q and k heads are adjacent in the fused row
                kv = qkv[:, Hq * hd:]
                ops.gather_rows(kv, src, out=kvc[i][first_row * S: (first_row + B) * S], D=2 * Hkv * hd)
                return ops.attn_f32(qkv[:, : Hq * hd], kv[:, : Hkv * hd], kv[:, Hkv * hd:], plan.cu, plan.cu, mx, mx, Hq, hd,
                                    kv_group=Hq // Hkv, causal=True, mfma=hd in (80, 128))
            self._layer(i, x32, attention)
        hn, _ = ops.norm_split(x32, self.R["llm.norm"], eps=cfg.rms_norm_eps, y0_mode=ops.OUT_F32)
        if nf is not None:
            ops.check_finite(hn, nf)
        return hn

    def step(self, sess):
        """One decode step for every row of the session: the default path's step_kernels with split-precision projections and fp32 attention on
        the f32-input MFMA.  No host-side state and no host-dependent argument: DecodeSession.run_steps captures it into a hipGraph like the
        default step (round 6; round 5 ran it eagerly, 10 launches per layer from Python)."""
        cfg = self.cfg
        Hq, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        B = sess.B
        ops.embed_tokens(sess.cur_tok, None, self.m.W["llm.embed"], sess.proto, None, out=sess.x_rm, err_flag=sess.err)   # range assert only
        tab = self.table(sess)
        x32 = ops.gather_rows(tab, sess.cur_tok.to(torch.int32), D=cfg.hidden_size)
        kvc = self._kv(sess)
        cos, sin = self._rope_tables(sess.pos3, sess)
        where = (sess.row_base + sess.slot).contiguous()                       # cache row of each sample's new token (index arithmetic only)
        for i in range(cfg.num_hidden_layers):
            def attention(qkv, i=i):
                ops.rope_half_f32_(qkv, cos, sin, Hq + Hkv, hd)
                ops.scatter_rows_f32(qkv[:, Hq * hd:], where, kvc[i], D=2 * Hkv * hd)
                return ops.attn_f32(qkv[:, : Hq * hd], kvc[i][:, : Hkv * hd], kvc[i][:, Hkv * hd:], sess.cu_q1, sess.cu_k32, 1, sess.s_max, Hq, hd,
                                    kv_group=Hq // Hkv, len_k=sess.lens, mfma=hd in (80, 128))
            self._layer(i, x32, attention)
        hn32, _ = ops.norm_split(x32, self.R["llm.norm"], eps=cfg.rms_norm_eps, y0_mode=ops.OUT_F32)
        ops.cast_f32_x16(hn32, out=sess.hn)
        ops.check_finite(hn32, sess.nf, rows_per_flag=1, rows=B)
        ops.stash_step_f32(hn32, sess.step, sess.hid32)                        # hid32[*step] = this step's rows (device counter: the step is graph-replayable)
        sess.head_and_select(sess.hn, advance=True)
