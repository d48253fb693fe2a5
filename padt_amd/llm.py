"""LLM side of the hot loop: dynamic VRT embedding, prefill, hipGraph-replayed decode steps, VRT head, greedy loop.

Host work here is integer bookkeeping (packing, position ids, cache slots) and kernel sequencing; all arithmetic runs in
libpadt_hip.so.  Differences from the reference's control flow that do not change results:
  * prompts are PACKED (left-padding is dropped instead of masked) — attention is per-sample varlen, so padded and
    packed runs see identical keys; rope positions follow transformers==4.50 ``get_rope_index`` and are
    padding-independent (position of step t = max prompt position + 1 + t);
  * the [embed_tokens ‖ prototypes] table is never concatenated (padt.py:194,228 rebuild it every forward);
  * only the last prompt position goes through the VRT head (padt.py:294 computes all L positions, :713 keeps one);
  * a decode step is ONE captured hipGraph replayed per token; argmax, EOS/pad bookkeeping, token append and the
    per-step hidden-state stash all happen on device, the host syncs once per `sync_every` steps.
"""
from dataclasses import dataclass
from typing import List, Optional


import torch

from . import ops
from .config import PaDTConfig

I32 = torch.int32


# ------------------------------------------------------------------------------------------------ host integer prep
def rope_index_packed(cfg: PaDTConfig, rows: List[List[int]], grids: List[List[int]]):
    """transformers==4.50.0 ``get_rope_index`` on unpadded token lists (called at padt.py:263).

    Returns (pos [3][T_total] list-of-lists per sample, next_pos per sample = max+1).
    text run: running index on all 3 axes; image: t = start, h = start+row, w = start+col over the merged grid;
    next run starts at previous max + 1.
    """
    m = cfg.vision_config.spatial_merge_size
    out, nxt = [], []
    gi = 0
    for toks in rows:
        chunks: List[torch.Tensor] = []
        st = 0
        n_img = sum(1 for i in range(len(toks) - 1)
                    if toks[i] == cfg.vision_start_token_id and toks[i + 1] == cfg.image_token_id)
        for _ in range(n_img):
            ed = toks.index(cfg.image_token_id, st)
            t, h, w = grids[gi]
            gi += 1
            lt, lh, lw = t, h // m, w // m
            text_len = ed - st
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            chunks.append(torch.arange(text_len).view(1, -1).expand(3, -1) + st_idx)
            ti = torch.zeros(lt * lh * lw, dtype=torch.long)
            hi = torch.arange(lh).view(1, -1, 1).expand(lt, -1, lw).flatten()
            wi = torch.arange(lw).view(1, 1, -1).expand(lt, lh, -1).flatten()
            chunks.append(torch.stack([ti, hi, wi]) + text_len + st_idx)
            st = ed + lt * lh * lw
        if st < len(toks):
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            chunks.append(torch.arange(len(toks) - st).view(1, -1).expand(3, -1) + st_idx)
        pos = torch.cat(chunks, dim=1).reshape(3, -1)
        out.append(pos)
        nxt.append(int(pos.max()) + 1)
    return out, nxt


@dataclass
class PromptPlan:
    B: int
    L_pad: int                       # padded prompt length of the caller's (B, L) input
    lens: List[int]                  # valid tokens per sample
    ids: torch.Tensor                # (T,) int64 packed global ids
    img_index: torch.Tensor          # (T,) int32: index into image_embeds or -1
    pos3: torch.Tensor               # (3, T) int32
    sample: torch.Tensor             # (T,) int32
    slot: torch.Tensor               # (T,) int32
    cu: torch.Tensor                 # (B+1,) int32
    last_idx: torch.Tensor           # (B,) int32 index of each sample's last token
    next_pos: List[int]
    rope_deltas: torch.Tensor        # (B,1) int64, 4.50 convention: max+1 - L_pad
    vrt_off: List[int]               # (B+1) prototype row offsets
    first_row: int = 0               # decode-session row of this batch's first sample (merged decode groups)


def plan_prompt(cfg: PaDTConfig, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor],
                grid_thw: torch.Tensor, device, row0: int = 0, proto_row0: int = 0) -> PromptPlan:
    """``row0`` / ``proto_row0``: first decode-session row and first prototype-table row of this batch when several
    batches share one session (merged decode, modeling.generate_launch): KV rows and VRT ids are shifted accordingly."""
    ids_cpu = input_ids.detach().cpu()
    B, L = ids_cpu.shape
    am = attention_mask.detach().cpu() if attention_mask is not None else torch.ones_like(ids_cpu)
    grids = [[int(x) for x in r] for r in grid_thw.tolist()]
    if len(grids) != B:
        raise ValueError("one image per sample is required (padt.py:301 indexes the logit mask by image)")
    rows = [ids_cpu[b][am[b] == 1].tolist() for b in range(B)]
    n_img_tok = sum(r.count(cfg.image_token_id) for r in rows)
    merged = [g[0] * g[1] * g[2] // cfg.merge_unit for g in grids]
    if n_img_tok != sum(merged):
        raise ValueError(f"Image features and image tokens do not match: tokens: {n_img_tok}, features {sum(merged)}")
    pos, nxt = rope_index_packed(cfg, rows, grids)
    lens = [len(r) for r in rows]
    ids = torch.tensor([t for r in rows for t in r], dtype=torch.int64)
    if proto_row0:
        ids = torch.where(ids >= cfg.vocab_size, ids + proto_row0, ids)
    is_img = ids == cfg.image_token_id
    img_index = torch.where(is_img, torch.cumsum(is_img.to(torch.int64), 0) - 1, torch.full_like(ids, -1)).to(I32)
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    sample = torch.cat([torch.full((l,), b + row0, dtype=I32) for b, l in enumerate(lens)])
    slot = torch.cat([torch.arange(l, dtype=I32) for l in lens])
    off = [0]
    for n in merged:
        off.append(off[-1] + n)
    return PromptPlan(
        B=B, L_pad=L, lens=lens, ids=ids.to(device), img_index=img_index.to(device),
        pos3=torch.cat(pos, dim=1).to(I32).contiguous().to(device), sample=sample.to(device), slot=slot.to(device),
        cu=torch.tensor(cu, dtype=I32, device=device), last_idx=torch.tensor([c - 1 for c in cu[1:]], dtype=I32, device=device),
        next_pos=nxt, rope_deltas=torch.tensor([[n - L] for n in nxt], dtype=torch.int64), vrt_off=off, first_row=row0)


# ------------------------------------------------------------------------------------------------ decode session
MODE = {"f": 0, None: 0, "t": 1, "v": 2, "e": 3}


def _call_hooks(hooks, input_ids, scores, chain: bool):
    """HF's LogitsProcessorList / StoppingCriteriaList are callables over (input_ids, scores); a plain list / tuple of callables is walked here:
    processors chain (each sees its predecessor's scores), criteria are OR-ed."""
    if callable(hooks):
        return hooks(input_ids, scores)
    out = scores if chain else None
    for h in hooks:
        if chain:
            out = h(input_ids, out)
        else:
            r = h(input_ids, scores)
            out = r if out is None else (out | r)
    return out


class DecodeSession:
    """Persistent device state for one (batch, S_max, max prototypes, T_max) shape: KV caches, per-step state, the
    prototype table and the captured decode-step hipGraph.  Pointers are stable across generate() calls so the graph
    is captured once."""

    def __init__(self, cfg: PaDTConfig, W, B: int, s_max: int, np_max: int, t_max: int, device):
        self.cfg, self.W, self.B, self.s_max, self.np_max, self.t_max = cfg, W, B, s_max, np_max, t_max
        D, hd, Hkv, nl = cfg.hidden_size, cfg.head_dim, cfg.num_key_value_heads, cfg.num_hidden_layers
        bf = W.op16                                         # 16-bit operand type of the model (fp16 by default, bf16: weights.prepare_weights)
        z = lambda *s, dt=bf: torch.zeros(*s, device=device, dtype=dt)
        # KV caches, one pair per layer.  head_dim 128 (PaDT_Pro_3B / 7B): the FRAGMENT-PACKED images of padt_decode_attn_rope (K
        # [S/16][D/32][64 lanes][8], V^T [D/16][S/32][64 lanes][8]: 1 KiB contiguous per wave-wide load) read by the one-launch decode
        # attention and written in place by the prompt pass (llm_qkv_post) and the decode step's append; otherwise row-major K / transposed V
        # with the two-launch split attention.  PADT_KV_PACKED=0 keeps the row-major form for A/B runs.
        import os
        self.cache_packed = hd == 128 and cfg.num_attention_heads // Hkv <= 16 and os.environ.get("PADT_KV_PACKED", "1") != "0"
        self.kc = [z(B, Hkv, s_max, hd) for _ in range(nl)]
        self.vtc = [z(B, Hkv, hd, s_max) for _ in range(nl)]
        self.proto = z(np_max, D)
        self.vrt_off = z(B + 1, dt=I32)
        self.mode_table = z(t_max + 1, dt=I32)
        self.step = z(1, dt=I32)
        self.unfinished = z(B, dt=I32)
        self.tokens = z(B, t_max, dt=torch.int64)
        self.cur_tok = z(B, dt=torch.int64)
        self.slot = z(B, dt=I32)
        self.lens = z(B, dt=I32)
        self.pos3 = z(3, B, dt=I32)
        self.hidden_buf = z(t_max, B, D)
        self.nblk = ops.vrt_head_nblk(cfg.vocab_size, np_max)
        self.part_val = z(self.nblk * B, dt=torch.float32)
        self.part_idx = z(self.nblk * B, dt=I32)
        self.attn_ws = ops.new_decode_workspace(B, Hkv, hd, s_max, device)
        # down_proj has only D/16 column blocks (128 for D = 2048): split K over 2 blocks each to occupy every CU
        self.down_split, self.o_split = 2, 1                # (o with split 2: 7.6 → 10.9 us at 64 rows, profiles/r03_decode_experiments.md §7)
        self.splitk_ws = ops.new_splitk_workspace(cfg.hidden_size, max(self.down_split, self.o_split, 1), device)
        half = hd // 2
        self.inv_freq = (1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))).to(device)
        assert self.inv_freq.numel() == half
        # decode-step activations (static addresses → graph-replayable)
        # x / att / h are in the 16-row fragment-packed activation layout (include/padt_hip.h): every projection reads its
        # input fragments as 1 KiB contiguous wave loads; x_rm is the row-major copy at the two ends of the layer loop
        I = W.llm_ipad
        B16 = (B + 15) // 16 * 16
        self.x = z(B16, D)
        self.x_rm = z(B, D)
        self.x32 = z(B, D, dt=torch.float32) if W.resid_f32 else None     # fp32 residual stream of the decode step (x is its packed bf16 mirror)
        self.n = z(B, D)
        self.qkv = z(B, (cfg.num_attention_heads + 2 * Hkv) * hd)
        self.q = z(B, cfg.num_attention_heads * hd)
        self.att = z(B16, cfg.num_attention_heads * hd)
        self.h = z(B16, I)
        self.hn = z(B, D)
        self.hn_first = z(B, D)          # last prompt token's post-norm hidden state per row (first-token selection)
        self.hn_pk = z(B16, D)           # packed copy of the head's input rows
        # generation-config slots (device memory, so the captured graph does not bake them in) + per-row seen-token bitmap
        self.gen_cfg = ops.gen_cfg_tensor(1.0, (), device)
        self.seen = z(B, (cfg.vocab_size + np_max + 31) // 32, dt=I32)
        self.err = z(1, dt=I32)
        self.nf = z(B, dt=I32)           # per row: a decode step produced a non-finite hidden row (fp16 operand overflow; ops.check_finite)
        self.nf_batch = z(B, dt=I32)     # per batch of the decode group (slot k): ViT rows / prototypes / prompt-pass rows not finite
        # generate_collect's ONE read-back per chunk (ops.collect_summary): [err, any unfinished, nf rows, nf batches, first EOS step per row]
        self.summary = z(2 + 3 * B, dt=I32)
        self.summary_host = torch.zeros(2 + 3 * B, dtype=I32).pin_memory() if torch.device(device).type == "cuda" else torch.zeros(2 + 3 * B, dtype=I32)
        self.keep_scores = False         # output_scores=True: every step's masked fp32 logit rows are filed in `scores` [t_max][B][W]
        self.scores = None
        self.rope_cs = z(B, hd // 2, 2, dt=torch.float32)
        self.n_qkv = (cfg.num_attention_heads + 2 * Hkv) * hd
        self.graphs = {}                 # captured decode-step graph per mode (greedy / sampling: different kernel sequences)
        self.do_sample = False
        self.logits = None               # fp32 [B][V + np_max] rows for the sampling kernel, allocated on first use
        self.np_cur = np_max
        self.step_fn = None              # precision="reference": the eager split-precision decode step (reference.ReferencePath.step) instead of step_kernels
        self.hooks = None                # caller-supplied logits_processor / stopping_criteria of this generate (modeling.generate): the HOOKED, eager loop
        self.hid32 = None                # ... and its fp32 per-step hidden rows [t_max][B][D]

    # one decode step, all on the current stream (eager or under capture)
    def step_kernels(self):
        cfg, W = self.cfg, self.W
        Hq, Hkv, hd, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
        B = self.B
        ops.embed_tokens(self.cur_tok, None, W["llm.embed"], self.proto, None, out=self.x_rm, err_flag=self.err)
        f32 = self.x32 is not None
        sc = ops.stream_scale(W.op16) if f32 else 1.0
        eps_n = W.eps_m(cfg.rms_norm_eps) if f32 else cfg.rms_norm_eps       # the fused norms read the (scaled) stream mirror
        if f32:
            ops.cast_x16_f32(self.x_rm, out=self.x32)
            if sc != 1.0:                                                     # first mirror of the step's stream: X(scale * x32)
                ops.cast_f32_x16(self.x32, out=self.x_rm, scale=sc)
        ops.pack_rows(self.x_rm, self.x, B, to_packed=True)
        ops.rope_table(self.pos3, self.inv_freq, self.rope_cs, hd, cfg.mrope_section)
        for i in range(cfg.num_hidden_layers):
            p = f"llm.{i}."
            # 6 launches per layer: [norm+qkv] [rope+append+split attention] [merge] [o+resid] [norm+gate/up+SwiGLU] [down+resid]
            if W.llm_weights == "fp8":                            # fp8 weight images (+ per-row scales): half the bytes per step
                ops.gemm_packed_fp8(self.x, W[p + "qkv.wq"], W[p + "qkv.ws"], self.n_qkv, W[p + "qkv.b"], out=self.qkv,
                                    norm_eps=eps_n, a_packed=True, rows=B)
                ops.decode_attn_rope(self.qkv, self.rope_cs, self.slot, self.kc[i], self.vtc[i], self.att, self.attn_ws, Hq, Hkv,
                                     hd, self.s_max, self.s_max, out_packed=True, cache_packed=self.cache_packed)
                if f32:
                    ops.gemm_packed_resid32(self.att, W[p + "o.wq"], D, self.x32, self.x, scales=W[p + "o.ws"], split_k=self.o_split,
                                            workspace=self.splitk_ws, rows=B)
                else:
                    ops.gemm_packed_fp8(self.att, W[p + "o.wq"], W[p + "o.ws"], D, out=self.x, epilogue=ops.EPI_RESID, residual=self.x,
                                        split_k=self.o_split, workspace=self.splitk_ws, a_packed=True, c_packed=True, rows=B)
                ops.gemm_packed_fp8(self.x, W[p + "gu.wq"], W[p + "gu.ws"], 2 * W.llm_ipad, out=self.h, epilogue=ops.EPI_SWIGLU,
                                    norm_eps=eps_n, a_packed=True, c_packed=True, rows=B)
                if f32:
                    ops.gemm_packed_resid32(self.h, W[p + "down.wq"], D, self.x32, self.x, scales=W[p + "down.ws"], split_k=self.down_split,
                                            workspace=self.splitk_ws, rows=B)
                else:
                    ops.gemm_packed_fp8(self.h, W[p + "down.wq"], W[p + "down.ws"], D, out=self.x, epilogue=ops.EPI_RESID, residual=self.x,
                                        split_k=self.down_split, workspace=self.splitk_ws, a_packed=True, c_packed=True, rows=B)
                continue
            ops.gemm_packed(self.x, W[p + "qkv.wp"], self.n_qkv, W[p + "qkv.b"], out=self.qkv, norm_eps=eps_n,
                            a_packed=True, rows=B)
            ops.decode_attn_rope(self.qkv, self.rope_cs, self.slot, self.kc[i], self.vtc[i], self.att, self.attn_ws, Hq, Hkv,
                                 hd, self.s_max, self.s_max, out_packed=True, cache_packed=self.cache_packed)
            if f32:
                ops.gemm_packed_resid32(self.att, W[p + "o.wp"], D, self.x32, self.x, split_k=self.o_split, workspace=self.splitk_ws, rows=B)
            else:
                ops.gemm_packed(self.att, W[p + "o.wp"], D, out=self.x, epilogue=ops.EPI_RESID, residual=self.x,
                                split_k=self.o_split, workspace=self.splitk_ws, a_packed=True, c_packed=True, rows=B)
            ops.gemm_packed(self.x, W[p + "gu.wp"], 2 * W.llm_ipad, out=self.h, epilogue=ops.EPI_SWIGLU, norm_eps=eps_n,
                            a_packed=True, c_packed=True, rows=B)
            if f32:
                ops.gemm_packed_resid32(self.h, W[p + "down.wp"], D, self.x32, self.x, split_k=self.down_split, workspace=self.splitk_ws, rows=B)
            else:
                ops.gemm_packed(self.h, W[p + "down.wp"], D, out=self.x, epilogue=ops.EPI_RESID, residual=self.x,
                                split_k=self.down_split, workspace=self.splitk_ws, a_packed=True, c_packed=True, rows=B)
        if f32:
            ops.rmsnorm_f32(self.x32, W["llm.norm"], out=self.hn, eps=cfg.rms_norm_eps)
        else:
            ops.pack_rows(self.x, self.x_rm, B, to_packed=False)
            ops.rmsnorm(self.x_rm, W["llm.norm"], out=self.hn, eps=cfg.rms_norm_eps)
        ops.check_finite(self.hn, self.nf, rows_per_flag=1, rows=B)      # sticky per-row flag, read once per generate (modeling.generate_collect)
        self.head_and_select(self.hn, advance=True)

    def head_and_select(self, hn, advance: bool):
        cfg, W = self.cfg, self.W
        hp = W.get("llm.head.wp")
        lg = None
        if self.do_sample or self.keep_scores:               # the sampling kernel / output_scores need the whole masked / penalised logit row
            if self.logits is None:
                self.logits = torch.empty((self.B, (cfg.vocab_size + self.np_max + 3) // 4 * 4), device=hn.device, dtype=torch.float32)
            lg = self.logits
        if hp is not None:                                   # packed table + packed hidden rows: 1 KiB contiguous wave loads
            ops.pack_rows(hn, self.hn_pk, self.B, to_packed=True)
            ops.vrt_head(self.hn_pk, W["llm.head"], self.proto, self.vrt_off, self.part_val, self.part_idx, cfg.eos_token_id,
                         mode_table=self.mode_table, step=self.step, table_packed=hp, rows=self.B, gen_cfg=self.gen_cfg,
                         seen=self.seen, logits=lg)
        else:
            ops.vrt_head(hn, W["llm.head"], self.proto, self.vrt_off, self.part_val, self.part_idx, cfg.eos_token_id,
                         mode_table=self.mode_table, step=self.step, gen_cfg=self.gen_cfg, seen=self.seen, logits=lg)
        hk = self.hooks
        if hk is not None and hk["processors"]:
            # padt.py:717 `next_token_scores = logits_processor(input_ids, next_token_logits)` with the CALLER's processors: they see the rows the head
            # wrote (logit mask + the built-in processors applied) and the sequences so far, and what they return is what is scored / selected / kept
            view = lg[: hk["B"], : hk["table_rows"]]
            new = _call_hooks(hk["processors"], hk["sequences"](), view, chain=True)
            if new is not view:
                view.copy_(new.to(torch.float32))
        if self.keep_scores:                                 # padt.py:719-720: scores += (next_token_scores,) — filed under the device step counter
            if self.scores is None:
                self.scores = torch.zeros((self.t_max,) + tuple(self.logits.shape), device=hn.device, dtype=torch.float32)
            ops.stash_step_f32(lg, self.step, self.scores)
        nblk = self.nblk
        if self.do_sample:                                   # padt.py:740-743: multinomial over softmax of the warped scores
            ops.sample_token(lg, cfg.vocab_size + self.np_max, self.gen_cfg, self.step, self.part_val, self.part_idx, self.B)
            nblk = 1
        elif hk is not None and hk["processors"]:            # the head's fused arg-max partials describe the rows BEFORE the caller's processors
            ops.argmax_rows(lg, cfg.vocab_size + self.np_max, self.part_val, self.part_idx, self.B)
            nblk = 1
        ops.greedy_step(self.part_val, self.part_idx, nblk, hn, self.hidden_buf, self.unfinished, self.tokens,
                        self.cur_tok, self.step, self.slot, self.lens, self.pos3, cfg.eos_token_id, cfg.pad_token_id,
                        advance=advance, gen_cfg=self.gen_cfg, seen=self.seen)
        if hk is not None:
            hk["t"] += 1
            if hk["criteria"]:
                # padt.py:752 `unfinished_sequences = unfinished_sequences & ~stopping_criteria(input_ids, scores)` for the caller's criteria (EOS and
                # the length limit are the loop's own: padt_greedy_step / max_new_tokens)
                kept = tuple(self.scores[i, : hk["B"], : hk["table_rows"]] for i in range(hk["t"])) if hk["pass_scores"] else None
                stop = _call_hooks(hk["criteria"], hk["sequences"](), kept, chain=False)
                if not isinstance(stop, torch.Tensor):       # criteria of the old API answer one bool for the whole batch
                    stop = torch.full((hk["B"],), bool(stop), device=hn.device)
                self.unfinished[: hk["B"]].mul_((~stop.to(device=hn.device, dtype=torch.bool)).to(self.unfinished.dtype))

    def run_steps(self, n: int, use_graph: bool = True):
        if n <= 0:
            return
        one_step = self.step_kernels if self.step_fn is None else (lambda: self.step_fn(self))    # reference precision: reference.ReferencePath.step
        if not use_graph:
            for _ in range(n):
                one_step()
            return
        gkey = (self.do_sample, self.keep_scores, self.step_fn is not None)   # different kernel sequences → one captured graph per mode
        if gkey not in self.graphs:
            one_step()                                       # real step; also pays every one-time kernel attribute call
            n -= 1
            g = torch.cuda.CUDAGraph()
            # thread_local: only THIS thread's calls are checked during capture.  With world > 1 the process group's watchdog thread polls
            # the events of an in-flight result gather (pipeline.ResultExchange) while a later lane captures its graph; under the default
            # global mode such a query from another thread invalidates the capture.
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                one_step()
            self.graphs[gkey] = g
        t = ops.STEP_TIMER
        ev = t.begin() if t is not None else None
        for _ in range(n):
            self.graphs[gkey].replay()
        if ev is not None:
            t.end(ev, (n, self.B))


class LanguageModel:
    def __init__(self, cfg: PaDTConfig, W, device):
        self.cfg, self.W, self.device = cfg, W, device
        self._sessions = {}

    def prototypes(self, image_embeds: torch.Tensor, out: Optional[torch.Tensor] = None, nf=None) -> torch.Tensor:
        """padt.py:187-191: LayerNorm then + W2(W1 x).  nf: int32 device flag set when a prototype row is not finite."""
        W = self.W
        if not self.cfg.use_visual_prototype_projection:
            if out is None:
                out = image_embeds.clone()
            else:
                out.copy_(image_embeds)
        else:
            p = ops.layernorm(image_embeds, W["proto.norm.w"], W["proto.norm.b"], eps=1e-5)
            t = ops.gemm(p, W["proto.0.w"])
            out = ops.gemm(t, W["proto.1.w"], out=out, epilogue=ops.EPI_RESID, residual=p)
        if nf is not None:
            ops.check_finite(out, nf)
        return out

    def session(self, B: int, need_s: int, need_np: int, need_t: int, lane: int = 0, grow: bool = True) -> Optional[DecodeSession]:
        """The lane's session for B rows, (re)allocated when too small; with grow=False returns None instead (a session
        that already holds other batches' KV rows must not be replaced)."""
        s_max = (need_s + 63) // 64 * 64
        key = (B, lane)
        s = self._sessions.get(key)
        if not grow:
            return s if (s is not None and s.s_max >= s_max and s.np_max >= need_np and s.t_max >= need_t) else None
        if s is None or s.s_max < s_max or s.np_max < need_np or s.t_max < need_t:
            s = DecodeSession(self.cfg, self.W, B, max(s_max, s.s_max if s else 0), max(need_np, s.np_max if s else 0),
                              max(need_t, s.t_max if s else 0), self.device)
            self._sessions[key] = s
        return s

    def prefill(self, plan: PromptPlan, image_embeds: torch.Tensor, sess: DecodeSession, nf=None):
        """Packed prefill; fills the session's KV caches; returns the post-norm hidden states of all prompt tokens (T,D).
        nf: int32 device flag set when a post-norm row is not finite (the residual stream absorbs every upstream inf / NaN)."""
        cfg, W = self.cfg, self.W
        Hq, Hkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        T = plan.ids.numel()
        dev = image_embeds.device
        bf = W.op16
        x = ops.embed_tokens(plan.ids, plan.img_index, W["llm.embed"], sess.proto, image_embeds, err_flag=sess.err)
        x32 = ops.cast_x16_f32(x) if W.resid_f32 else None       # fp32 residual stream; x stays its 16-bit mirror
        eps_n = W.eps_m(cfg.rms_norm_eps) if x32 is not None else cfg.rms_norm_eps
        if x32 is not None and ops.stream_scale(bf) != 1.0:
            ops.cast_f32_x16(x32, out=x, scale=ops.stream_scale(bf))           # mirrors hold X(scale * x32) (fp16: 2^-4, see ops.stream_scale)
        n = torch.empty_like(x)
        rstd = torch.empty((T,), device=dev, dtype=torch.float32)
        qkv = torch.empty((T, (Hq + 2 * Hkv) * hd), device=dev, dtype=bf)
        q = torch.empty((T, Hq * hd), device=dev, dtype=bf)
        kp = torch.empty((T, Hkv * hd), device=dev, dtype=bf)
        att = torch.empty((T, Hq * hd), device=dev, dtype=bf)
        h = torch.empty((T, W.llm_ipad), device=dev, dtype=bf)
        mx = max(plan.lens)
        f8 = W.fp8_prefill and x32 is not None
        if f8:                                                       # e4m3 images of the three GEMM inputs of a layer + their row scales
            x8 = torch.empty((T, cfg.hidden_size), device=dev, dtype=torch.uint8)
            a8 = torch.empty((T, Hq * hd), device=dev, dtype=torch.uint8)
            h8 = torch.empty((T, W.llm_ipad), device=dev, dtype=torch.uint8)
            rs8 = torch.empty((T,), device=dev, dtype=torch.float32)
        for i in range(cfg.num_hidden_layers):
            p = f"llm.{i}."
            # fp8 x fp8 MFMA path (llm_weights="fp8"): the GEMM input rows are quantised to e4m3 (power-of-two row scale x the folded norm's
            # rstd) and multiplied with the e4m3 weight image; a projection whose shape the fp8 kernel does not take keeps the bf16 GEMM
            if f8 and (p + "qkv.w8") in W:
                ops.quant_rows_fp8(x, norm_eps=eps_n, out=x8, rs=rs8)
                ops.gemm_fp8(x8, W[p + "qkv.w8"], W[p + "qkv.ws"], rs8, bias=W[p + "qkv.b"], out=qkv)
            else:
                ops.row_rstd(x, eps=eps_n, out=rstd)                       # norm weight is folded into qkv.w
                ops.gemm(x, W[p + "qkv.w"], W[p + "qkv.b"], out=qkv, row_scale=rstd)
            ops.llm_qkv_post(qkv, plan.pos3, sess.inv_freq, q, sess.kc[i], sess.vtc[i], Hq, Hkv, hd, sess.s_max,
                             cfg.mrope_section, sample=plan.sample, slot=plan.slot, k_pack=kp, cache_packed=sess.cache_packed)
            ops.attn_varlen(q, kp, qkv[:, (Hq + Hkv) * hd:], att, plan.cu, plan.cu, mx, Hq, Hkv, hd, causal=True)
            if f8 and (p + "o.w8") in W:
                ops.quant_rows_fp8(att, out=a8, rs=rs8)
                ops.gemm_fp8(a8, W[p + "o.w8"], W[p + "o.ws"], rs8, epilogue=ops.EPI_RESID, x32=x32, xb=x)
            elif x32 is not None:
                ops.gemm_resid32(att, W[p + "o.w"], None, x32, x)
            else:
                ops.gemm(att, W[p + "o.w"], out=x, epilogue=ops.EPI_RESID, residual=x)
            if f8 and (p + "gu.w8") in W:
                ops.quant_rows_fp8(x, norm_eps=eps_n, out=x8, rs=rs8)
                ops.gemm_fp8(x8, W[p + "gu.w8"], W[p + "gu.ws"], rs8, out=h, epilogue=ops.EPI_SWIGLU)
            else:
                ops.row_rstd(x, eps=eps_n, out=rstd)                       # norm weight is folded into gu.w
                ops.gemm(x, W[p + "gu.w"], out=h, epilogue=ops.EPI_SWIGLU, row_scale=rstd)
            if f8 and (p + "down.w8") in W:
                ops.quant_rows_fp8(h, out=h8, rs=rs8)
                ops.gemm_fp8(h8, W[p + "down.w8"], W[p + "down.ws"], rs8, epilogue=ops.EPI_RESID, x32=x32, xb=x)
            elif x32 is not None:
                ops.gemm_resid32(h, W[p + "down.w"], None, x32, x)
            else:
                ops.gemm(h, W[p + "down.w"], out=x, epilogue=ops.EPI_RESID, residual=x)
        if x32 is not None:
            ops.rmsnorm_f32(x32, W["llm.norm"], out=n, eps=cfg.rms_norm_eps)
        else:
            ops.rmsnorm(x, W["llm.norm"], out=n, eps=cfg.rms_norm_eps)
        if nf is not None:
            ops.check_finite(n, nf)
        return n
