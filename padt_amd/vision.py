"""ViT patch encoder — host orchestration of the HIP kernels (``custom_visual_forward``, padt.py:48-106).

Index preparation (window permutation, (h,w) rotary table, segment boundaries) is integer/host work, cached per grid
signature; everything arithmetic runs in libpadt_hip.so.  Outputs follow the reference's conventions literally
(SURVEY.md Appendix C.1): merged tokens in RASTER order, pre-merger tokens and cos/sin in WINDOW order.
"""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .config import PaDTConfig


# ------------------------------------------------------------------ integer prep (HF vision_utils.py:111-127,155-188)
def vision_position_ids(grid_thw: List[List[int]], merge: int) -> torch.Tensor:
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h).unsqueeze(1).expand(h, w)
        wp = torch.arange(w).unsqueeze(0).expand(h, w)
        shp = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shp).permute(0, 2, 1, 3).flatten()
        wp = wp.reshape(shp).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    return torch.cat(out, dim=0)


def window_index(grid_thw: List[List[int]], merge: int, window_size: int, patch_size: int):
    """→ (window_index over merged tokens, cu_window_seqlens in patches, consecutive duplicates removed)."""
    win = window_size // merge // patch_size
    unit = merge * merge
    idx_all, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        lh, lw = h // merge, w // merge
        index = torch.arange(t * lh * lw).reshape(t, lh, lw)
        pad_h = win - lh % win                      # a full extra (empty) window when divisible — kept as upstream
        pad_w = win - lw % win
        nh, nw = (lh + pad_h) // win, (lw + pad_w) // win
        ip = F.pad(index, (0, pad_w, 0, pad_h), "constant", -100)
        ip = ip.reshape(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win, win)
        seqlens = (ip != -100).sum([2, 3]).reshape(-1)
        ip = ip.reshape(-1)
        idx_all.append(ip[ip != -100] + base)
        cu.extend((seqlens.cumsum(0) * unit + cu[-1]).tolist())
        base += t * lh * lw
    cu_t = torch.unique_consecutive(torch.tensor(cu, dtype=torch.int32))
    return torch.cat(idx_all), cu_t


class VisionPlan:
    """Device-resident index tables for one batch signature (tuple of (t,h,w))."""

    def __init__(self, cfg: PaDTConfig, grid: Tuple[Tuple[int, int, int], ...], device):
        v = cfg.vision_config
        mu = cfg.merge_unit
        g = [list(x) for x in grid]
        win_idx, cu_win = window_index(g, v.spatial_merge_size, v.window_size, v.patch_size)
        P = sum(t * h * w for t, h, w in g)
        self.P, self.N = P, P // mu
        # patch-level permutation: groups of `mu` patches move together (padt.py:70-72)
        patch_perm = (win_idx[:, None] * mu + torch.arange(mu)[None, :]).reshape(-1)
        hd = v.hidden_size // v.num_heads
        dim = hd // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
        pos = vision_position_ids(g, v.spatial_merge_size)
        freqs = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)[patch_perm]      # window order (padt.py:73-75)
        emb = torch.cat((freqs, freqs), dim=-1)
        seg = [0]
        for t, h, w in g:
            for _ in range(t):
                seg.append(seg[-1] + h * w)
        self.cos = emb.cos().contiguous().to(device)
        self.sin = emb.sin().contiguous().to(device)
        self.patch_perm = patch_perm.to(torch.int32).to(device)
        self.reverse = torch.argsort(win_idx).to(torch.int32).to(device)
        self.cu_win = cu_win.to(device)
        self.cu_full = torch.tensor(seg, dtype=torch.int32, device=device)
        self.max_win = int((cu_win[1:] - cu_win[:-1]).max())
        self.max_full = max(b - a for a, b in zip(seg[:-1], seg[1:]))


class VisionEncoder:
    def __init__(self, cfg: PaDTConfig, W, device):
        self.cfg, self.W, self.device = cfg, W, device
        self._plans: Dict[tuple, VisionPlan] = {}

    def plan(self, grid_thw: torch.Tensor) -> VisionPlan:
        key = tuple(tuple(int(x) for x in r) for r in grid_thw.tolist())
        if key not in self._plans:
            self._plans[key] = VisionPlan(self.cfg, key, self.device)
            torch.cuda.current_stream().synchronize()        # tables are shared by every stream that runs this grid later
        return self._plans[key]

    def block(self, i: int, x, plan: VisionPlan, rstd, qkv, att, hbuf, force_full=None, x32=None):
        """One ViT block in place on x (P, vh): x += proj(attn(rope(qkv(RMSNorm(x))))); x += down(SwiGLU(RMSNorm(x)))
        (HF:297-321).  Window segments except for the full-attention layers (padt.py:89-93).
        x32 given: the residual stream is the fp32 tensor x32, updated in place by the residual GEMMs' epilogues, and x is its bf16
        mirror (the A operand of the qkv / gate-up GEMMs), rewritten by the same epilogues."""
        cfg, W = self.cfg, self.W
        v = cfg.vision_config
        vh, H = v.hidden_size, v.num_heads
        hd = vh // H
        p = f"vit.{i}."
        full = (i in v.fullatt_block_indexes) if force_full is None else force_full
        cu, mx = (plan.cu_full, plan.max_full) if full else (plan.cu_win, plan.max_win)
        eps = W.eps_m(1e-6) if x32 is not None else 1e-6                   # x is the (scaled) mirror of x32: rstd comes out as rstd / scale
        ops.row_rstd(x, eps=eps, out=rstd)                                 # RMSNorm = rstd x (weight folded into qkv.w)
        # qkv projection with the rotary embedding applied in its epilogue (q / k columns are pair-interleaved per head by
        # prepare_weights): no separate pass over q and k
        if W.vit_rope_fused:
            ops.gemm_rope(x, W[p + "qkv.w"], W[p + "qkv.b"], qkv, plan.cos, plan.sin, 2 * vh, hd, row_scale=rstd)
        else:                                                              # head widths the pair epilogue cannot take
            ops.gemm(x, W[p + "qkv.w"], W[p + "qkv.b"], out=qkv, row_scale=rstd)
            ops.rope_half_(qkv, plan.cos, plan.sin, 2 * H, hd)
        ops.attn_varlen(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], att, cu, cu, mx, H, H, hd)
        if x32 is not None:
            ops.gemm_resid32(att, W[p + "proj.w"], W[p + "proj.b"], x32, x)
        else:
            ops.gemm(att, W[p + "proj.w"], W[p + "proj.b"], out=x, epilogue=ops.EPI_RESID, residual=x)
        ops.row_rstd(x, eps=eps, out=rstd)
        ops.gemm(x, W[p + "gu.w"], W[p + "gu.b"], out=hbuf, epilogue=ops.EPI_SWIGLU, row_scale=rstd)
        if x32 is not None:
            ops.gemm_resid32(hbuf, W[p + "down.w"], W[p + "down.b"], x32, x)
        else:
            ops.gemm(hbuf, W[p + "down.w"], W[p + "down.b"], out=x, epilogue=ops.EPI_RESID, residual=x)

    def __call__(self, pixel_values: torch.Tensor, grid_thw: torch.Tensor, proto_out=None, nf=None):
        """pixel_values (P, C*T*p*p) fp32 / bf16 / fp16 on device → (image_embeds (N,D), high_res (P,vh), (cos,sin) (P,hd)).
        nf: int32 device flag, set when the encoder's output rows are not finite (an fp16 operand overflowed in some block; ops.check_finite)."""
        cfg, W = self.cfg, self.W
        v = cfg.vision_config
        plan = self.plan(grid_thw)
        P, vh, H = plan.P, v.hidden_size, v.num_heads
        hd = vh // H
        if pixel_values.shape[0] != P:
            raise ValueError(f"pixel_values has {pixel_values.shape[0]} rows, image_grid_thw implies {P}")
        op16 = W.op16                                                          # the operand type the weights were prepared in
        if pixel_values.dtype == op16:
            pix = pixel_values
        elif pixel_values.dtype == torch.float32:
            pix = ops.cast_f32_x16(pixel_values.contiguous(), dtype=op16)
        else:                                                                  # the other 16-bit type: through fp32 (exact), one rounding
            pix = ops.cast_f32_x16(ops.cast_x16_f32(pixel_values.contiguous()), dtype=op16)
        f32 = W.resid_f32
        x0 = ops.gemm(pix, W["vit.patch_embed"], out_f32=f32)                  # conv3d-as-GEMM (HF:116-122)
        x32 = None
        if f32:                                                                # fp32 residual stream + its bf16 mirror
            x32 = ops.gather_rows(x0, plan.patch_perm)                         # window order
            x = ops.cast_f32_x16(x32, dtype=op16, scale=ops.stream_scale(op16))   # the stream's first mirror
        else:
            x = ops.gather_rows(x0, plan.patch_perm)
        n = torch.empty_like(x)
        rstd = torch.empty((P,), device=x.device, dtype=torch.float32)
        qkv = torch.empty((P, 3 * vh), device=x.device, dtype=x.dtype)
        att = torch.empty_like(x)
        hbuf = torch.empty((P, W.vit_ipad), device=x.device, dtype=x.dtype)
        for i in range(v.depth):
            self.block(i, x, plan, rstd, qkv, att, hbuf, x32=x32)
        high = x32 if f32 else x                                               # the PaDT decoder reads fp32 or bf16 rows
        if nf is not None:                                                     # inf / NaN of any block is absorbing in the residual stream
            ops.check_finite(high, nf)
        if f32:
            ops.rmsnorm_f32(x32, W["vit.merger.ln_q"], out=n)
        else:
            ops.rmsnorm(x, W["vit.merger.ln_q"], out=n)
        m = ops.gemm(n.view(plan.N, vh * cfg.merge_unit), W["vit.merger.0.w"], W["vit.merger.0.b"], epilogue=ops.EPI_GELU)
        low_win = ops.gemm(m, W["vit.merger.2.w"], W["vit.merger.2.b"])
        low = ops.gather_rows(low_win, plan.reverse)                           # raster order (padt.py:103-104)
        return low, high, (plan.cos.clone(), plan.sin.clone())     # the caller owns past_visual_pe (plan tables are cached)
