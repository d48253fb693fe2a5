"""Thin torch-tensor → C-ABI wrappers.  PyTorch is used for device memory and streams only; every arithmetic op below
is a hand-written gfx950 kernel in libpadt_hip.so (see include/padt_hip.h).  No fallbacks."""
import torch

from . import _lib

EPI_NONE, EPI_GELU, EPI_RESID, EPI_SWIGLU = 0, 1, 2, 3
BF16 = torch.bfloat16
F16 = torch.float16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == BF16 and t.stride(-1) == 1, (t.dtype, t.device, t.stride())


def _x16(*ts):
    """→ the ONE 16-bit operand type (bf16 or fp16) of the given device tensors (None entries skipped); mixing them is an error."""
    dt = None
    for t in ts:
        if t is None:
            continue
        assert t.is_cuda and t.dtype in (BF16, F16) and t.stride(-1) == 1, (t.dtype, t.device, t.stride())
        assert dt is None or t.dtype == dt, ("mixed 16-bit operand types", dt, t.dtype)
        dt = t.dtype
    return dt


def _fn(name, dt):
    """The library entry point of `name` for operand type dt: the bf16 instantiation keeps the header's names, its fp16 twin replaces
    `bf16` by `f16` in the name or appends `_f16` (include/padt_hip_f16.h)."""
    lib = _lib.load()
    if dt != F16:
        return getattr(lib, name)
    return getattr(lib, name.replace("bf16", "f16") if "bf16" in name else name + "_f16")


_STREAM_SCALE = {}


def stream_scale(dt):
    """Factor between an fp32 residual stream and its 16-bit mirror of type dt (1 for bf16, 2^-4 for fp16: padt_stream_scale)."""
    if dt not in _STREAM_SCALE:
        _STREAM_SCALE[dt] = float(_lib.load().padt_stream_scale(1 if dt == F16 else 0))
    return _STREAM_SCALE[dt]


def mirror_eps(eps, dt):
    """The eps a scale-invariant consumer of a stream mirror (row_rstd, the fused RMSNorm of gemm_packed / quant_rows_fp8) must be given
    so that it returns rstd(x) / stream_scale: rsqrt(mean((s x)^2) + s^2 eps) = rstd(x) / s."""
    return float(eps) * stream_scale(dt) ** 2


GEMM_LOG = None   # bench.py sets this to a list to record (and later replay) every GEMM launch of one step


class EventTimer:
    """HIP-event brackets on the stream a kernel is launched on (bench.py's in-situ roofline: torch.cuda.Event only sees torch's current
    stream object, these see whatever stream the launch uses).  begin() / end(ev, meta) around a launch; results() → [(ms, meta)]."""

    def __init__(self):
        import ctypes
        self._ct = ctypes
        self.recs = []
        self._free = []
        self._all = []

    def _event(self):
        if self._free:
            return self._free.pop()
        ev = self._ct.c_void_p()
        _lib.check(_lib.load().padt_event_create(self._ct.byref(ev)), "padt_event_create")
        self._all.append(ev)
        return ev

    def begin(self):
        ev = self._event()
        _lib.load().padt_event_record(ev, _stream())
        return ev

    def end(self, ev0, meta):
        ev1 = self._event()
        _lib.load().padt_event_record(ev1, _stream())
        self.recs.append((ev0, ev1, meta))

    def results(self):
        out = []
        ms = self._ct.c_float()
        for ev0, ev1, meta in self.recs:
            _lib.check(_lib.load().padt_event_elapsed_ms(ev0, ev1, self._ct.byref(ms)), "padt_event_elapsed_ms")
            out.append((ms.value, meta))
            self._free += [ev0, ev1]
        self.recs = []
        return out

    def close(self):
        for ev in self._all:
            _lib.load().padt_event_destroy(ev)
        self._all, self._free, self.recs = [], [], []


class GemmProfile:
    """In-kernel timing of every tile-GEMM call (rows > 64) between start() and stop() (padt_gemm_profile): the kernels fold their first
    block start / last block end (100 MHz wall-clock ticks) into one slot per call — no event packets in the queue, nothing serialised,
    the dispatch gap in front of a kernel is not counted.  results() → [(ms, (kind, M, N, K))] in call order."""

    def __init__(self, capacity, device):
        self.cap = int(capacity)
        self.slots = torch.empty((self.cap, 2), dtype=torch.int64, device=device)
        self.meta = []

    def start(self):
        global GEMM_PROF
        if torch.cuda.is_current_stream_capturing():                 # the slot pointer would be baked into the captured graph
            raise RuntimeError("GemmProfile.start() during a stream capture")
        self.slots[:, 0] = -1                                        # 0xFFFF...: the atomicMin identity
        self.slots[:, 1] = 0
        self.meta = []
        torch.cuda.synchronize()
        _lib.load().padt_gemm_profile(self.slots.data_ptr(), self.cap)
        GEMM_PROF = self

    def stop(self):
        global GEMM_PROF
        GEMM_PROF = None
        n = _lib.load().padt_gemm_profile(None, 0)
        torch.cuda.synchronize()
        assert n == len(self.meta) or n == self.cap, (n, len(self.meta))
        return n

    def results(self):
        t = self.slots[: min(len(self.meta), self.cap)].cpu()
        ticks = (t[:, 1] - t[:, 0]).tolist()
        return [(d * 1e-5, m) for d, m in zip(ticks, self.meta)]     # 100 MHz ticks → ms


GEMM_PROF = None    # a started GemmProfile: the wrappers below append (kind, M, N, K) for every call the library gives a slot
STEP_TIMER = None   # an EventTimer: every chunk of decode-step graph replays is bracketed (llm.DecodeSession.run_steps), meta = (steps, rows)


def _tg_note(kind, M, N, K):
    if GEMM_PROF is not None and M > 64:
        GEMM_PROF.meta.append((kind, M, N, K))


def row_rstd(x, eps=1e-6, out=None):
    """fp32 rsqrt(mean(x^2) + eps) per row — feeds gemm(..., row_scale=) for an RMSNorm whose weight is folded into W."""
    M, D = x.shape
    if out is None:
        out = torch.empty((M,), device=x.device, dtype=torch.float32)
    _lib.check(_fn("padt_row_rstd", _x16(x))(_stream(), _p(x), x.stride(0), _p(out), M, D, float(eps)), "padt_row_rstd")
    return out


def gemm(a, w, bias=None, out=None, epilogue=EPI_NONE, residual=None, out_f32=False, K=None, row_scale=None):
    """out[M,N'] = epi(row_scale[m] * (a[M,K] @ w[N,K]^T) + bias).  a/w/out may be row-strided 2-D views.  N' = N/2 for SwiGLU."""
    dt = _x16(a, w, bias, residual)
    M = a.shape[0]
    N = w.shape[0]
    K = K if K is not None else a.shape[1]
    assert w.shape[1] >= K or w.shape[1] == K, (w.shape, K)
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=torch.float32 if out_f32 else dt)
    assert out.stride(-1) == 1 and out.shape[0] == M and out.shape[1] >= n_out and out.dtype == (torch.float32 if out_f32 else dt)
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.numel() >= M and row_scale.is_contiguous()
    if GEMM_LOG is not None:
        GEMM_LOG.append((a, w, bias, out, epilogue, residual, out_f32, K, row_scale))
    _tg_note("gemm", M, N, K)
    _lib.check(_fn("padt_gemm_bf16", dt)(_stream(), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0),
                                  _p(residual), residual.stride(0) if residual is not None else 0, M, N, K, epilogue,
                                  1 if out_f32 else 0, _p(row_scale)), "padt_gemm_bf16")
    return out


def gemm_resid32(a, w, bias, x32, xb=None):
    """fp32 residual stream: x32[M,N] += a @ w^T + bias (in place); xb = X(stream_scale * x32) is the next projection's A operand."""
    dt = _x16(a, w, bias, xb)
    M, K = a.shape
    N = w.shape[0]
    assert x32.dtype == torch.float32 and x32.stride(-1) == 1 and x32.shape[0] >= M and x32.shape[1] >= N
    if GEMM_LOG is not None:
        GEMM_LOG.append(("r32", a, w, bias, x32, xb))
    _tg_note("r32", M, N, K)
    _lib.check(_fn("padt_gemm_resid32", dt)(_stream(), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(x32), x32.stride(0), _p(xb),
                                     xb.stride(0) if xb is not None else 0, M, N, K), "padt_gemm_resid32")
    return x32


def quant_rows_fp8(x, norm_eps=None, out=None, rs=None):
    """bf16 rows → (e4m3 bytes (M, K) uint8, fp32 row scales (M,)) for gemm_fp8; norm_eps: fold rsqrt(mean(x^2) + eps) into the row scale."""
    dt = _x16(x)
    M, K = x.shape
    if out is None:
        out = torch.empty((M, K), device=x.device, dtype=torch.uint8)
    if rs is None:
        rs = torch.empty((M,), device=x.device, dtype=torch.float32)
    _lib.check(_fn("padt_quant_rows_fp8", dt)(_stream(), _p(x), x.stride(0), _p(out), out.stride(0), _p(rs), M, K,
                                               -1.0 if norm_eps is None else float(norm_eps)), "padt_quant_rows_fp8")
    return out, rs


def gemm_fp8(a8, w8, col_scale, row_scale, bias=None, out=None, epilogue=EPI_NONE, x32=None, xb=None, out_dtype=BF16):
    """fp8 x fp8 MFMA GEMM (padt_gemm_fp8): a8 (M, K) / w8 (N, K) uint8 e4m3, row_scale (M,) / col_scale (N,) fp32.
    epilogue EPI_NONE / EPI_SWIGLU → bf16 `out`; EPI_RESID → x32 += ... in place (+ bf16 mirror xb)."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(-1) == 1 and w8.stride(-1) == 1
    assert row_scale.dtype == torch.float32 and col_scale.dtype == torch.float32 and col_scale.is_contiguous()
    dt = _x16(bias, xb, out) or out_dtype
    M, K = a8.shape
    N = w8.shape[0]
    if epilogue == EPI_RESID:
        assert x32 is not None and x32.dtype == torch.float32 and x32.stride(-1) == 1
    else:
        n_out = N // 2 if epilogue == EPI_SWIGLU else N
        if out is None:
            out = torch.empty((M, n_out), device=a8.device, dtype=dt)
        assert _x16(out) == dt
    _tg_note("fp8", M, N, K)
    _lib.check(_fn("padt_gemm_fp8", dt)(_stream(), _p(a8), a8.stride(0), _p(w8), w8.stride(0), _p(row_scale), _p(col_scale), _p(bias), _p(out),
                                 out.stride(0) if out is not None else 0, _p(x32), x32.stride(0) if x32 is not None else 0, _p(xb),
                                 xb.stride(0) if xb is not None else 0, M, N, K, int(epilogue)), "padt_gemm_fp8")
    return x32 if epilogue == EPI_RESID else out


def gemm_knobs(mode256=-1, mf=-1, peel=-1, colsplit=-1, group_m=-1):
    """Dispatch knobs of the 256-row tile kernel (tests / tuning tools); -1 keeps a field.  Defaults: (1, 0, 1, 1, 8)."""
    _lib.check(_lib.load().padt_gemm_knobs(int(mode256), int(mf), int(peel), int(colsplit), int(group_m)), "padt_gemm_knobs")


def gemm_rope(a, w, bias, out, cos, sin, rope_cols, head_dim, row_scale=None):
    """out = rope(row_scale[m] * (a @ w^T) + bias) with the leading rope_cols columns pair-interleaved per head (see
    weights.interleave_rope_rows): the ViT qkv projection with RoPE fused into the epilogue."""
    dt = _x16(a, w, bias, out)
    M, K = a.shape
    N = w.shape[0]
    assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.stride(0) == sin.stride(0)
    if GEMM_LOG is not None:
        GEMM_LOG.append((a, w, bias, out, EPI_NONE, None, False, None, row_scale))
    _tg_note("gemm", M, N, K)
    _lib.check(_fn("padt_gemm_rope_bf16", dt)(_stream(), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), out.stride(0), M, N, K,
                                       _p(row_scale), _p(cos), _p(sin), cos.stride(0), int(rope_cols), int(head_dim)),
               "padt_gemm_rope_bf16")
    return out


def gemm_rmsnorm(a, w, bias=None, out=None, epilogue=EPI_NONE, eps=1e-6):
    """out = epi(rstd(a) * (a @ w^T) + bias) for decode-sized batches (rows <= 64); w carries the folded norm weight."""
    dt = _x16(a, w, bias, out)
    M, K, N = a.shape[0], a.shape[1], w.shape[0]
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=dt)
    _lib.check(_fn("padt_gemm_rmsnorm_bf16", dt)(_stream(), _p(a), a.stride(0), float(eps), _p(w), w.stride(0), _p(bias), _p(out),
                                          out.stride(0), M, N, K, epilogue), "padt_gemm_rmsnorm_bf16")
    return out


def pack_weight(w):
    """[N][K] row-major → MFMA-fragment-packed [N/16][K/32][4 (k-quarter)][16 (row)][8] (N, K zero-padded to 16 / 32).
    One 16x32 tile = 1 KiB in exactly the lane order of the decode kernel's weight fragment load."""
    N, K = w.shape
    Np, Kp = (N + 15) // 16 * 16, (K + 31) // 32 * 32
    if (Np, Kp) != (N, K):
        wp = w.new_zeros((Np, Kp))
        wp[:N, :K] = w
        w = wp
    return w.view(Np // 16, 16, Kp // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(Np, Kp)


def quantize_fp8_rows(w, deq_dtype=BF16):
    """Per-output-row fp8 quantisation with POWER-OF-TWO scales: → (wq uint8 OCP e4m3 bits [N][K], scale fp32 [N], w_deq bf16 [N][K]).
    scale[n] = 2^ceil(log2(max|w[n]| / 448)); e4m3 x 2^k is exactly representable in bf16, so w_deq (what prefill multiplies with and
    what the oracle sees) and scale * wq (what the decode kernel computes) are the same numbers, bit for bit."""
    wf = w.float()
    amax = wf.abs().amax(dim=1).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[:, None]).to(deq_dtype)
    return q.view(torch.uint8), scale.contiguous(), deq


def pack_weight_fp8(wq):
    """uint8 e4m3 [N][K] → fp8 fragment-packed image [N/16][K/64][64 lanes][16 B] (N, K zero-padded to 16 / 64): lane fq*16 + frow holds
    row n16*16 + frow, k = kp*64 + half*32 + fq*8 + e at byte half*8 + e."""
    N, K = wq.shape
    Np, Kp = (N + 15) // 16 * 16, (K + 63) // 64 * 64
    if (Np, Kp) != (N, K):
        t = wq.new_zeros((Np, Kp))
        t[:N, :K] = wq
        wq = t
    return wq.view(Np // 16, 16, Kp // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous().view(Np, Kp)


def gemm_packed_fp8(a, wq_packed, scales, n, bias=None, out=None, epilogue=EPI_NONE, residual=None, norm_eps=None, split_k=1, workspace=None,
                    a_packed=False, c_packed=False, rows=None):
    """gemm_packed over fp8 weights: out = epi(rstd?(a) * scales[n] * (a @ wq^T) + bias)."""
    dt = _x16(a, bias, residual, out)
    assert wq_packed.dtype == torch.uint8 and scales.dtype == torch.float32 and scales.is_contiguous()
    M, K = a.shape
    if rows is not None:
        M = rows
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=dt)
    _lib.check(_fn("padt_gemm_packed_fp8", dt)(_stream(), _p(a), a.stride(0), _p(wq_packed), wq_packed.shape[1], _p(scales), _p(bias), _p(out),
                                        out.stride(0), _p(residual), residual.stride(0) if residual is not None else 0, M, n, K, epilogue,
                                        -1.0 if norm_eps is None else float(norm_eps), int(split_k), _p(workspace),
                                        (1 if a_packed else 0) | (2 if c_packed else 0)), "padt_gemm_packed_fp8")
    return out


def new_splitk_workspace(n, split_k, device):
    """Zero-initialised split-K workspace (ticket header must start at zero), one per concurrently decoding stream."""
    return torch.zeros(_lib.load().padt_gemm_splitk_workspace(n, split_k), dtype=torch.uint8, device=device)


def pack_rows(src, dst, M, to_packed=True):
    """Row-major (M, K) <-> 16-row fragment-packed activation layout (buffers hold whole 16-row blocks)."""
    K = src.shape[1]
    _lib.check(_lib.load().padt_pack_rows(_stream(), _p(src), src.stride(0), _p(dst), dst.stride(0), M, K, 1 if to_packed else 0),
               "padt_pack_rows")
    return dst


def gemm_packed(a, wp, n, bias=None, out=None, epilogue=EPI_NONE, residual=None, norm_eps=None, split_k=1, workspace=None,
                a_packed=False, c_packed=False, rows=None):
    """Decode-step projection over a pack_weight() image (rows <= 128): out = epi(rstd?(a) * (a @ w^T) + bias).
    a_packed / c_packed: a / (out and residual) are fragment-packed activation buffers holding `rows` valid rows."""
    dt = _x16(a, wp, bias, residual, out)
    M, K = a.shape
    if rows is not None:
        M = rows
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=dt)
    _lib.check(_fn("padt_gemm_packed_bf16", dt)(_stream(), _p(a), a.stride(0), _p(wp), wp.shape[1], _p(bias), _p(out), out.stride(0),
                                         _p(residual), residual.stride(0) if residual is not None else 0, M, n, K, epilogue,
                                         -1.0 if norm_eps is None else float(norm_eps), int(split_k), _p(workspace),
                                         (1 if a_packed else 0) | (2 if c_packed else 0)), "padt_gemm_packed_bf16")
    return out


def gemm_packed_resid32(a, wp, n, x32, xb_packed, scales=None, split_k=1, workspace=None, a_packed=True, rows=None):
    """Decode-step residual projection over the fp32 stream: x32[rows, n] += scales?[n] * (a @ w^T) in place, xb_packed = bf16(x32) in the
    fragment-packed activation layout.  wp: pack_weight() image, or with scales the fp8 image."""
    dt = _x16(a, xb_packed, None if scales is not None else wp)
    M, K = a.shape
    if rows is not None:
        M = rows
    assert x32.dtype == torch.float32 and x32.stride(-1) == 1 and (scales is None or scales.dtype == torch.float32)
    _lib.check(_fn("padt_gemm_packed_resid32", dt)(_stream(), _p(a), a.stride(0), _p(wp), wp.shape[1], _p(scales), _p(x32), x32.stride(0),
                                            _p(xb_packed), xb_packed.stride(0), M, n, K, int(split_k), _p(workspace), 1 if a_packed else 0),
               "padt_gemm_packed_resid32")
    return x32


def attn_varlen(q, k, v, out, cu_q, cu_k, max_seqlen_q, n_heads, n_kv_heads, head_dim, causal=False, scale=None, rope=None):
    """q/k/v/out: 2-D (tokens, row) bf16 views whose row holds the heads contiguously; cu_*: int32 device tensors.
    rope=(cos, sin) fp32 [tokens][>= head_dim/2]: rotate q and k inside the kernel (self-attention, max_seqlen_q < 256)."""
    dt = _x16(q, k, v, out)
    assert cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    nseg = cu_q.numel() - 1
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_fn("padt_attn_varlen", dt)(_stream(), _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out),
                                    out.stride(0), _p(cu_q), _p(cu_k), nseg, int(max_seqlen_q), n_heads, n_kv_heads,
                                    head_dim, float(scale), 1 if causal else 0, _p(rope[0]) if rope else 0,
                                    _p(rope[1]) if rope else 0, rope[0].stride(0) if rope else 0), "padt_attn_varlen")
    return out


def decode_attn_workspace(batch, n_kv_heads, head_dim, s_max):
    return _lib.load().padt_decode_attn_workspace(batch, n_kv_heads, head_dim, s_max)


def new_decode_workspace(batch, n_kv_heads, head_dim, s_max, device):
    """Split-attention partials; one per stream that decodes concurrently."""
    return torch.zeros(decode_attn_workspace(batch, n_kv_heads, head_dim, s_max), dtype=torch.uint8, device=device)


def decode_attn(q, k_cache, vt_cache, lens, out, workspace, n_heads, n_kv_heads, head_dim, s_max, max_len, scale=None):
    dt = _x16(q, k_cache, vt_cache, out)
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_fn("padt_decode_attn", dt)(_stream(), _p(q), _p(k_cache), _p(vt_cache), _p(lens), _p(out), _p(workspace),
                                    q.shape[0], n_heads, n_kv_heads, head_dim, s_max, int(max_len), float(scale)),
               "padt_decode_attn")
    return out


def rope_table(pos3, inv_freq, out, head_dim, sections):
    lib = _lib.load()
    assert pos3.dtype == torch.int32 and out.dtype == torch.float32
    _lib.check(lib.padt_rope_table(_stream(), _p(pos3), _p(inv_freq), _p(out), pos3.shape[1], head_dim, sections[0], sections[1]),
               "padt_rope_table")
    return out


def decode_attn_rope(qkv, rope_cs, slot, k_cache, vt_cache, out, workspace, n_heads, n_kv_heads, head_dim, s_max, max_len,
                     scale=None, out_packed=False, cache_packed=False):
    """cache_packed: k_cache / vt_cache hold the fragment-packed images (pack_k_cache / pack_vt_cache, written by llm_qkv_post(cache_packed=True))
    → the one-launch kernel (no workspace needed); 2 / 3 force one / two blocks per (kv head, sample) (True = 1: chosen from Hkv x batch)."""
    dt = _x16(qkv, k_cache, vt_cache, out)
    scale = head_dim ** -0.5 if scale is None else scale
    _lib.check(_fn("padt_decode_attn_rope", dt)(_stream(), _p(qkv), qkv.stride(0), _p(rope_cs), _p(slot), _p(k_cache), _p(vt_cache),
                                         _p(out), _p(workspace), qkv.shape[0], n_heads, n_kv_heads, head_dim, s_max,
                                         int(max_len), float(scale), 1 if out_packed else 0, int(cache_packed)), "padt_decode_attn_rope")
    return out


def pack_k_cache(kc):
    """Row-major K cache (B, Hkv, S, D) → the fragment-packed image [S/16][D/32][fq 4][frow 16][8] (same shape, padt_decode_attn_rope)."""
    B, G, S, D = kc.shape
    return kc.view(B, G, S // 16, 16, D // 32, 4, 8).permute(0, 1, 2, 4, 5, 3, 6).contiguous().view(B, G, S, D)


def unpack_k_cache(kp):
    B, G, S, D = kp.shape
    return kp.view(B, G, S // 16, D // 32, 4, 16, 8).permute(0, 1, 2, 5, 3, 4, 6).reshape(B, G, S, D)


def pack_vt_cache(vt):
    """Transposed V cache (B, Hkv, D, S) → the fragment-packed image [D/16][S/32][fq 4][frow 16][run 2][4] (same shape)."""
    B, G, D, S = vt.shape
    return vt.view(B, G, D // 16, 16, S // 32, 2, 4, 4).permute(0, 1, 2, 4, 6, 3, 5, 7).contiguous().view(B, G, D, S)


def unpack_vt_cache(vp):
    B, G, D, S = vp.shape
    return vp.view(B, G, D // 16, S // 32, 4, 16, 2, 4).permute(0, 1, 2, 5, 3, 6, 4, 7).reshape(B, G, D, S)


def rmsnorm(x, w, out=None, eps=1e-6, add=None, add_div=1, D=None, gelu=False):
    dt = _x16(x, w, add, out)
    D = D if D is not None else x.shape[1]
    if out is None:
        out = torch.empty((x.shape[0], D), device=x.device, dtype=dt)
    _lib.check(_fn("padt_rmsnorm", dt)(_stream(), _p(x), x.stride(0), _p(add), add.stride(0) if add is not None else 0, add_div,
                                _p(w), _p(out), out.stride(0), x.shape[0], D, float(eps), 1 if gelu else 0), "padt_rmsnorm")
    return out


def layernorm(x, w, b, out=None, eps=1e-5):
    dt = _x16(x, w, b, out)
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_fn("padt_layernorm", dt)(_stream(), _p(x), x.stride(0), _p(w), _p(b), _p(out), out.stride(0), x.shape[0],
                                  x.shape[1], float(eps)), "padt_layernorm")
    return out


def rope_half_(x, cos, sin, n_heads, head_dim):
    """in place on the first n_heads*head_dim columns of x (T, row)."""
    dt = _x16(x)
    assert cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.stride(-1) == 1 and sin.stride() == cos.stride()
    _lib.check(_fn("padt_rope_half", dt)(_stream(), _p(x), x.stride(0), _p(cos), _p(sin), cos.stride(0), x.shape[0], n_heads,
                                  head_dim), "padt_rope_half")
    return x


def gather_rows(src, idx, out=None, D=None):
    lib = _lib.load()
    assert idx.dtype == torch.int32 and src.stride(-1) == 1
    D = D if D is not None else src.shape[1]
    n = idx.numel()
    if out is None:
        out = torch.empty((n, D), device=src.device, dtype=src.dtype)
    if src.dtype in (BF16, F16):                                     # a 16-byte-vector copy: the operand type does not matter
        assert out.dtype == src.dtype
        _lib.check(lib.padt_gather_rows(_stream(), _p(src), src.stride(0), _p(idx), _p(out), out.stride(0), n, D),
                   "padt_gather_rows")
    elif src.dtype == torch.float32:
        _lib.check(lib.padt_gather_rows_f32(_stream(), _p(src), src.stride(0), _p(idx), _p(out), out.stride(0), n, D),
                   "padt_gather_rows_f32")
    else:
        raise TypeError(src.dtype)
    return out


def add_rows(a, b, out=None):
    dt = _x16(a, b, out)
    if out is None:
        out = torch.empty_like(a)
    _lib.check(_fn("padt_add_rows", dt)(_stream(), _p(a), a.stride(0), _p(b), b.stride(0), b.shape[0], _p(out), out.stride(0),
                                 a.shape[0], a.shape[1]), "padt_add_rows")
    return out


def cast_f32_x16(x, D_pad=None, out=None, dtype=BF16, scale=1.0):
    """out = X(scale * x) for fp32 rows x, X = dtype (or out's); scale = stream_scale(dtype) writes the mirror of an fp32 residual stream."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    D = x.shape[1]
    D_pad = D_pad or D
    if out is None:
        out = torch.empty((x.shape[0], D_pad), device=x.device, dtype=dtype)
    dt = _x16(out)
    _lib.check(_fn("padt_cast_f32_bf16", dt)(_stream(), _p(x), x.stride(0), _p(out), out.stride(0), x.shape[0], D, D_pad, float(scale)),
               "padt_cast_f32_bf16")
    return out


def cast_f32_bf16(x, D_pad=None, out=None):
    return cast_f32_x16(x, D_pad, out, BF16)


def cast_x16_f32(x, out=None):
    dt = _x16(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _lib.check(_fn("padt_cast_bf16_f32", dt)(_stream(), _p(x), x.stride(0), _p(out), out.stride(0), x.shape[0], x.shape[1]), "padt_cast_bf16_f32")
    return out


cast_bf16_f32 = cast_x16_f32


def rmsnorm_f32(x32, w, out=None, eps=1e-6):
    """bf16 RMSNorm of fp32 rows (the norms that read the fp32 residual stream)."""
    dt = _x16(w, out)
    assert x32.dtype == torch.float32 and x32.stride(-1) == 1
    if out is None:
        out = torch.empty(x32.shape, device=x32.device, dtype=dt)
    _lib.check(_fn("padt_rmsnorm_f32", dt)(_stream(), _p(x32), x32.stride(0), _p(w), _p(out), out.stride(0), x32.shape[0], x32.shape[1], float(eps)),
               "padt_rmsnorm_f32")
    return out


_KIND = {torch.float32: 0, BF16: 1, F16: 2}


def check_finite(x, flags, rows_per_flag=None, rows=None):
    """flags[row // rows_per_flag] |= 1 (int32 device tensor, caller zeroes it) for every row of the 2-D tensor x that holds +-inf or NaN
    (padt_check_finite: the fp16-operand safety net).  rows_per_flag None → one flag for the whole tensor."""
    assert x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.dtype in _KIND, (x.dtype, x.shape, x.stride())
    assert flags.dtype == torch.int32 and flags.is_cuda and flags.is_contiguous()
    rows = x.shape[0] if rows is None else int(rows)
    rpf = max(rows, 1) if rows_per_flag is None else int(rows_per_flag)
    assert flags.numel() >= (rows + rpf - 1) // rpf
    _lib.check(_lib.load().padt_check_finite(_stream(), _p(x), x.stride(0), rows, x.shape[1], _KIND[x.dtype], _p(flags), rpf), "padt_check_finite")
    return flags


def sigmoid_f32_(x):
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    _lib.check(lib.padt_sigmoid_f32(_stream(), _p(x), x.numel()), "padt_sigmoid_f32")
    return x


def embed_tokens(ids, img_index, table, proto, image_embeds, out=None, err_flag=None):
    lib = _lib.load()
    dt = _x16(table, proto, image_embeds, out)                       # rows are copied: one entry point serves both operand types
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    T, D = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((T, D), device=table.device, dtype=dt)
    _lib.check(lib.padt_embed_tokens(_stream(), _p(ids), _p(img_index), _p(table), _p(proto), _p(image_embeds), _p(out),
                                     T, table.shape[0], 0 if proto is None else proto.shape[0], D, _p(err_flag)),
               "padt_embed_tokens")
    return out


def llm_qkv_post(qkv, pos3, inv_freq, q_out, k_cache, vt_cache, n_heads, n_kv_heads, head_dim, s_max, sections,
                 sample=None, slot=None, lens=None, k_pack=None, cache_packed=False):
    dt = _x16(qkv, q_out, k_cache, vt_cache, k_pack)
    assert pos3.dtype == torch.int32 and pos3.is_contiguous() and inv_freq.dtype == torch.float32
    _lib.check(_fn("padt_llm_qkv_post", dt)(_stream(), _p(qkv), qkv.stride(0), _p(pos3), _p(sample), _p(slot), _p(lens),
                                     _p(inv_freq), _p(q_out), q_out.stride(0), _p(k_pack),
                                     k_pack.stride(0) if k_pack is not None else 0, _p(k_cache), _p(vt_cache),
                                     qkv.shape[0], n_heads, n_kv_heads, head_dim, s_max, sections[0], sections[1], 1 if cache_packed else 0),
               "padt_llm_qkv_post")


def mask_scatter(e2, mask_tok, cu_patch, obj_w, masks, n_obj, total_patches, dm):
    lib = _lib.load()
    _chk_bf16(e2, mask_tok)
    assert masks.dtype == torch.float32 and masks.is_contiguous()
    _lib.check(lib.padt_mask_scatter(_stream(), _p(e2), e2.stride(0), _p(mask_tok), mask_tok.stride(0), _p(cu_patch),
                                     _p(obj_w), _p(masks), n_obj, total_patches, masks.shape[1], masks.shape[2], dm),
               "padt_mask_scatter")
    return masks


def vrt_head_nblk(vocab, n_proto):
    return _lib.load().padt_vrt_head_nblk(vocab, n_proto)


def vrt_head(hidden, table, proto, vrt_off, part_val, part_idx, eos, mode_table=None, step=None, logits=None,
             table_packed=None, rows=None, gen_cfg=None, seen=None):
    """table_packed: pack_weight(table) — then `hidden` is a fragment-packed activation buffer holding `rows` valid rows."""
    dt = _x16(hidden, table, proto, table_packed)
    B = hidden.shape[0] if rows is None else rows
    _lib.check(_fn("padt_vrt_head", dt)(_stream(), _p(hidden), hidden.stride(0), _p(table), table.shape[0], _p(proto),
                                 proto.shape[0], _p(vrt_off), _p(mode_table), _p(step), _p(logits),
                                 logits.stride(0) if logits is not None else 0, _p(part_val), _p(part_idx),
                                 B, hidden.shape[1], eos, _p(table_packed), _p(gen_cfg), _p(seen),
                                 seen.shape[1] if seen is not None else 0), "padt_vrt_head")


def seen_init(ids, rows, seen):
    """ids int64 (n,), rows int32 (n,) device tensors: set bit ids[i] in seen[rows[i]] (prompt tokens of a generate call)."""
    assert ids.dtype == torch.int64 and rows.dtype == torch.int32 and ids.numel() == rows.numel() and seen.dtype == torch.int32
    _lib.check(_lib.load().padt_seen_init(_stream(), _p(ids), _p(rows), ids.numel(), _p(seen), seen.shape[1]), "padt_seen_init")


def gen_cfg_tensor(repetition_penalty=1.0, eos_ids=(), device="cuda", do_sample=False, seed=0, temperature=1.0, top_k=0, top_p=1.0):
    """Device copy of the generation-config slots the head / greedy / sampling kernels read:
    {float penalty; int eos[4]; int do_sample; unsigned seed; float temperature; int top_k; float top_p; int pad[2]}."""
    import struct
    eos = list(eos_ids)[:4] + [-1] * (4 - min(4, len(eos_ids)))
    raw = struct.pack("<f4iiIfifii", float(repetition_penalty), *eos, 1 if do_sample else 0, int(seed) & 0xFFFFFFFF, float(temperature),
                      int(top_k), float(top_p), 0, 0)
    return torch.frombuffer(bytearray(raw), dtype=torch.int32).clone().to(device)


def sample_token(logits, n_rows_table, gen_cfg, step, part_val, part_idx, batch):
    """One multinomial draw per row from the warped fp32 logits (padt_sample_token) into the (value, index) partial layout."""
    assert logits.dtype == torch.float32 and logits.stride(-1) == 1
    _lib.check(_lib.load().padt_sample_token(_stream(), _p(logits), logits.stride(0), int(n_rows_table), _p(gen_cfg), _p(step), _p(part_val),
                                             _p(part_idx), int(batch)), "padt_sample_token")


def argmax_rows(scores, n_cols, part_val, part_idx, batch):
    """Arg-max per fp32 score row (ties → lowest index) into the (value, index) partial layout (padt_argmax_rows_f32): the hooked decode loop."""
    assert scores.dtype == torch.float32 and scores.stride(-1) == 1
    _lib.check(_lib.load().padt_argmax_rows_f32(_stream(), _p(scores), scores.stride(0), int(n_cols), _p(part_val), _p(part_idx), int(batch)),
               "padt_argmax_rows_f32")


def greedy_step(part_val, part_idx, nblk, hidden, hidden_buf, unfinished, tokens_out, cur_tok, step, slot, lens, pos3,
                eos, pad, advance=True, gen_cfg=None, seen=None):
    lib = _lib.load()
    B, D = hidden.shape
    _lib.check(lib.padt_greedy_step(_stream(), _p(part_val), _p(part_idx), nblk, B, D, eos, pad, tokens_out.shape[1],
                                    _p(unfinished), _p(tokens_out), _p(cur_tok), _p(step), _p(slot), _p(lens), _p(pos3),
                                    _p(hidden), _p(hidden_buf), 1 if advance else 0, _p(gen_cfg), _p(seen),
                                    seen.shape[1] if seen is not None else 0), "padt_greedy_step")


def collect_summary(err, unfinished, nf_rows, nf_batch, n_batch, tokens, done, eos, gen_cfg, out):
    """One launch: out = [err, any(unfinished), nf_rows…, nf_batch[:n_batch]…, first_eos step per row…] (padt_collect_summary)."""
    n_rows = unfinished.numel()
    assert out.dtype == torch.int32 and out.numel() >= 2 + 2 * n_rows + n_batch and tokens.dtype == torch.int64 and tokens.is_contiguous()
    _lib.check(_lib.load().padt_collect_summary(_stream(), _p(err), _p(unfinished), _p(nf_rows), _p(nf_batch), n_rows, int(n_batch), _p(tokens),
                                                tokens.shape[1], int(done), int(eos), _p(gen_cfg), _p(out)), "padt_collect_summary")
    return out


def assemble_sequences(input_ids, tokens, n_steps, vocab, proto_row0):
    """→ (B, L + n_steps) int64: [input_ids | tokens[:, :n_steps]] with session-global VRT ids shifted back by proto_row0 (padt.py:751)."""
    assert input_ids.dtype == torch.int64 and input_ids.stride(1) == 1 and tokens.dtype == torch.int64 and tokens.stride(1) == 1
    B, L = input_ids.shape
    out = torch.empty((B, L + int(n_steps)), dtype=torch.int64, device=tokens.device)
    _lib.check(_lib.load().padt_assemble_sequences(_stream(), _p(input_ids), input_ids.stride(0), L, _p(tokens), tokens.stride(0), int(n_steps),
                                                   int(vocab), int(proto_row0), _p(out), B), "padt_assemble_sequences")
    return out


def logit_mask(vrt_off, vocab, table_rows, proto_row0, batch):
    """→ (batch, table_rows) bool past_logit_mask (padt.py:196-201) from the session's prototype row offsets of the batch's rows."""
    assert vrt_off.dtype == torch.int32 and vrt_off.numel() >= batch + 1
    out = torch.empty((batch, table_rows), dtype=torch.uint8, device=vrt_off.device)
    _lib.check(_lib.load().padt_logit_mask(_stream(), _p(vrt_off), int(vocab), int(table_rows), int(proto_row0), _p(out), int(batch)), "padt_logit_mask")
    return out.view(torch.bool)


def stash_step_f32(src, step, dst):
    """dst[*step] = src (fp32, whole buffer) under the device step counter (padt_stash_step_f32): per-step score rows for output_scores."""
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst[0].numel() == src.numel()
    _lib.check(_lib.load().padt_stash_step_f32(_stream(), _p(src), src.numel(), _p(step), dst.shape[0], _p(dst)), "padt_stash_step_f32")
    return dst


def mask_upsample_binarize(masks, src_h, src_w, dst_h, dst_w, max_h, max_w, want_logits=False):
    """masks (n_obj, Hm, Wm) fp32 logits; src/dst sizes int32 device tensors (n_obj,) → uint8 (n_obj, max_h, max_w)
    [, fp32 up-sampled logits]."""
    lib = _lib.load()
    assert masks.dtype == torch.float32 and masks.stride(-1) == 1
    n = masks.shape[0]
    out = torch.zeros((n, max_h, max_w), dtype=torch.uint8, device=masks.device)
    up = torch.zeros((n, max_h, max_w), dtype=torch.float32, device=masks.device) if want_logits else None
    for t in (src_h, src_w, dst_h, dst_w):
        assert t.dtype == torch.int32 and t.numel() == n
    _lib.check(lib.padt_mask_upsample_binarize(_stream(), _p(masks), masks.stride(0), masks.stride(1), _p(src_h), _p(src_w),
                                               _p(dst_h), _p(dst_w), _p(out), out.stride(0), out.stride(1), _p(up),
                                               up.stride(0) if up is not None else 0, up.stride(1) if up is not None else 0,
                                               n, int(max_h), int(max_w)), "padt_mask_upsample_binarize")
    return (out, up) if want_logits else out


def _rle_caps(mh, mw, worst_case):
    """(runs, string bytes) capacity per object.  The worst case — every pixel its own run — is 4 + 2 bytes of scratch per PIXEL per object (an OVD batch
    of 56 objects at 640 x 640: 140 MB per call); a segmentation mask has a handful of runs per COLUMN, so the default bound is 128 runs per column
    (+ the zero-length first run; the noise-like masks random-init weights produce have ~80), 5 bytes per run (counts below 2^24).  A mask beyond it is reported (n_counts < 0) and the caller falls back to the
    host statement for that batch."""
    if worst_case:
        return mh * mw + 2, 2 * mh * mw + 16
    runs = min(mh * mw + 2, 128 * mw + 2)
    return runs, 5 * runs + 16


_RLE_SCRATCH = {}                                                     # (device, n, cap_c, cap_s) → buffers, reused by the next call of that shape on the stream


def mask_rle_launch(bin_masks, dst_h, dst_w, worst_case=False):
    """Enqueue padt_mask_rle for (n_obj, max_h, max_w) uint8 masks on the current stream; → handle for mask_rle_fetch (no host sync here, so a
    caller can put its other device work and copies behind the same single wait).  Scratch: bounded capacities (_rle_caps), one cached set per shape
    (a handle is valid until the next launch of the same shape: fetch first — postprocess does)."""
    lib = _lib.load()
    assert bin_masks.dtype == torch.uint8 and bin_masks.is_cuda and bin_masks.dim() == 3 and bin_masks.stride(2) == 1
    n, mh, mw = bin_masks.shape
    if n == 0:
        return None
    dev = bin_masks.device
    cap_c, cap_s = _rle_caps(mh, mw, worst_case)
    key = (dev, n, cap_c, cap_s)
    buf = _RLE_SCRATCH.get(key)
    if buf is None:
        if len(_RLE_SCRATCH) >= 8:
            _RLE_SCRATCH.clear()
        buf = dict(counts=torch.empty((n, cap_c), dtype=torch.int32, device=dev), strs=torch.empty((n, cap_s), dtype=torch.uint8, device=dev),
                   n_counts=torch.empty(n, dtype=torch.int32, device=dev), str_len=torch.empty(n, dtype=torch.int32, device=dev),
                   packed=torch.empty(n * cap_s, dtype=torch.uint8, device=dev), offsets=torch.empty(n + 1, dtype=torch.int32, device=dev))
        _RLE_SCRATCH[key] = buf
    h = dict(buf, n=n, masks=bin_masks, worst_case=worst_case)
    _lib.check(lib.padt_mask_rle(_stream(), _p(bin_masks), bin_masks.stride(0), bin_masks.stride(1), _p(dst_h), _p(dst_w), n, mh, _p(h["counts"]), cap_c,
                                 _p(h["strs"]), cap_s, _p(h["n_counts"]), _p(h["str_len"]), _p(h["packed"]), h["packed"].numel(), _p(h["offsets"])),
               "padt_mask_rle")
    return h


def mask_rle_fetch(h, want_counts=False, on_overflow="raise"):
    """→ list of n_obj ASCII COCO `counts` strings [, list of count lists]: the offset table, then exactly the string bytes.  A mask beyond the
    bounded scratch (or beyond max_h / ld_row): PaDTHipError, or None with on_overflow="none" (the caller computes the batch on the host)."""
    if h is None:
        return ([], []) if want_counts else []
    n = h["n"]
    off = h["offsets"].cpu().tolist()
    if off[-1] < 0:
        if on_overflow == "none":
            return None
        raise _lib.PaDTHipError("padt_mask_rle: capacity exceeded (n_counts %s)" % h["n_counts"].cpu().tolist())
    raw = h["packed"][: off[-1]].cpu().numpy().tobytes()
    out = [raw[off[i]: off[i + 1]].decode("ascii") for i in range(n)]
    if want_counts:
        nc = h["n_counts"].cpu().tolist()
        cc = h["counts"].cpu()
        return out, [cc[i, : nc[i]].tolist() for i in range(n)]
    return out


def mask_rle(bin_masks, dst_h, dst_w, want_counts=False):
    """COCO RLE `counts` strings of binarised masks, computed on the device (padt_mask_rle): bin_masks (n_obj, max_h, max_w) uint8 as
    mask_upsample_binarize returns them, dst_h / dst_w int32 device tensors (n_obj,).  → list of n_obj ASCII strings [, list of count lists].
    Two small device-to-host copies (the offset table, then exactly the string bytes) instead of the masks themselves."""
    r = mask_rle_fetch(mask_rle_launch(bin_masks, dst_h, dst_w), want_counts, on_overflow="none")
    if r is None:                                                     # noise-like masks beyond the bounded scratch: once more with the worst-case capacities
        r = mask_rle_fetch(mask_rle_launch(bin_masks, dst_h, dst_w, worst_case=True), want_counts)
    return r


def patchify_normalize(img_u8, lut, out, patch=14, merge=2, temporal=2):
    """img_u8 (H, W, 3) uint8 device tensor → rows of `out` ((H/patch)*(W/patch), 3*temporal*patch*patch), fp32 or bf16."""
    assert img_u8.dtype == torch.uint8 and img_u8.is_contiguous() and img_u8.shape[2] == 3
    assert lut.dtype == torch.float32 and lut.shape == (3, 256) and out.stride(1) == 1
    H, W = img_u8.shape[:2]
    _lib.check(_fn("padt_patchify_normalize", out.dtype)(_stream(), _p(img_u8), H, W, _p(lut), _p(out), out.stride(0),
                                                         1 if out.dtype in (BF16, F16) else 0, patch, merge, temporal), "padt_patchify_normalize")
    return out


# ------------------------------------------------------------------------------------------------ split-precision decoder ops
F32 = torch.float32
OUT_NONE, OUT_F32, OUT_SPLIT = 0, 1, 2


def gemm_hp(a_split, w2, bias=None, out=None, epilogue=EPI_NONE, residual=None, out_mode=OUT_F32, M=None, w_packed=None):
    """Split-precision GEMM (include/padt_hip.h "hp decoder"): a_split (M, 2K) bf16 rows [hi | lo], w2 (N, 2K) = [W | W].
    out_mode OUT_F32 → fp32 (M, N) [+ fp32 residual, in place allowed]; OUT_SPLIT → bf16 (M, 2N) rows [hi | lo]; epilogue EPI_SWIGLU (OUT_SPLIT only,
    weight rows [gate16 | up16]-interleaved) → bf16 (M, 2 * N/2) split rows of silu(gate) * up.
    w_packed (few rows only): the pack_weight() image of W — the rows are then packed too (one padt_pack_rows launch) so that every wave load of
    either operand is 1 KiB contiguous (padt_gemm_split_rows, layout 3)."""
    lib = _lib.load()
    _chk_bf16(a_split, w2, bias)
    M = a_split.shape[0] if M is None else M
    N, K2 = w2.shape
    assert a_split.shape[1] == K2, (a_split.shape, w2.shape)
    n_out = N // 2 if epilogue == EPI_SWIGLU else N
    if out_mode == OUT_F32:
        assert epilogue != EPI_SWIGLU
        if out is None:
            out = torch.empty((M, (N + 3) // 4 * 4), device=a_split.device, dtype=F32)
        assert out.dtype == F32 and out.stride(-1) == 1
        if residual is not None:
            assert residual.dtype == F32 and epilogue == EPI_RESID
        lo_off = 0
    else:
        assert residual is None and n_out % 4 == 0
        if out is None:
            out = torch.empty((M, 2 * n_out), device=a_split.device, dtype=BF16)
        assert out.dtype == BF16 and out.shape[1] >= 2 * n_out
        lo_off = n_out
    if M <= 64 and K2 % 16 == 0 and ((out_mode == OUT_F32 and epilogue in (EPI_NONE, EPI_RESID)) or epilogue == EPI_SWIGLU):
        # few rows (a decode step): bound by the weight stream — read W once and multiply it with the hi and the lo fragments (padt_gemm_split_rows)
        K = K2 // 2
        if w_packed is not None and K % 32 == 0 and N % 16 == 0 and a_split.stride(0) == K2:
            ap = torch.empty(((M + 15) // 16 * 16, K2), device=a_split.device, dtype=BF16)
            pack_rows(a_split, ap, M, to_packed=True)
            _lib.check(lib.padt_gemm_split_rows(_stream(), _p(ap), K2, 16 * K, _p(w_packed), w_packed.stride(0), _p(bias), _p(out), out.stride(0), lo_off,
                                                _p(residual), residual.stride(0) if residual is not None else 0, M, N, K, epilogue, 3), "padt_gemm_split_rows")
            return out
        _lib.check(lib.padt_gemm_split_rows(_stream(), _p(a_split), a_split.stride(0), K, _p(w2), w2.stride(0), _p(bias), _p(out), out.stride(0), lo_off,
                                            _p(residual), residual.stride(0) if residual is not None else 0, M, N, K, epilogue, 0), "padt_gemm_split_rows")
        return out
    if GEMM_LOG is not None:
        GEMM_LOG.append(("hp", a_split, w2, bias, out, epilogue, residual, out_mode, M))
    _tg_note("hp", M, N, K2 // 2)                                    # algorithmic K: the MFMA pipe executes 2K (hi and lo operands)
    _lib.check(lib.padt_gemm_bf16_ex(_stream(), _p(a_split), a_split.stride(0), _p(w2), w2.stride(0), _p(bias), _p(out), out.stride(0),
                                     _p(residual), residual.stride(0) if residual is not None else 0, M, N, K2, epilogue,
                                     1 if out_mode == OUT_F32 else 0, 0, 1 if residual is not None else 0, lo_off), "padt_gemm_bf16_ex")
    return out


def norm_split(x, w=None, eps=1e-6, act=0, idx=None, add=None, add_div=1, pos=None, y0_mode=OUT_SPLIT, y1_mode=OUT_NONE, chunk=None,
               rows=None, D=None):
    """→ (y0, y1) of padt_norm_split; x bf16 or fp32 (rows, >= D); modes OUT_NONE / OUT_F32 / OUT_SPLIT (split rows are 2*D wide)."""
    lib = _lib.load()
    assert x.dtype in (BF16, F32) and x.stride(-1) == 1
    D = x.shape[1] if D is None else D
    rows = (idx.numel() if idx is not None else x.shape[0]) if rows is None else rows
    chunk = D if chunk is None else chunk

    def mk(mode):
        if mode == OUT_NONE:
            return None
        return torch.empty((rows, D if mode == OUT_F32 else 2 * D), device=x.device, dtype=F32 if mode == OUT_F32 else BF16)
    y0, y1 = mk(y0_mode), mk(y1_mode)
    if add is not None:
        assert add.dtype == F32 and add.stride(-1) == 1
    if pos is not None:
        assert pos.dtype == F32 and pos.stride(-1) == 1
    if idx is not None:
        assert idx.dtype == torch.int32
    _lib.check(lib.padt_norm_split(_stream(), _p(x), x.stride(0), 1 if x.dtype == F32 else 0, _p(idx), _p(add),
                                   add.stride(0) if add is not None else 0, add_div, _p(w), float(eps), int(act), _p(pos),
                                   pos.stride(0) if pos is not None else 0, pos.shape[0] if pos is not None else 1,
                                   _p(y0), y0.stride(0) if y0 is not None else 0, y0_mode, _p(y1), y1.stride(0) if y1 is not None else 0,
                                   y1_mode, rows, D, chunk), "padt_norm_split")
    return y0, y1


def swiglu_split(gu, I, out=None):
    """fp32 rows [gate(I) | up(I)] → split rows [hi(I) | lo(I)] of silu(gate) * up (padt_swiglu_split)."""
    assert gu.dtype == F32 and gu.stride(-1) == 1 and gu.shape[1] >= 2 * I
    if out is None:
        out = torch.empty((gu.shape[0], 2 * I), device=gu.device, dtype=BF16)
    _lib.check(_lib.load().padt_swiglu_split(_stream(), _p(gu), gu.stride(0), int(I), _p(out), out.stride(0), gu.shape[0]), "padt_swiglu_split")
    return out


def layernorm_f32(x, w, b, eps=1e-5, out=None):
    """fp32 LayerNorm rows (bf16 weight / bias) → fp32 rows (padt_layernorm_f32)."""
    assert x.dtype == F32 and x.stride(-1) == 1
    _chk_bf16(w, b)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=F32)
    _lib.check(_lib.load().padt_layernorm_f32(_stream(), _p(x), x.stride(0), _p(w), _p(b), float(eps), _p(out), out.stride(0), x.shape[0], x.shape[1]),
               "padt_layernorm_f32")
    return out


def rope_half_f32_(x, cos, sin, n_heads, head_dim):
    """in place on the first n_heads*head_dim columns of the fp32 rows x (T, row)."""
    assert x.dtype == F32 and cos.dtype == F32 and sin.dtype == F32 and cos.stride(-1) == 1 and sin.stride() == cos.stride()
    _lib.check(_lib.load().padt_rope_half_f32(_stream(), _p(x), x.stride(0), _p(cos), _p(sin), cos.stride(0), x.shape[0], n_heads,
                                              head_dim), "padt_rope_half_f32")
    return x


def scatter_rows_f32(src, idx, dst, D=None):
    """dst[idx[i]] = src[i] for fp32 rows (distinct int32 indices)."""
    assert src.dtype == F32 and dst.dtype == F32 and idx.dtype == torch.int32 and src.stride(-1) == 1 and dst.stride(-1) == 1
    D = src.shape[1] if D is None else D
    _lib.check(_lib.load().padt_scatter_rows_f32(_stream(), _p(src), src.stride(0), _p(idx), _p(dst), dst.stride(0), idx.numel(), D), "padt_scatter_rows_f32")
    return dst


def attn_f32(q, k, v, cu_q, cu_k, max_q, max_k, n_heads, head_dim, scale=None, out=None, kv_group=1, causal=False, len_k=None, mfma=False):
    """fp32 varlen attention → split rows (Tq, 2*n_heads*head_dim) bf16.  kv_group: q heads per kv head (GQA); causal: bottom-right aligned
    mask; len_k (int32 device, per segment): key counts for segments that start at cu_k[s] and are not packed back to back (a KV cache)."""
    assert q.dtype == F32 and k.dtype == F32 and v.dtype == F32 and cu_q.dtype == torch.int32 and cu_k.dtype == torch.int32
    Dm = n_heads * head_dim
    if out is None:
        out = torch.empty((q.shape[0], 2 * Dm), device=q.device, dtype=BF16)
    scale = head_dim ** -0.5 if scale is None else scale
    fn = _lib.load().padt_attn_f32_mfma if mfma else _lib.load().padt_attn_f32     # mfma: v_mfma_f32_16x16x4_f32 (head_dim 80 / 128)
    _lib.check(fn(_stream(), _p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out), out.stride(0), Dm,
                                         _p(cu_q), _p(cu_k), cu_q.numel() - 1, int(max_q), int(max_k), n_heads, head_dim, float(scale), int(kv_group),
                                         1 if causal else 0, _p(len_k)),
               "padt_attn_f32")
    return out


def mask_scatter_f32(e2, mask_tok, cu_patch, obj_w, masks, n_obj, total_patches, dm):
    assert e2.dtype == F32 and mask_tok.dtype == F32 and masks.dtype == F32 and masks.is_contiguous()
    _lib.check(_lib.load().padt_mask_scatter_f32(_stream(), _p(e2), e2.stride(0), _p(mask_tok), mask_tok.stride(0), _p(cu_patch), _p(obj_w),
                                                 _p(masks), n_obj, total_patches, masks.shape[1], masks.shape[2], dm), "padt_mask_scatter_f32")
    return masks


def resample_pass_u8(img, out, bounds, kk, horizontal):
    """One pass of Pillow's 8-bit resample (csrc/resize.hip): img / out (H, W, C) uint8 device tensors, bounds (n, 2) / kk (n, ksize) int32."""
    assert img.dtype == torch.uint8 and out.dtype == torch.uint8 and img.is_contiguous() and out.is_contiguous()
    assert bounds.dtype == torch.int32 and kk.dtype == torch.int32 and bounds.is_contiguous() and kk.is_contiguous()
    n = out.shape[1] if horizontal else out.shape[0]
    assert bounds.shape == (n, 2) and kk.shape[0] == n and img.shape[2] == out.shape[2]
    _lib.check(_lib.load().padt_resample_pass_u8(_stream(), _p(img), img.shape[0], img.shape[1], img.shape[2], _p(out), out.shape[0],
                                                 out.shape[1], _p(bounds), _p(kk), kk.shape[1], 1 if horizontal else 0), "padt_resample_pass_u8")
    return out


def pack_results(out, n, cap, mask_hw, sample_idx, valid_h, valid_w, boxes, scores, masks):
    """One batch's vl_decode output → the fixed-capacity int32 exchange record `out` (padt_pack_results), on the current stream."""
    assert out.dtype == torch.int32 and out.is_contiguous() and out.is_cuda
    H, Wd = (masks.shape[1], masks.shape[2]) if masks is not None else (0, 0)
    if n:
        assert sample_idx.dtype == torch.int32 and boxes.dtype == torch.float32 and scores.dtype == torch.float32
        assert boxes.stride(-1) == 1 and (masks is None or (masks.dtype == torch.float32 and masks.stride(-1) == 1))
        assert masks is None or (valid_h.dtype == torch.int64 and valid_w.dtype == torch.int64)
    _lib.check(_lib.load().padt_pack_results(_stream(), _p(out), out.numel(), int(n), int(cap), int(mask_hw), _p(sample_idx), _p(valid_h), _p(valid_w),
                                             _p(boxes), boxes.stride(0) if n else 0, _p(scores), scores.stride(0) if n else 0,
                                             _p(masks) if n else 0, masks.stride(0) if masks is not None else 0,
                                             masks.stride(1) if masks is not None else 0, int(H), int(Wd)), "padt_pack_results")
    return out


def memset(t, value=0):
    lib = _lib.load()
    _lib.check(lib.padt_memset(_stream(), _p(t), value, t.numel() * t.element_size()), "padt_memset")
    return t
