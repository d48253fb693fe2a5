"""Per-kernel parity of the fp16-operand instantiation (include/padt_hip_f16.h): every `_f16` entry point vs a plain PyTorch fp32 statement
of the same op on the same fp16 inputs — the fp16 twins of tests/test_kernels_gpu.py, which keeps covering the bf16 instantiation.

Tolerance: an fp16 result must be within one fp16 rounding of the fp32 reference, |out - ref| <= ulps * (2^-11 |ref| + 2.5e-4 * rms(ref))
(ulps = 1; attention 6: the probabilities are rounded to fp16 before the P·V MFMA).  The 16-bit mirror of an fp32 residual stream is EXACTLY
fp16(stream_scale * x32), and its consumers return rstd / stream_scale when given eps * stream_scale^2 — asserted bit for bit where the
arithmetic allows, else at fp32 accuracy.
"""
import pytest
import torch

from test_kernels_gpu import close_f32, interleave_gate_up, ref_attn

pytestmark = pytest.mark.gpu

H = torch.float16
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops as _ops
    return _ops


@pytest.fixture
def knobs(ops):
    yield ops.gemm_knobs
    ops.gemm_knobs(mode256=1, mf=0, peel=1, colsplit=1, group_m=8)


def rnd(*shape, scale=1.0, seed=0, dt=H):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dt).cuda()


def close_f16(out, ref, what="", ulps=1.0):
    assert out.dtype == H, (what, out.dtype)
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    err = (out - ref).abs()
    lim = (ref.abs() * 2 ** -11 + 2.5e-4 * rms) * ulps
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} outside tolerance, max err {err.max().item():.3e} (rms {rms:.3e})"


def test_operand_types_do_not_mix(ops):
    a, w = rnd(32, 64), rnd(48, 64, dt=BF)
    with pytest.raises(AssertionError, match="mixed 16-bit operand types"):
        ops.gemm(a, w)
    assert ops.stream_scale(H) == 2.0 ** -4 and ops.stream_scale(BF) == 1.0
    assert ops.mirror_eps(1e-6, H) == 1e-6 * 2.0 ** -8 and ops.mirror_eps(1e-6, BF) == 1e-6


# ------------------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 136), (2116, 3840, 1280), (1000, 1280, 3456), (577, 2048, 1176), (8, 2048, 2048), (50, 4, 1280),
                                   (16, 2560, 2048)])
def test_gemm_f16_plain_bias_and_f32_out(ops, M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2), rnd(N, seed=3)
    ref = a.float() @ w.float().T + b.float()
    n_pad = (N + 3) // 4 * 4
    out = torch.full((M, n_pad), 7.0, device="cuda", dtype=H)
    ops.gemm(a, w, b, out=out)
    close_f16(out[:, :N], ref, f"gemm {M}x{N}x{K}")
    if n_pad != N:
        assert (out[:, N:] == 7.0).all(), "wrote outside N"
    out32 = torch.zeros((M, n_pad), device="cuda", dtype=torch.float32)
    ops.gemm(a, w, None, out=out32, out_f32=True)
    close_f32(out32[:, :N], a.float() @ w.float().T, f"gemm f32 {M}x{N}x{K}", rel=2e-5)
    # the same numbers through both instantiations (operands exactly representable in both types): same tiles, same K order, exact
    # products, fp32 accumulation — the fp32 outputs agree to the MFMA's internal summation
    def common(t):
        t = t.float().to(BF).float()
        return torch.where(t.abs() < 2.0 ** -12, torch.zeros_like(t), t)
    ac, wc = common(a), common(w)
    assert torch.equal(ac.to(H).float(), ac) and torch.equal(wc.to(BF).float(), wc)
    o1, o2 = torch.zeros_like(out32), torch.zeros_like(out32)
    ops.gemm(ac.to(H), wc.to(H), None, out=o1, out_f32=True)
    ops.gemm(ac.to(BF), wc.to(BF), None, out=o2, out_f32=True)
    close_f32(o1[:, :N], o2[:, :N], "bf16 vs fp16 instantiation on operands both types hold exactly", rel=2e-6)


@pytest.mark.parametrize("mf", [2, 3, 4])
def test_gemm_f16_tile256_heights_and_epilogues(ops, mf, knobs):
    knobs(mode256=2, mf=mf)
    M, N, K = 1000, 1280, 1280
    a, w, b, r = rnd(M, K, seed=4), rnd(N, K, scale=0.05, seed=5), rnd(N, seed=6), rnd(M, N, seed=7)
    lin = a.float() @ w.float().T + b.float()
    close_f16(ops.gemm(a, w, b), lin, f"tile256 mf={mf}")
    close_f16(ops.gemm(a, w, b, epilogue=ops.EPI_GELU), torch.nn.functional.gelu(lin), "GELU")
    close_f16(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, residual=r), lin + r.float(), "RESID")
    wg, wu = rnd(N // 2, K, scale=0.05, seed=8), rnd(N // 2, K, scale=0.05, seed=9)
    bi = rnd(N, seed=10)
    y = (a.float() @ interleave_gate_up(wg, wu).float().T + bi.float()).view(M, N // 32, 2, 16)
    close_f16(ops.gemm(a, interleave_gate_up(wg, wu), bi, epilogue=ops.EPI_SWIGLU),
              (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, N // 2), "SwiGLU")


@pytest.mark.parametrize("M,N,K", [(1000, 1280, 1280), (40, 640, 512), (300, 200, 136)])
def test_gemm_f16_row_scale_consumes_a_scaled_mirror(ops, M, N, K):
    """The folded RMSNorm over a stream MIRROR: row_rstd(mirror, eps * s^2) = rstd(x) / s, and rstd' * ((s x) W^T) = rstd * (x W^T)."""
    s = ops.stream_scale(H)
    x32 = torch.randn(M, K, device="cuda", generator=torch.Generator(device="cuda").manual_seed(11)) * 5
    mirror = ops.cast_f32_x16(x32, dtype=H, scale=s)
    assert torch.equal(mirror, (x32 * s).to(H))
    rstd = ops.row_rstd(mirror, eps=ops.mirror_eps(1e-6, H))
    xm = mirror.float() / s                                              # what the mirror holds of x
    ref_rstd = torch.rsqrt(xm.pow(2).mean(-1) + 1e-6)
    close_f32(rstd * s, ref_rstd, "row_rstd of a scaled mirror", rel=1e-5)
    w, b = rnd(N, K, scale=0.05, seed=12), rnd(N, seed=13)
    out = ops.gemm(mirror, w, b, row_scale=rstd)
    close_f16(out, (xm * ref_rstd[:, None]) @ w.float().T + b.float(), "gemm(row_scale) over the mirror")


@pytest.mark.parametrize("M,K", [(1000, 640), (40, 256), (300, 136)])
def test_gemm_rope_f16(ops, M, K, knobs):
    Hh, D = 4, 80
    N = 3 * Hh * D
    from padt_amd.weights import interleave_rope_rows
    a, w, b = rnd(M, K, seed=14), rnd(N, K, scale=0.05, seed=15), rnd(N, seed=16)
    ang = torch.rand(M, D // 2, generator=torch.Generator().manual_seed(17)).cuda() * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    lin = (a.float() @ w.float().T + b.float()).view(M, 3, Hh, D)
    c, s_ = torch.cat([cos, cos], -1)[:, None, None, :], torch.cat([sin, sin], -1)[:, None, None, :]
    qk = lin[:, :2]
    rot = torch.cat([-qk[..., D // 2:], qk[..., : D // 2]], -1)
    ref = torch.cat([(qk * c + rot * s_), lin[:, 2:]], 1).reshape(M, N)
    out = torch.zeros(M, N, device="cuda", dtype=H)
    ops.gemm_rope(a, interleave_rope_rows(w, 2 * Hh, D), interleave_rope_rows(b, 2 * Hh, D), out, cos, sin, 2 * Hh * D, D)
    # undo the pair interleave of the q / k columns
    idx = torch.arange(2 * Hh * D).view(2 * Hh, 2, D // 2).transpose(1, 2).reshape(-1).cuda()
    un = out.clone()
    un[:, idx] = out[:, : 2 * Hh * D]
    close_f16(un, ref, f"gemm_rope {M}x{K}")


@pytest.mark.parametrize("M,N,K", [(1000, 1280, 1280), (577, 2048, 2048), (300, 200, 136), (40, 640, 512), (2 * 256 + 24, 768, 640)])
def test_gemm_resid32_f16_stream_and_scaled_mirror(ops, M, N, K, knobs):
    if M == 2 * 256 + 24:
        knobs(peel=2, mf=4)
    s = ops.stream_scale(H)
    a, w, b = rnd(M, K, seed=91), rnd(N, K, scale=0.05, seed=92), rnd(N, seed=93)
    x0 = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(94)) * 3
    x0[0, 0] = 3.0e5                                                     # a "massive activation": beyond fp16's 65504, inside the scaled mirror's range
    ld = (N + 7) // 8 * 8
    x32 = torch.zeros(M, ld, device="cuda")
    x32[:, :N] = x0
    xb = torch.full((M, ld), 7.0, device="cuda", dtype=H)
    ops.gemm_resid32(a, w, b, x32[:, :N], xb[:, :N])
    ref = x0 + a.float() @ w.float().T + b.float()
    assert (x32[:, :N] - ref).abs().max().item() <= 3e-5 * ref[1:].pow(2).mean().sqrt().item() + 0.05      # 0.05: one fp32 ulp at 3e5
    assert torch.equal(xb[:, :N], (x32[:, :N] * s).to(H)), "mirror != fp16(stream_scale * stream)"
    assert torch.isfinite(xb[:, :N].float()).all()
    if ld != N:
        assert (xb[:, N:] == 7.0).all() and (x32[:, N:] == 0).all(), "wrote outside N"


# ------------------------------------------------------------------------------------------------------------ decode-step projections
@pytest.mark.parametrize("M,N,K", [(8, 2560, 2048), (8, 22016, 2048), (64, 2048, 11008), (21, 704, 512), (128, 3584, 3584)])
def test_gemm_packed_f16_fused_norm_resid32_and_fp8(ops, M, N, K):
    s = ops.stream_scale(H)
    x32 = torch.randn(M, K, device="cuda", generator=torch.Generator(device="cuda").manual_seed(21)) * 2
    w = rnd(N, K, scale=0.05, seed=22)
    wp = ops.pack_weight(w)
    M16 = (M + 15) // 16 * 16
    mirror = ops.cast_f32_x16(x32, dtype=H, scale=s)
    xp = torch.zeros(M16, K, device="cuda", dtype=H)
    ops.pack_rows(mirror, xp, M, to_packed=True)
    xm = mirror.float() / s
    n = xm * torch.rsqrt(xm.pow(2).mean(-1, keepdim=True) + 1e-6)
    out = torch.zeros(M, N, device="cuda", dtype=H)
    ops.gemm_packed(xp, wp, N, out=out, norm_eps=ops.mirror_eps(1e-6, H), a_packed=True, rows=M)
    close_f16(out, n @ w.float().T, f"packed fused norm {M}x{N}x{K}", ulps=1.5)
    rm = ops.gemm_rmsnorm(mirror, w, eps=ops.mirror_eps(1e-6, H)) if M <= 64 else None
    if rm is not None:
        close_f16(rm, n @ w.float().T, "gemm_rmsnorm", ulps=1.5)
    if N % 32 == 0:
        o2 = torch.zeros(M16, N // 2, device="cuda", dtype=H)
        ops.gemm_packed(xp, wp, N, out=o2, epilogue=ops.EPI_SWIGLU, norm_eps=ops.mirror_eps(1e-6, H), a_packed=True, c_packed=True, rows=M)
        un = torch.zeros(M, N // 2, device="cuda", dtype=H)
        ops.pack_rows(o2, un, M, to_packed=False)
        y = (n @ w.float().T).view(M, N // 32, 2, 16)
        close_f16(un, (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, N // 2), "packed SwiGLU", ulps=1.5)
    # residual projection over the fp32 stream: mirror == fp16(scale * stream), fp16 and fp8 weights, split-K
    a = rnd(M, K, seed=23)
    ap = torch.zeros(M16, K, device="cuda", dtype=H)
    ops.pack_rows(a, ap, M, to_packed=True)
    r = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(24))
    ws = ops.new_splitk_workspace(N, 2, "cuda")
    for split in (1, 2):
        st = r.clone()
        mir = torch.zeros(M16, N, device="cuda", dtype=H)
        ops.gemm_packed_resid32(ap, wp, N, st, mir, split_k=split, workspace=ws, rows=M)
        close_f32(st, r + a.float() @ w.float().T, f"packed resid32 split {split}", rel=3e-5)
        un = torch.zeros(M, N, device="cuda", dtype=H)
        ops.pack_rows(mir, un, M, to_packed=False)
        assert torch.equal(un, (st * s).to(H)), "packed mirror != fp16(stream_scale * stream)"
    q, sc, deq = ops.quantize_fp8_rows(w, deq_dtype=H)
    assert torch.equal(deq.float(), q.view(torch.float8_e4m3fn).float() * sc[:, None]), "scale * e4m3 is exact in fp16"
    if N % 16 == 0:
        o8 = torch.zeros(M, N, device="cuda", dtype=H)
        ops.gemm_packed_fp8(xp, ops.pack_weight_fp8(q), sc, N, out=o8, norm_eps=ops.mirror_eps(1e-6, H), a_packed=True, rows=M)
        close_f16(o8, n @ deq.float().T, "packed fp8 weights", ulps=1.5)
        st = r.clone()
        mir = torch.zeros(M16, N, device="cuda", dtype=H)
        ops.gemm_packed_resid32(ap, ops.pack_weight_fp8(q), N, st, mir, scales=sc, rows=M)
        close_f32(st, r + a.float() @ deq.float().T, "packed resid32 fp8", rel=3e-5)


@pytest.mark.parametrize("rows", [8, 16, 32, 64])
def test_decode_projection_f16_rows_do_not_depend_on_the_batch(ops, rows):
    """In-flight batching must not change a sample's bits on fp16 operands either (the K-step → wave map is the bf16 instantiation's)."""
    N, K = 2560, 2048
    x, w = rnd(64, K, seed=25), rnd(N, K, scale=0.05, seed=26)
    wp = ops.pack_weight(w)
    res = []
    for m in (rows, 64):
        m16 = (m + 15) // 16 * 16
        xp = torch.zeros(m16, K, device="cuda", dtype=H)
        ops.pack_rows(x[:m].contiguous(), xp, m, to_packed=True)
        out = torch.zeros(m, N, device="cuda", dtype=H)
        ops.gemm_packed(xp, wp, N, out=out, norm_eps=1e-6, a_packed=True, rows=m)
        res.append(out[:8].clone())
    assert torch.equal(res[0], res[1])


# ------------------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("D,Hh,Hkv,lens,causal", [
    (80, 16, 16, [64] * 5 + [48] * 3 + [36], False), (80, 16, 16, [2116], False), (128, 16, 2, [577, 100, 1, 65], True),
    (128, 28, 4, [577, 33, 7], True), (32, 4, 2, [10, 130, 64], True)])
def test_attn_varlen_f16(ops, D, Hh, Hkv, lens, causal):
    T = sum(lens)
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    qkv = rnd(T, (Hh + 2 * Hkv) * D, seed=20)
    q, k, v = qkv[:, : Hh * D], qkv[:, Hh * D: (Hh + Hkv) * D], qkv[:, (Hh + Hkv) * D:]
    out = torch.zeros(T, Hh * D, device="cuda", dtype=H)
    cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
    ops.attn_varlen(q, k, v, out, cu_t, cu_t, max(lens), Hh, Hkv, D, causal=causal)
    close_f16(out, ref_attn(q, k, v, cu, cu, Hh, Hkv, D, causal), f"attn D={D} lens={lens[:3]}..", ulps=6)


@pytest.mark.parametrize("D,Hq,Hkv,sec", [(128, 16, 2, (16, 24, 24)), (32, 4, 2, (4, 6, 6)), (128, 28, 4, (16, 24, 24))])
def test_decode_attn_rope_f16_matches_unfused_pipeline(ops, D, Hq, Hkv, sec):
    B, S_max = 3, 1344
    slots = [577, 63, 1290]
    qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=33)
    kc = rnd(B, Hkv, S_max, D, seed=34)
    v = rnd(B, Hkv, S_max, D, seed=35)
    vt = v.transpose(2, 3).contiguous()
    slot_t = torch.tensor(slots, dtype=torch.int32, device="cuda")
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
    for seed in range(1, 25):                                        # prompt path and decode path rotate a token bit-identically
        gpos = torch.randint(0, 4000, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(seed)).cuda()
        qa, qb = torch.zeros(B, Hq * D, device="cuda", dtype=H), torch.zeros(B, Hq * D, device="cuda", dtype=H)
        ka, va, kb, vb = kc.clone(), vt.clone(), kc.clone(), vt.clone()
        csx = torch.zeros(B, D // 2, 2, device="cuda")
        ops.rope_table(gpos, inv, csx, D, sec)
        ops.decode_attn_rope(qkv, csx, slot_t, ka, va, qa, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max)
        ops.llm_qkv_post(qkv, gpos, inv, qb, kb, vb, Hq, Hkv, D, S_max, sec, slot=slot_t)
        assert torch.equal(ka, kb) and torch.equal(va, vb), f"rotated K / V differ between the decode and the prompt path (seed {seed})"
        ob = torch.zeros_like(qa)
        ops.decode_attn(qb, kb, vb, slot_t + 1, ob, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, max(slots) + 1)
        # the fused kernel against the unfused pair: bit-identical, as in the bf16 instantiation (test_kernels_gpu.py).  Until the end of round 4
        # a fraction of a per cent of the fp16 outputs differed at D = 128: hipcc had folded the fused kernel's scalar rotation + conversion
        # into one v_fma_mixlo_f16 (a single rounding) while llm_qkv_post's vector path rounds twice — one q element in ~2^13 sat on an fp16
        # tie (tools/diag/decode_attn_paths.py; rounded32() in csrc/common.h)
        assert torch.equal(qa, ob), f"attention differs between the two paths (seed {seed}): {int((qa != ob).sum())} of {qa.numel()} outputs"
    rep = Hq // Hkv
    ref = torch.zeros(B, Hq * D, device="cuda")
    for b in range(B):
        L = slots[b] + 1
        kk = kb[b, :, :L].float().repeat_interleave(rep, 0)
        vv = vb[b, :, :, :L].float().transpose(1, 2).repeat_interleave(rep, 0)
        sc = torch.einsum("hd,hld->hl", qb[b].float().view(Hq, D), kk) * D ** -0.5
        ref[b] = torch.einsum("hl,hld->hd", torch.softmax(sc, -1), vv).reshape(-1)
    close_f16(qa, ref, "decode attn + rope", ulps=6)


# ------------------------------------------------------------------------------------------------------------ row kernels
def test_row_kernels_f16(ops):
    x, w, b = rnd(77, 1280, seed=40), rnd(1280, seed=41) * 0.1 + 1, rnd(1280, seed=42)
    xf = x.float()
    close_f16(ops.rmsnorm(x, w), xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float(), "rmsnorm")
    close_f16(ops.layernorm(x, w, b), torch.nn.functional.layer_norm(xf, (1280,), w.float(), b.float(), 1e-5), "layernorm")
    close_f16(ops.add_rows(x, x), xf + xf, "add_rows")
    assert torch.equal(ops.cast_x16_f32(x), xf)
    x32 = torch.randn(77, 1280, device="cuda") * 2
    assert torch.equal(ops.cast_f32_x16(x32, dtype=H), x32.to(H))
    assert torch.equal(ops.cast_f32_x16(x32, D_pad=1288, dtype=H)[:, 1280:], torch.zeros(77, 8, device="cuda", dtype=H))
    close_f16(ops.rmsnorm_f32(x32, w), x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float(), "rmsnorm_f32")
    # rotate-half in place == the fp32 statement, and gather / pack / embed move fp16 words untouched
    T, nh, D = 50, 6, 80
    q = rnd(T, nh * D, seed=43)
    ang = torch.rand(T, D // 2, generator=torch.Generator().manual_seed(44)).cuda() * 6.28
    qf = q.float().view(T, nh, D)
    c, s = torch.cat([ang.cos(), ang.cos()], -1)[:, None], torch.cat([ang.sin(), ang.sin()], -1)[:, None]
    ref = (qf * c + torch.cat([-qf[..., D // 2:], qf[..., : D // 2]], -1) * s).reshape(T, nh * D)
    close_f16(ops.rope_half_(q.clone(), ang.cos().contiguous(), ang.sin().contiguous(), nh, D), ref, "rope_half")
    idx = torch.randperm(77, generator=torch.Generator().manual_seed(45)).to(torch.int32).cuda()
    assert torch.equal(ops.gather_rows(x, idx), x[idx.long()])
    ids = torch.tensor([3, 76, 80, 5], dtype=torch.int64, device="cuda")
    proto = rnd(8, 1280, seed=46)
    assert torch.equal(ops.embed_tokens(ids, None, x, proto, None), torch.stack([x[3], x[76], proto[3], x[5]]))


@pytest.mark.parametrize("B", [8, 40])
def test_vrt_head_f16(ops, B):
    V, D, per = 3008, 256, 23
    NP = B * per
    E, P, h = rnd(V, D, seed=95), rnd(NP, D, seed=96), rnd(B, D, seed=97)
    off = torch.arange(0, NP + 1, per, dtype=torch.int32, device="cuda")
    nblk = ops.vrt_head_nblk(V, NP)
    B16 = (B + 15) // 16 * 16
    hp = torch.zeros(B16, D, device="cuda", dtype=H)
    ops.pack_rows(h, hp, B, to_packed=True)
    Ep = ops.pack_weight(E)
    res = []
    for packed in (False, True):
        pv = torch.zeros(nblk * B, device="cuda")
        pi = torch.zeros(nblk * B, dtype=torch.int32, device="cuda")
        lg = torch.zeros(B, V + NP, device="cuda")
        if packed:
            ops.vrt_head(hp, E, P, off, pv, pi, 7, logits=lg, table_packed=Ep, rows=B)
        else:
            ops.vrt_head(h, E, P, off, pv, pi, 7, logits=lg)
        res.append((pv, pi, lg))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    full = h.float() @ torch.cat([E, P]).float().T
    fin = torch.isfinite(res[1][2])
    close_f32(res[1][2][fin], full[fin], "head logits", rel=2e-5)
    # greedy bookkeeping copies the fp16 hidden row untouched
    hbuf = torch.zeros(2, B, D, device="cuda", dtype=H)
    z = lambda *s, dt=torch.int32: torch.zeros(*s, dtype=dt, device="cuda")
    ops.greedy_step(res[0][0], res[0][1], nblk, h, hbuf, torch.ones(B, dtype=torch.int32, device="cuda"), z(B, 2, dt=torch.int64),
                    z(B, dt=torch.int64), z(1), z(B), z(B), z(3, B), 7, 3)
    assert torch.equal(hbuf[0], h)


# ------------------------------------------------------------------------------------------------------------ fp8 x fp8 prompt pass
@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (577, 4608, 3584)])
def test_quant_rows_and_gemm_fp8_f16(ops, M, N, K):
    s = ops.stream_scale(H)
    x32 = torch.randn(M, K, device="cuda", generator=torch.Generator(device="cuda").manual_seed(50)) * 3
    mirror = ops.cast_f32_x16(x32, dtype=H, scale=s)
    x8, rs = ops.quant_rows_fp8(mirror, norm_eps=ops.mirror_eps(1e-6, H))
    xm = mirror.float()
    amax = xm.abs().amax(-1, keepdim=True)
    qs = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
    assert torch.equal(x8.view(torch.float8_e4m3fn).float(), (xm / qs).to(torch.float8_e4m3fn).float()), "e4m3 codes"
    close_f32(rs, (qs * torch.rsqrt(xm.pow(2).mean(-1, keepdim=True) + 1e-6 * s * s)).reshape(-1), "row scale x rstd of the mirror", rel=1e-5)
    w = rnd(N, K, scale=0.05, seed=51)
    q, sc, deq = ops.quantize_fp8_rows(w, deq_dtype=H)
    b = rnd(N, seed=52)
    out = ops.gemm_fp8(x8, q.contiguous(), sc, rs, bias=b, out_dtype=H)
    a_deq = x8.view(torch.float8_e4m3fn).float() * rs[:, None]
    ref = a_deq @ deq.float().T + b.float()
    close_f16(out, ref, f"gemm_fp8 {M}x{N}x{K}", ulps=2)
    st = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(53))
    st0 = st.clone()
    xb = torch.zeros(M, N, device="cuda", dtype=H)
    a8, rs2 = ops.quant_rows_fp8(rnd(M, K, seed=54))
    ops.gemm_fp8(a8, q.contiguous(), sc, rs2, epilogue=ops.EPI_RESID, x32=st, xb=xb)
    # the fp8 MFMA does not sum its 128 products in full fp32 (≈2^-15 of a dot product's magnitude sum: test_kernels_gpu.py's fp8 test)
    close_f32(st, st0 + (a8.view(torch.float8_e4m3fn).float() * rs2[:, None]) @ deq.float().T, "gemm_fp8 resid", rel=4e-4)
    assert torch.equal(xb, (st * s).to(H)), "fp8 GEMM's mirror != fp16(stream_scale * stream)"


def test_patchify_normalize_f16(ops):
    from padt_amd.preprocess import normalize_lut
    img = torch.randint(0, 256, (56, 84, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(60)).cuda()
    lut = torch.from_numpy(normalize_lut()).cuda()
    o32 = torch.zeros(24, 1176, device="cuda")
    o16 = torch.zeros(24, 1176, device="cuda", dtype=H)
    ops.patchify_normalize(img, lut, o32)
    ops.patchify_normalize(img, lut, o16)
    assert torch.equal(o16, o32.to(H))
