"""Split-precision ("hp") decoder kernels (csrc/decoder_hp.hip + the padt_gemm_bf16_ex epilogues) against plain PyTorch fp32 / fp64
statements of the same ops.  Tolerances: the (hi, lo) bf16 pair carries 16 mantissa bits → 2^-16 relative per element on the way
into a GEMM; everything else is fp32 arithmetic.  Written at each assert."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops as _ops
    return _ops


def rndf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).cuda()


def join(split, D, chunk=None):
    """(rows, 2D) split rows [hi(chunk) lo(chunk)]... → fp32 (rows, D) = hi + lo."""
    chunk = D if chunk is None else chunk
    r = split.float().view(split.shape[0], D // chunk, 2, chunk)
    return (r[:, :, 0] + r[:, :, 1]).reshape(split.shape[0], D)


def split_ref(x):
    hi = x.to(BF)
    lo = (x - hi.float()).to(BF)
    return torch.cat([hi, lo], dim=1)


def rel_err(out, ref):
    return ((out.double() - ref.double()).abs().max() / (ref.double().pow(2).mean().sqrt() + 1e-30)).item()


def split_close(got, ref, what="", noise=2e-6):
    """a value read back from a (hi, lo) pair: |err| <= 2^-17 |x| (hi: half an ulp of 8 bits, lo: half an ulp of the remainder) —
    asserted at 2^-16 |x| + fp32 noise (`noise` x the rms: the accumulation error of the fp32 sums themselves)."""
    err = (got.double() - ref.double()).abs()
    lim = ref.double().abs() * 2.0 ** -16 + noise * ref.double().pow(2).mean().sqrt()
    assert bool((err <= lim).all()), f"{what}: max err {err.max().item():.3e}, worst ratio {(err / lim).max().item():.2f}"


@pytest.mark.parametrize("x_f32", [True, False])
def test_norm_split_all_stages(ops, x_f32):
    rows, D, src_rows = 37, 1280, 50
    x = rndf(src_rows, D, seed=1)
    if not x_f32:
        x = x.to(BF)
    idx = torch.randint(0, src_rows, (rows,), generator=torch.Generator().manual_seed(3)).to(torch.int32).cuda()
    add = rndf((rows + 3) // 4, D, seed=2)
    w = (1 + 0.1 * rndf(D, seed=4)).to(BF)
    pos = rndf(5, D, seed=5)
    y0, y1 = ops.norm_split(x, w, eps=1e-6, act=1, idx=idx, add=add, add_div=4, pos=pos, y0_mode=ops.OUT_F32, y1_mode=ops.OUT_SPLIT, chunk=320)
    xs = x.float()[idx.long()] + add[torch.arange(rows, device="cuda") // 4]
    ref = xs * torch.rsqrt(xs.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
    ref = torch.nn.functional.gelu(ref)
    assert rel_err(y0, ref) < 2e-6                                # fp32 arithmetic
    ref1 = ref + pos[torch.arange(rows, device="cuda") % 5]
    split_close(join(y1, D, 320), ref1, "norm_split y1")
    # pass-through (no norm): exact split of the input
    s0, _ = ops.norm_split(x if x_f32 else x.float())
    assert torch.equal(s0, split_ref(x.float()))


@pytest.mark.parametrize("M,N,K,epi", [(64, 1280, 1280, 0), (20, 1280, 2048, 1), (700, 640, 1280, 1), (2116, 2560, 1280, 0), (9, 4, 1280, 0),
                                        (300, 320, 320, 1), (1000, 1280, 3456, 2), (40, 1280, 1280, 2)])
def test_gemm_hp_against_fp64(ops, M, N, K, epi):
    """A = fp32 activations as (hi, lo) pairs, W bf16 doubled along K: result within 16-mantissa-bit input rounding of the exact
    product — three orders of magnitude tighter than the plain bf16 GEMM's 2^-9."""
    a = rndf(M, K, seed=10)
    w = rndf(N, K, scale=0.05, seed=11).to(BF)
    b = rndf(N, scale=0.1, seed=12).to(BF)
    w2 = torch.cat([w, w], dim=1).contiguous()
    a_s, _ = ops.norm_split(a)
    ref = a.double() @ w.double().T + b.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    wpk = ops.pack_weight(w) if (M <= 64 and N % 16 == 0 and K % 32 == 0) else None      # few rows: also through the packed operand layouts (layout 3)
    if epi == 2:
        r = rndf(M, N, seed=13)
        out = r.clone()
        ops.gemm_hp(a_s, w2, b, out=out, epilogue=ops.EPI_RESID, residual=out)      # in place, fp32 residual stream
        assert rel_err(out, ref + r.double()) < 5e-5
        if wpk is not None:
            o2 = r.clone()
            ops.gemm_hp(a_s, w2, b, out=o2, epilogue=ops.EPI_RESID, residual=o2, w_packed=wpk)
            assert torch.equal(o2, out), "packed operands change the few-row projection's bits"
        return
    out = ops.gemm_hp(a_s, w2, b, epilogue=epi)
    assert out.dtype == F32 and rel_err(out[:, :N], ref) < 5e-5
    if wpk is not None and epi == 0:
        assert torch.equal(ops.gemm_hp(a_s, w2, b, epilogue=epi, w_packed=wpk), out), "packed operands change the few-row projection's bits"
    if N % 4 == 0:
        sp = ops.gemm_hp(a_s, w2, b, epilogue=epi, out_mode=ops.OUT_SPLIT)
        assert sp.shape == (M, 2 * N) and rel_err(join(sp, N), ref) < 8e-5
        if M > 64:
            assert torch.equal(sp[:, :N], out[:, :N].to(BF))       # hi half = the bf16 rounding of the fp32 result
        else:
            # few rows, fp32 output: padt_gemm_split_rows (W read once, hi / lo products interleaved per K-step) — another fp32 summation
            # order than the K' = 2K loop that writes the split output: the two agree to fp32 noise, i.e. within one bf16 step after rounding
            d = (sp[:, :N].float() - out[:, :N].to(BF).float()).abs()
            assert bool((d <= out[:, :N].abs() * 2.0 ** -7 + 1e-6).all())


@pytest.mark.parametrize("M,I,K,bias", [(40, 1088, 2048, False), (64, 320, 1280, True), (9, 64, 320, True),        # few rows: padt_gemm_split_rows (W read once)
                                         (700, 320, 1280, True), (2116, 3456, 1280, True), (577, 1088, 2048, False),   # 256-row tile kernel: ragged / full column tiles
                                         (300, 160, 320, True)])                                                      # 128^2 tile kernel
def test_gemm_hp_split_swiglu_against_fp64(ops, M, I, K, bias):
    """Round 6 (precision="reference"): the gate/up projection with the SwiGLU as its epilogue — weight rows [gate16 | up16]-interleaved, silu(gate) * up
    evaluated on the fp32 accumulators (exact expf / division) and written as a (hi, lo) pair: the pair carries 16 mantissa bits, the GEMM inputs too
    → within 2^-16-class rounding of the fp64 statement, and equal to the unfused fp32 gate / up rows + padt_swiglu_split up to fp32 noise."""
    from padt_amd.weights import interleave16
    a = rndf(M, K, seed=20)
    wg, wu = rndf(I, K, scale=0.05, seed=21).to(BF), rndf(I, K, scale=0.05, seed=22).to(BF)
    bg, bu = rndf(I, scale=0.1, seed=23).to(BF), rndf(I, scale=0.1, seed=24).to(BF)
    w2 = torch.cat([interleave16(wg, wu)] * 2, dim=1).contiguous()
    b = interleave16(bg, bu).contiguous() if bias else None
    a_s, _ = ops.norm_split(a)
    g = a.double() @ wg.double().T + (bg.double() if bias else 0.0)
    u = a.double() @ wu.double().T + (bu.double() if bias else 0.0)
    ref = torch.nn.functional.silu(g) * u
    h = ops.gemm_hp(a_s, w2, b, epilogue=ops.EPI_SWIGLU, out_mode=ops.OUT_SPLIT)
    assert h.dtype == BF and h.shape == (M, 2 * I)
    if M <= 64 and K % 32 == 0:
        hp_ = ops.gemm_hp(a_s, w2, b, epilogue=ops.EPI_SWIGLU, out_mode=ops.OUT_SPLIT, w_packed=ops.pack_weight(interleave16(wg, wu)))
        assert torch.equal(hp_, h), "packed operands change the few-row split SwiGLU's bits"
    assert rel_err(join(h, I), ref) < 2e-4          # max |d| / rms: gate and up each carry the GEMM's 5e-5-class input rounding, the product both (measured <= 8.5e-5)
    # the unfused form of round 5: fp32 [gate | up] rows, then the SwiGLU kernel
    w2s = torch.cat([torch.cat([wg, wu], 0)] * 2, dim=1).contiguous()
    bs = torch.cat([bg, bu]).contiguous() if bias else None
    gu = ops.gemm_hp(a_s, w2s, bs)
    h0 = ops.swiglu_split(gu, I)
    # both are (hi, lo) pairs of the same fp32-class value: each within 2^-17 |x| of it, plus the fp32 noise of expf / the division
    d = (join(h, I).double() - join(h0, I).double()).abs()
    lim = join(h0, I).double().abs() * 2.0 ** -15 + 4e-6 * ref.pow(2).mean().sqrt()
    assert bool((d <= lim).all()), f"fused vs unfused SwiGLU: worst ratio {(d / lim).max().item():.2f}"
    if M > 64:
        # every dispatch variant of the tile kernel writes the same pairs: forced tile heights, a peeled ragged tail (its rows go through the few-row
        # kernel at K' = 2K), the column split (second launch at a column offset of the SAME hi / lo halves), the 128^2 kernel
        try:
            for kn in (dict(mf=2), dict(mf=3), dict(mf=4), dict(peel=2), dict(colsplit=2), dict(mode256=0)):
                ops.gemm_knobs(**kn)
                hv = ops.gemm_hp(a_s, w2, b, epilogue=ops.EPI_SWIGLU, out_mode=ops.OUT_SPLIT)
                if "peel" in kn or "mode256" in kn:               # peeled rows are summed in the few-row kernel's wave order, the 128^2 kernel walks K in its own phase order: fp32 noise apart
                    dd = (join(hv, I).double() - join(h, I).double()).abs()
                    assert bool((dd <= lim).all()), f"split SwiGLU under {kn}: worst ratio {(dd / lim).max().item():.2f}"
                else:
                    assert torch.equal(hv, h), f"split SwiGLU differs under dispatch knob {kn}"
                ops.gemm_knobs(mode256=1, mf=0, peel=1, colsplit=1, group_m=8)
        finally:
            ops.gemm_knobs(mode256=1, mf=0, peel=1, colsplit=1, group_m=8)


def ref_attn64(q, k, v, cq, ck, H, D):
    out = torch.zeros(q.shape[0], H * D, dtype=torch.float64, device=q.device)
    q, k, v = q.double(), k.double(), v.double()
    for i in range(len(cq) - 1):
        qs = q[cq[i]:cq[i + 1]].view(-1, H, D)
        ks = k[ck[i]:ck[i + 1]].view(-1, H, D)
        vs = v[ck[i]:ck[i + 1]].view(-1, H, D)
        if qs.shape[0] == 0:
            continue
        s = torch.einsum("qhd,khd->hqk", qs, ks) * D ** -0.5
        out[cq[i]:cq[i + 1]] = torch.einsum("hqk,khd->qhd", s.softmax(-1), vs).reshape(-1, H * D)
    return out


@pytest.mark.parametrize("D,H,lq,lk", [
    (80, 16, [8, 5, 10], [529, 345, 16]),           # query→image: key chunks of 256 incl. a ragged last chunk
    (80, 16, [8, 8], [8, 8]),                       # self-attention of the object queries
    (80, 4, [20, 3, 0, 17], [300, 9, 4, 257]),      # > 16 queries per segment (two query blocks), empty segment
    (80, 16, [2116, 64], [8, 7]),                   # image→query: thread per row
    (80, 2, [300, 270], [40, 33]),                  # image→query with two key chunks (online rescale across chunks)
    (32, 3, [5], [700]), (64, 2, [400], [5]), (128, 2, [7], [260]), (128, 2, [260], [7]),
])
@pytest.mark.parametrize("mfma", [False, True], ids=["valu", "mfma"])
def test_attn_f32(ops, D, H, lq, lk, mfma):
    if mfma and D not in (80, 128):
        pytest.skip("padt_attn_f32_mfma: head widths 80 / 128")
    cq, ck = [0], [0]
    for a, b in zip(lq, lk):
        cq.append(cq[-1] + a)
        ck.append(ck[-1] + b)
    # fused-projection style inputs: k and v are column slices of one (Tk, 2*H*D) buffer
    q = rndf(cq[-1], H * D, seed=21)
    kv = rndf(ck[-1], 2 * H * D, seed=22)
    k, v = kv[:, : H * D], kv[:, H * D:]
    k[0] = q[0] * 3                                               # one dominant key → exercises the running-max path
    out = ops.attn_f32(q, k, v, torch.tensor(cq, dtype=torch.int32, device="cuda"), torch.tensor(ck, dtype=torch.int32, device="cuda"),
                       max(lq), max(lk), H, D, mfma=mfma)
    ref = ref_attn64(q, k, v, cq, ck, H, D)
    got = join(out, H * D)
    rows = torch.cat([torch.arange(cq[i], cq[i + 1]) for i in range(len(lq)) if lk[i] > 0]).cuda()
    split_close(got[rows], ref[rows], f"attn_f32 {lq}x{lk}", noise=8e-6 if mfma else 2e-6)


@pytest.mark.parametrize("D,Hq,Hkv,lq,lk", [
    (128, 16, 2, [577, 570, 33], [577, 570, 33]),    # the prompt pass: causal GQA self-attention, thread-per-row kernel (two key-chunk regimes)
    (128, 4, 2, [40, 7, 64], [40, 7, 64]),           # short prompts: the few-queries kernel with the causal mask
    (128, 28, 4, [5, 3], [300, 9]),                  # more keys than queries: bottom-right alignment (a query block appended to a cache)
    (80, 4, 1, [100], [100]),
    (128, 28, 4, [1, 1, 1], [300, 9, 33]),           # one query per segment, G = 7: the decode step's head mode (7 of 16 columns, keys split over 4 waves)
    (128, 16, 2, [2, 1], [70, 130]),                 # G = 8: two tokens per tile in head mode, causal between them
])
@pytest.mark.parametrize("mfma", [False, True], ids=["valu", "mfma"])
def test_attn_f32_causal_gqa(ops, D, Hq, Hkv, lq, lk, mfma):
    """Round 5 (the reference-precision LLM, HF:641-689): kv_group + causal against the fp64 statement — key j visible to query i iff j <= i + nk - nq."""
    cq, ck = [0], [0]
    for a, b in zip(lq, lk):
        cq.append(cq[-1] + a)
        ck.append(ck[-1] + b)
    g = Hq // Hkv
    qkv = rndf(max(cq[-1], ck[-1]), (Hq + 2 * Hkv) * D, seed=41)   # one fused row buffer: q | k | v column slices (strided views)
    q, k, v = qkv[: cq[-1], : Hq * D], qkv[: ck[-1], Hq * D: (Hq + Hkv) * D], qkv[: ck[-1], (Hq + Hkv) * D:]
    out = ops.attn_f32(q, k, v, torch.tensor(cq, dtype=torch.int32, device="cuda"), torch.tensor(ck, dtype=torch.int32, device="cuda"),
                       max(lq), max(lk), Hq, D, kv_group=g, causal=True, mfma=mfma)
    ref = torch.zeros(cq[-1], Hq * D, dtype=torch.float64, device="cuda")
    for i in range(len(lq)):
        qs = q[cq[i]:cq[i + 1]].double().view(-1, Hq, D)
        ks = k[ck[i]:ck[i + 1]].double().view(-1, Hkv, D).repeat_interleave(g, 1)
        vs = v[ck[i]:ck[i + 1]].double().view(-1, Hkv, D).repeat_interleave(g, 1)
        sc = torch.einsum("qhd,khd->hqk", qs, ks) * D ** -0.5
        vis = torch.arange(lk[i], device="cuda")[None, :] <= torch.arange(lq[i], device="cuda")[:, None] + (lk[i] - lq[i])
        sc = sc.masked_fill(~vis[None], float("-inf"))
        ref[cq[i]:cq[i + 1]] = torch.einsum("hqk,khd->qhd", sc.softmax(-1), vs).reshape(-1, Hq * D)
    # mfma: every output is ONE fp32 fmaf chain over up to 577 keys (the VALU kernels keep 2-4 partial sums): sqrt(n) eps x rms, 4-5 sigma over 1e6 outputs
    split_close(join(out, Hq * D), ref, f"attn_f32 causal GQA {lq}x{lk}", noise=8e-6 if mfma else 2e-6)


@pytest.mark.parametrize("mfma", [False, True], ids=["valu", "mfma"])
def test_attn_f32_over_a_strided_kv_cache_and_scatter_rows(ops, mfma):
    """The decode step of the reference-precision LLM: one query row per sample against an fp32 [K | V] cache whose samples sit S_max rows apart
    (len_k = valid keys), after the new row was scattered into its slot (padt_scatter_rows_f32)."""
    B, S, Hq, Hkv, D = 5, 96, 16, 2, 128
    lens = [37, 96, 1, 64, 80]
    if mfma:
        B, S, lens = 6, 1024, [37, 96, 1, 64, 610, 1024]           # enough keys for several rounds of the four key-splitting waves
    cache = rndf(B * S, 2 * Hkv * D, seed=51)
    new = rndf(B, (Hq + 2 * Hkv) * D, seed=52)
    where = torch.tensor([b * S + lens[b] - 1 for b in range(B)], dtype=torch.int32, device="cuda")
    expect = cache.clone()
    expect[where.long()] = new[:, Hq * D:]
    ops.scatter_rows_f32(new[:, Hq * D:], where, cache, D=2 * Hkv * D)
    assert torch.equal(cache, expect)
    cu_q = torch.arange(B + 1, dtype=torch.int32, device="cuda")
    cu_k = (torch.arange(B + 1, dtype=torch.int32, device="cuda") * S).contiguous()
    len_k = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = ops.attn_f32(new[:, : Hq * D], cache[:, : Hkv * D], cache[:, Hkv * D:], cu_q, cu_k, 1, S, Hq, D, kv_group=Hq // Hkv, len_k=len_k, mfma=mfma)
    ref = torch.zeros(B, Hq * D, dtype=torch.float64, device="cuda")
    for b in range(B):
        qs = new[b: b + 1, : Hq * D].double().view(1, Hq, D)
        ks = cache[b * S: b * S + lens[b], : Hkv * D].double().view(-1, Hkv, D).repeat_interleave(Hq // Hkv, 1)
        vs = cache[b * S: b * S + lens[b], Hkv * D:].double().view(-1, Hkv, D).repeat_interleave(Hq // Hkv, 1)
        sc = torch.einsum("qhd,khd->hqk", qs, ks) * D ** -0.5
        ref[b] = torch.einsum("hqk,khd->qhd", sc.softmax(-1), vs).reshape(-1)
    split_close(join(out, Hq * D), ref, "attn_f32 over a strided cache", noise=8e-6 if mfma else 2e-6)


def test_rope_half_f32_and_mask_scatter_f32(ops):
    T, H, D = 77, 16, 80
    x = rndf(T, 2 * H * D, seed=30)
    cos, sin = rndf(T, D, seed=31), rndf(T, D, seed=32)
    ref = x.clone()
    xv = ref[:, : H * D].view(T, H, D)
    c, s = cos[:, None, : D // 2], sin[:, None, : D // 2]
    x1, x2 = xv[..., : D // 2].clone(), xv[..., D // 2:].clone()
    xv[..., : D // 2] = x1 * c - x2 * s
    xv[..., D // 2:] = x2 * c + x1 * s
    ops.rope_half_f32_(x[:, : H * D], cos, sin, H, D)              # strided view: the k half of a fused k|v projection
    assert rel_err(x, ref) < 1e-6
    dm, n_obj = 80, 3
    grids = [(6, 8), (6, 8), (4, 4)]
    pn = [h * w for h, w in grids]
    cu = [0]
    for p in pn:
        cu.append(cu[-1] + p)
    N = cu[-1]
    e2, tok = rndf(4 * N, 4 * dm, seed=33), rndf(n_obj, dm, seed=34)
    masks = torch.zeros(n_obj, 24, 32, device="cuda")
    ops.mask_scatter_f32(e2, tok, torch.tensor(cu, dtype=torch.int32, device="cuda"),
                         torch.tensor([w for _, w in grids], dtype=torch.int32, device="cuda"), masks, n_obj, N, dm)
    ref = torch.zeros_like(masks)
    e = e2.view(N, 2, 2, 2, 2, dm)
    for o in range(n_obj):
        W = grids[o][1]
        for pi in range(pn[o]):
            n = cu[o] + pi
            r, c_ = pi // W, pi % W
            ref[o, 4 * r: 4 * r + 4, 4 * c_: 4 * c_ + 4] = (e[n] * tok[o]).sum(-1).permute(0, 2, 1, 3).reshape(4, 4)
    assert rel_err(masks, ref) < 1e-5


def test_gemm_ex_argument_errors(ops):
    from padt_amd import _lib
    a = torch.zeros(8, 64, device="cuda", dtype=BF)
    w = torch.zeros(16, 64, device="cuda", dtype=BF)
    c = torch.zeros(8, 32, device="cuda", dtype=BF)
    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    # split output needs ldc >= lo_off + N and lo_off >= N
    assert lib.padt_gemm_bf16_ex(s, a.data_ptr(), 64, w.data_ptr(), 64, 0, c.data_ptr(), 16, 0, 0, 8, 16, 64, 0, 0, 0, 0, 16) == -1
    assert lib.padt_gemm_bf16_ex(s, a.data_ptr(), 64, w.data_ptr(), 64, 0, c.data_ptr(), 32, 0, 0, 8, 16, 64, 0, 0, 0, 0, 8) == -1
    # fp32 residual needs epilogue 2 + fp32 output
    assert lib.padt_gemm_bf16_ex(s, a.data_ptr(), 64, w.data_ptr(), 64, 0, c.data_ptr(), 32, c.data_ptr(), 32, 8, 16, 64, 2, 0, 0, 1, 0) == -1
    assert lib.padt_gemm_bf16_ex(s, a.data_ptr(), 64, w.data_ptr(), 64, 0, c.data_ptr(), 32, 0, 0, 8, 16, 64, 0, 0, 0, 0, 16) == 0
