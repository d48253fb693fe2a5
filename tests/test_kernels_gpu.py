"""Per-kernel parity: every C-ABI kernel vs a plain PyTorch fp32 statement of the same op on the same bf16 inputs.

Tolerance (written once, used everywhere): a bf16 result must be within one bf16 rounding of the fp32 reference,
|out - ref| <= ulps * (2^-8 |ref| + 2e-3 * rms(ref)), ulps = 1  (fp32 outputs: 1e-4 relative to rms).  Attention uses
ulps = 6: the probabilities are rounded to bf16 before the P·V MFMA (exactly what flash-attn 2, the reference's GPU
attention, does), which adds up to 2^-8 relative error per probability on top of the output rounding.
Integer outputs are bit-exact.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from padt_amd import ops as _ops
    return _ops


@pytest.fixture
def knobs(ops):
    """Force a dispatch variant of the 256-row tile kernel for one test (padt_gemm_knobs), defaults restored afterwards."""
    yield ops.gemm_knobs
    ops.gemm_knobs(mode256=1, mf=0, peel=1, colsplit=1, group_m=8)


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(BF).cuda()


def close_bf16(out, ref, what="", ulps=1.0):
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    rms = ref.pow(2).mean().sqrt().item() + 1e-12
    err = (out - ref).abs()
    lim = (ref.abs() * 2 ** -8 + 2e-3 * rms) * ulps
    bad = err > lim
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} outside tolerance, max err {err.max().item():.3e} (rms {rms:.3e})"


def close_f32(out, ref, what="", rel=1e-4):
    rms = ref.float().pow(2).mean().sqrt().item() + 1e-12
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= rel * rms + 1e-6, f"{what}: max err {err:.3e} vs rms {rms:.3e}"


# ------------------------------------------------------------------------------------------------------------ GEMM
def interleave_gate_up(wg, wu):
    n, k = wg.shape
    return torch.stack([wg.view(n // 16, 16, k), wu.view(n // 16, 16, k)], dim=1).reshape(2 * n, k)


@pytest.mark.parametrize("M,N,K", [(300, 200, 136), (129, 128, 64), (2116, 3840, 1280), (1000, 1280, 3456), (577, 2048, 1176),
                                   (8, 2048, 2048), (5, 1000, 72), (20, 96, 320), (50, 4, 1280), (64, 1, 1280), (16, 2560, 2048)])
def test_gemm_plain_bias(ops, M, N, K):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2), rnd(N, seed=3)
    ref = a.float() @ w.float().T + b.float()
    n_pad = (N + 3) // 4 * 4
    out = torch.full((M, n_pad), 7.0, device="cuda", dtype=BF)
    ops.gemm(a, w, b, out=out)
    close_bf16(out[:, :N], ref, f"gemm {M}x{N}x{K}")
    if n_pad != N:
        assert (out[:, N:] == 7.0).all(), "wrote outside N"
    out32 = torch.zeros((M, n_pad), device="cuda", dtype=torch.float32)
    ops.gemm(a, w, None, out=out32, out_f32=True)
    close_f32(out32[:, :N], a.float() @ w.float().T, f"gemm f32 {M}x{N}x{K}", rel=2e-5)


@pytest.mark.parametrize("mf", [2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(1000, 1280, 1280), (577, 768, 512), (2116, 512, 3456)])
def test_gemm_tile256_heights(ops, mf, M, N, K, knobs):
    """The phase-pipelined kernel at every tile height (128 / 192 / 256 rows x 256 columns): ragged M and N tails, all
    epilogues, f32 output."""
    knobs(mf=mf)
    a, w, b, r = rnd(M, K, seed=41), rnd(N, K, scale=0.05, seed=42), rnd(N, seed=43), rnd(M, N, seed=44)
    lin = a.float() @ w.float().T + b.float()
    close_bf16(ops.gemm(a, w, b), lin, f"mf{mf} plain {M}x{N}x{K}")
    out = r.clone()
    ops.gemm(a, w, b, out=out, epilogue=ops.EPI_RESID, residual=out)
    close_bf16(out, lin + r.float(), f"mf{mf} resid {M}x{N}x{K}")
    close_bf16(ops.gemm(a, w, b, epilogue=ops.EPI_GELU), torch.nn.functional.gelu(lin), f"mf{mf} gelu {M}x{N}x{K}")
    out32 = torch.zeros((M, N), device="cuda", dtype=torch.float32)
    ops.gemm(a, w, None, out=out32, out_f32=True)
    close_f32(out32, a.float() @ w.float().T, f"mf{mf} f32 {M}x{N}x{K}", rel=2e-5)
    n2 = N // 2
    wg, wu = w[:n2].contiguous(), w[n2:].contiguous()
    wi = interleave_gate_up(wg, wu)
    bi = interleave_gate_up(b[:n2].reshape(n2, 1), b[n2:].reshape(n2, 1)).view(-1)
    ref = torch.nn.functional.silu(a.float() @ wg.float().T + b[:n2].float()) * (a.float() @ wu.float().T + b[n2:].float())
    close_bf16(ops.gemm(a, wi, bi, epilogue=ops.EPI_SWIGLU), ref, f"mf{mf} swiglu {M}x{N}x{K}")


@pytest.mark.parametrize("mf", [3, 4])
def test_gemm_tile256_peeled_tail(ops, mf, knobs):
    """Ragged last tile row (<= 64 rows) peeled off to the skinny kernel: same results as the un-peeled launch."""
    knobs(mf=mf)
    M, N, K = 2 * 64 * mf + 24, 1024, 640
    a, w, b, r = rnd(M, K, seed=45), rnd(N, K, scale=0.05, seed=46), rnd(N, seed=47), rnd(M, N, seed=48)
    wi = interleave_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous())
    outs = {}
    for peel in ("0", "2"):
        knobs(peel=int(peel))
        o1 = r.clone()
        ops.gemm(a, w, b, out=o1, epilogue=ops.EPI_RESID, residual=o1)
        o32 = torch.zeros((M, N), device="cuda", dtype=torch.float32)
        ops.gemm(a, w, None, out=o32, out_f32=True)
        outs[peel] = (o1, ops.gemm(a, wi, None, epilogue=ops.EPI_SWIGLU), o32)
    lin = a.float() @ w.float().T
    close_bf16(outs["2"][0], lin + b.float() + r.float(), "peeled resid")
    close_f32(outs["2"][2], lin, "peeled f32", rel=2e-5)
    body = 2 * 64 * mf
    for x0, x2 in zip(outs["0"], outs["2"]):
        assert torch.equal(x0[:body], x2[:body]), "tile rows must not change"
        close_bf16(x2[body:].float(), x0[body:].float(), "tail rows (skinny vs tile kernel)", ulps=2)


@pytest.mark.parametrize("M,N,K", [(1000, 1280, 1280), (40, 640, 512), (300, 200, 136), (2116 + 24, 768, 640)])
def test_gemm_row_scale_is_folded_rmsnorm(ops, M, N, K):
    """row_rstd + gemm(row_scale=) == RMSNorm (unit weight) → Linear, on every kernel family (256-row phase kernel,
    128^2 kernel, skinny kernel, peeled tails), plain / residual / SwiGLU epilogues."""
    x, w, b, r = rnd(M, K, seed=51), rnd(N, K, scale=0.05, seed=52), rnd(N, seed=53), rnd(M, N, seed=54)
    rstd = ops.row_rstd(x, eps=1e-6)
    ref_rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6)
    assert (rstd - ref_rstd).abs().max().item() <= 2e-6 * ref_rstd.abs().max().item()
    lin = (x.float() @ w.float().T) * ref_rstd[:, None]
    close_bf16(ops.gemm(x, w, b, row_scale=rstd), lin + b.float(), f"row_scale plain {M}x{N}x{K}")
    out = r.clone()
    ops.gemm(x, w, b, out=out, epilogue=ops.EPI_RESID, residual=out, row_scale=rstd)
    close_bf16(out, lin + b.float() + r.float(), f"row_scale resid {M}x{N}x{K}")
    if N % 32 == 0:
        n2 = N // 2
        wi = interleave_gate_up(w[:n2].contiguous(), w[n2:].contiguous())
        bi = interleave_gate_up(b[:n2].reshape(n2, 1), b[n2:].reshape(n2, 1)).view(-1)
        ref = torch.nn.functional.silu(lin[:, :n2] + b[:n2].float()) * (lin[:, n2:] + b[n2:].float())
        close_bf16(ops.gemm(x, wi, bi, epilogue=ops.EPI_SWIGLU, row_scale=rstd), ref, f"row_scale swiglu {M}x{N}x{K}")


def test_gemm_tile256_column_split_is_bit_identical(ops, knobs):
    """Column-split dispatch (last tile columns as a second launch with its own tile height) == the single launch."""
    M, N, K = 1100, 1536, 640
    a, w, b, r = rnd(M, K, seed=55), rnd(N, K, scale=0.05, seed=56), rnd(N, seed=57), rnd(M, N, seed=58)
    wi = interleave_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous())
    rstd = ops.row_rstd(a)
    outs = {}
    for cs in ("0", "2", "3"):
        knobs(colsplit=int(cs))
        o1 = r.clone()
        ops.gemm(a, w, b, out=o1, epilogue=ops.EPI_RESID, residual=o1)
        o32 = torch.zeros((M, N), device="cuda", dtype=torch.float32)
        ops.gemm(a, w, b, out=o32, out_f32=True, row_scale=rstd)
        outs[cs] = (o1, ops.gemm(a, wi, b, epilogue=ops.EPI_SWIGLU, row_scale=rstd), o32)
    for cs in ("2", "3"):
        for x0, x1 in zip(outs["0"], outs[cs]):
            assert torch.equal(x0, x1)
    close_bf16(outs["2"][0], a.float() @ w.float().T + b.float() + r.float(), "column split resid")


@pytest.mark.parametrize("name,N,K,epi,split", [("qkv", 2560, 2048, 0, 1), ("o", 2048, 2048, 2, 1), ("gate/up", 22016, 2048, 3, 1),
                                                 ("down", 2048, 11008, 2, 2), ("down 7B", 3584, 18944, 2, 2)])
def test_decode_projection_rows_do_not_depend_on_the_batch(ops, name, N, K, epi, split):
    """In-flight batching must not change a sample: a row's output bits are the same whether 8, 16, 32, 64 or 128 rows share the launch
    (the K-step-pair → wave map and the wave count of gemm_skinny_kernel do not depend on the row count; only the number of pairs in
    flight does).  Real PaDT_Pro_3B / 7B decode shapes — the small test config has too few K-steps to tell."""
    w = rnd(N, K, scale=0.02, seed=171)
    wp = ops.pack_weight(w)
    x64 = rnd(128, K, seed=172)
    n_out = N // 2 if epi == 3 else N
    res64 = rnd(128, n_out, seed=173)
    first = None
    for B in (8, 16, 32, 64, 128):
        B16 = (B + 15) // 16 * 16
        xp = torch.zeros(B16, K, device="cuda", dtype=BF)
        ops.pack_rows(x64[:B].contiguous(), xp, B, to_packed=True)
        o = torch.zeros(B16, n_out, device="cuda", dtype=BF)
        if epi == 2:
            ops.pack_rows(res64[:B].contiguous(), o, B, to_packed=True)
            ops.gemm_packed(xp, wp, N, out=o, epilogue=2, residual=o, split_k=split, workspace=ops.new_splitk_workspace(N, 2, "cuda"),
                            a_packed=True, c_packed=True, rows=B)
        else:
            ops.gemm_packed(xp, wp, N, out=o, epilogue=epi, norm_eps=1e-6, a_packed=True, c_packed=True, rows=B)
        un = torch.zeros(B, n_out, device="cuda", dtype=BF)
        ops.pack_rows(o, un, B, to_packed=False)
        if first is None:
            first = un[:8].clone()
        assert torch.equal(un[:8], first), f"{name}: rows 0..7 change when {B} rows share the launch"
        if B == 128:                                             # and the 128-row launch is right on all of its rows
            xf = x64.float()
            lin = xf @ w.float().T
            if epi == 2:
                ref = lin + res64.float()
            else:
                lin = lin * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
                ref = lin if epi == 0 else torch.nn.functional.silu(lin.view(128, -1, 2, 16)[:, :, 0]) .reshape(128, -1) * lin.view(128, -1, 2, 16)[:, :, 1].reshape(128, -1)
            close_bf16(un, ref, f"{name} at 128 rows", ulps=2)


@pytest.mark.parametrize("split", [2, 4, 8])
def test_split_k_sum_does_not_depend_on_which_block_finishes(ops, split):
    """split-K partials are combined by whichever block arrives last; the fp32 sum must not depend on who that is (partials are added in
    split order) — 40 launches of the 3B down-projection shape at 64 rows, all bit-identical."""
    M, N, K = 64, 2048, 11008
    x, w, r = rnd(M, K, seed=501), rnd(N, K, scale=0.02, seed=502), rnd(M, N, seed=503)
    wp = ops.pack_weight(w)
    ws = ops.new_splitk_workspace(N, split, "cuda")
    first = None
    for _ in range(40):
        o = r.clone()
        ops.gemm_packed(x, wp, N, out=o, epilogue=ops.EPI_RESID, residual=o, split_k=split, workspace=ws)
        if first is None:
            first = o
            close_bf16(o, x.float() @ w.float().T + r.float(), f"split-{split}", ulps=2)
        assert torch.equal(o, first)


@pytest.mark.parametrize("B", [40, 72, 96])
def test_decode_projection_reads_only_the_row_blocks_it_was_given(ops, B):
    """40 / 72 / 96 decode rows run the 64- / 128-row launch shapes with 3 of 4 / 5 of 8 / 6 of 8 sixteen-row blocks present.  The packed
    activation buffer holds ceil(B / 16) blocks and what follows it in memory is NaN: the rows that exist must come out as the 128-row
    launch computes them, bit for bit (gate/up with the fused norm).  (The kernels skip the absent blocks — gemm_skinny_kernel's xok[];
    an out-of-buffer read cannot be observed from here, the e2e group-size test is what faulted without the guard.)"""
    N, K = 1024, 512
    w = rnd(N, K, scale=0.05, seed=401)
    wp = ops.pack_weight(w)
    x = rnd(128, K, seed=402)
    B16 = (B + 15) // 16 * 16
    ref_x = torch.zeros(128, K, device="cuda", dtype=BF)
    ops.pack_rows(x, ref_x, 128, to_packed=True)
    ref = torch.zeros(128, N // 2, device="cuda", dtype=BF)
    ops.gemm_packed(ref_x, wp, N, out=ref, epilogue=3, norm_eps=1e-6, a_packed=True, c_packed=True, rows=128)
    back = torch.full((B16 + 64, K), float("nan"), device="cuda", dtype=BF)
    xp = back[:B16]
    xp.zero_()
    ops.pack_rows(x[:B].contiguous(), xp, B, to_packed=True)
    out = torch.zeros(B16, N // 2, device="cuda", dtype=BF)
    ops.gemm_packed(xp, wp, N, out=out, epilogue=3, norm_eps=1e-6, a_packed=True, c_packed=True, rows=B)
    a, b = torch.zeros(B, N // 2, device="cuda", dtype=BF), torch.zeros(128, N // 2, device="cuda", dtype=BF)
    ops.pack_rows(out, a, B, to_packed=False)
    ops.pack_rows(ref, b, 128, to_packed=False)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b[:B])


@pytest.mark.parametrize("M,K", [(1000, 640), (40, 256), (2 * 256 + 24, 512), (300, 136)])
def test_gemm_rope_epilogue(ops, M, K, knobs):
    """qkv projection with RoPE fused into the epilogue (pair-interleaved q/k rows) vs Linear → rotate-half RoPE in fp32 on the
    ORIGINAL layout, compared after undoing the permutation; every kernel family (phase kernel, 128^2, skinny, peeled tail)."""
    from padt_amd.weights import interleave_rope_rows
    H, D = 4, 80
    N = 3 * H * D
    x, w, b = rnd(M, K, seed=65), rnd(N, K, scale=0.05, seed=66), rnd(N, seed=67)
    ang = torch.rand(M, D // 2, device="cuda") * 30
    cos = torch.cat([ang, ang], -1).cos().contiguous()
    sin = torch.cat([ang, ang], -1).sin().contiguous()
    rstd = ops.row_rstd(x)
    lin = (x.float() @ w.float().T) * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6) + b.float()
    qk = lin[:, : 2 * H * D].view(M, 2 * H, D)
    rot = torch.cat([-qk[..., D // 2:], qk[..., : D // 2]], -1)
    ref = lin.clone()
    ref[:, : 2 * H * D] = (qk * cos[:, None] + rot * sin[:, None]).reshape(M, -1)
    wi, bi = interleave_rope_rows(w, 2 * H, D), interleave_rope_rows(b, 2 * H, D)
    if M == 2 * 256 + 24:
        knobs(peel=2)
        knobs(mf=4)
    out = torch.zeros(M, N, device="cuda", dtype=BF)
    ops.gemm_rope(x, wi, bi, out, cos, sin, 2 * H * D, D, row_scale=rstd)
    # undo the pair interleave of the q / k columns: column 2i <- d = i, column 2i + 1 <- d = i + D/2
    un = out.float().clone()
    v = out.float()[:, : 2 * H * D].view(M, 2 * H, D // 2, 2)
    un[:, : 2 * H * D] = torch.cat([v[..., 0], v[..., 1]], -1).reshape(M, -1)
    close_bf16(un, ref, f"gemm_rope {M}x{N}x{K}")
    with pytest.raises(Exception, match="padt_gemm_rope_bf16"):
        ops.gemm_rope(x, wi, bi, out, cos, sin, 2 * H * D + 4, D)


@pytest.mark.parametrize("M", [7, 40, 700])
def test_gemm_epilogues(ops, M):
    K, N = 256, 384
    a, w, b, r = rnd(M, K, seed=4), rnd(N, K, scale=0.06, seed=5), rnd(N, seed=6), rnd(M, N, seed=7)
    lin = a.float() @ w.float().T + b.float()
    close_bf16(ops.gemm(a, w, b, epilogue=ops.EPI_GELU), torch.nn.functional.gelu(lin), f"gelu M={M}")
    close_bf16(ops.gemm(a, w, b, epilogue=ops.EPI_RESID, residual=r), lin + r.float(), f"resid M={M}")
    out = r.clone()                                           # in-place residual (C aliases R) as the model uses it
    ops.gemm(a, w, b, out=out, epilogue=ops.EPI_RESID, residual=out)
    close_bf16(out, lin + r.float(), f"resid inplace M={M}")
    wg, wu, bg, bu = rnd(N, K, scale=0.06, seed=8), rnd(N, K, scale=0.06, seed=9), rnd(N, seed=10), rnd(N, seed=11)
    wi = interleave_gate_up(wg, wu)
    bi = interleave_gate_up(bg.view(N, 1), bu.view(N, 1)).view(-1)
    ref = torch.nn.functional.silu(a.float() @ wg.float().T + bg.float()) * (a.float() @ wu.float().T + bu.float())
    close_bf16(ops.gemm(a, wi, bi, epilogue=ops.EPI_SWIGLU), ref, f"swiglu M={M}")


@pytest.mark.parametrize("M,N,K", [(8, 2560, 2048), (8, 22016, 2048), (8, 2048, 11008), (3, 96, 256), (20, 640, 512), (64, 704, 256)])
def test_gemm_fused_rmsnorm(ops, M, N, K):
    """out = rstd(x) * (x @ W^T) + b with W carrying the folded norm weight: the kernel's operands are exactly x and W, so
    the fp32 statement of the same expression is a 1-ulp reference."""
    x = rnd(M, K, seed=14)
    w, b = rnd(N, K, scale=0.05, seed=16), rnd(N, seed=17)
    xf = x.float()
    rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    close_bf16(ops.gemm_rmsnorm(x, w, b), (xf @ w.float().T) * rstd + b.float(), f"norm+gemm {M}x{N}x{K}")
    wi = interleave_gate_up(w[: N // 2], w[N // 2:])
    a_, b_ = (xf @ w[: N // 2].float().T) * rstd, (xf @ w[N // 2:].float().T) * rstd
    close_bf16(ops.gemm_rmsnorm(x, wi, None, epilogue=ops.EPI_SWIGLU), torch.nn.functional.silu(a_) * b_, f"norm+swiglu {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(8, 2560, 2048), (8, 2048, 2048), (8, 22016, 2048), (8, 2048, 11008), (1, 2048, 3584), (5, 640, 256),
                                   (16, 512, 2048), (40, 96, 4096), (8, 1008, 6152), (64, 704, 264)])
def test_gemm_packed_weights(ops, M, N, K):
    """Decode projections over the fragment-packed weight image: plain / norm+bias / residual in place / norm+SwiGLU."""
    x = rnd(M, K, seed=18)
    w, b, r = rnd(N, K, scale=0.05, seed=19), rnd(N, seed=20), rnd(M, N, seed=21)
    wp = ops.pack_weight(w)
    xf = x.float()
    lin = xf @ w.float().T
    rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    close_bf16(ops.gemm_packed(x, wp, N), lin, f"packed {M}x{N}x{K}")
    close_bf16(ops.gemm_packed(x, wp, N, b, norm_eps=1e-6), lin * rstd + b.float(), f"packed norm+bias {M}x{N}x{K}")
    out = r.clone()
    ops.gemm_packed(x, wp, N, out=out, epilogue=ops.EPI_RESID, residual=out)
    for split in (2, 4):                                  # split-K: partials merged by the last block, ticket self-resets
        ws = ops.new_splitk_workspace(N, split, "cuda")
        for _ in range(2):
            o2 = r.clone()
            ops.gemm_packed(x, wp, N, out=o2, epilogue=ops.EPI_RESID, residual=o2, split_k=split, workspace=ws)
            close_bf16(o2, lin + r.float(), f"packed split-{split} resid {M}x{N}x{K}")
        close_bf16(ops.gemm_packed(x, wp, N, b, split_k=split, workspace=ws), lin + b.float(), f"packed split-{split} {M}x{N}x{K}")
    close_bf16(out, lin + r.float(), f"packed resid {M}x{N}x{K}")
    if N % 32 == 0:
        wi = ops.pack_weight(interleave_gate_up(w[: N // 2], w[N // 2:]))
        bi = interleave_gate_up(b[: N // 2].view(-1, 1), b[N // 2:].view(-1, 1)).view(-1)
        g_, u_ = (xf @ w[: N // 2].float().T) * rstd + b[: N // 2].float(), (xf @ w[N // 2:].float().T) * rstd + b[N // 2:].float()
        close_bf16(ops.gemm_packed(x, wi, N, bi, epilogue=ops.EPI_SWIGLU, norm_eps=1e-6), torch.nn.functional.silu(g_) * u_,
                   f"packed swiglu {M}x{N}x{K}")


def test_gemm_strided_views_and_argument_errors(ops):
    from padt_amd._lib import PaDTHipError
    big = rnd(100, 512, seed=12)
    a = big[:, 128:128 + 200]                                  # row-strided view, 16-byte aligned offset
    w = rnd(64, 200, scale=0.1, seed=13)
    close_bf16(ops.gemm(a, w), a.float() @ w.float().T, "strided A")
    with pytest.raises(PaDTHipError):
        ops.gemm(big[:, :100], rnd(8, 100))                    # K not a multiple of 8


# ------------------------------------------------------------------------------------------------------------ attention
def ref_attn(q, k, v, cu_q, cu_k, H, Hkv, D, causal):
    Tq = q.shape[0]
    out = torch.zeros(Tq, H * D, device=q.device)
    qf, kf, vf = q.float().view(Tq, H, D), k.float().view(-1, Hkv, D), v.float().view(-1, Hkv, D)
    rep = H // Hkv
    for s in range(len(cu_q) - 1):
        q0, q1, k0, k1 = cu_q[s], cu_q[s + 1], cu_k[s], cu_k[s + 1]
        if q1 == q0:
            continue
        qs = qf[q0:q1].transpose(0, 1)
        ks = kf[k0:k1].transpose(0, 1).repeat_interleave(rep, 0)
        vs = vf[k0:k1].transpose(0, 1).repeat_interleave(rep, 0)
        sc = qs @ ks.transpose(1, 2) * D ** -0.5
        if causal:
            Lq, Lk = q1 - q0, k1 - k0
            sc = sc.masked_fill(~torch.ones(Lq, Lk, dtype=torch.bool, device=q.device).tril(Lk - Lq), float("-inf"))
        out[q0:q1] = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(q1 - q0, H * D)
    return out


@pytest.mark.parametrize("D,H,Hkv,lens,causal", [
    (80, 16, 16, [64] * 5 + [48] * 3 + [36], False),          # ViT window layer segments
    (80, 16, 16, [2116], False),                               # ViT full layer
    (128, 16, 2, [577, 100, 1, 65], True),                     # LLM prefill, GQA 8:1, ragged
    (32, 4, 2, [10, 130, 64], True),
    (64, 2, 2, [70], False),
    (128, 28, 4, [577, 33, 7], True),                          # 7B heads: group 7 (64 rows = 9 tokens x 7 heads + 1 dead row)
    (80, 16, 4, [300, 64, 257], False),                        # 32 query rows per wave with a kv group of 4
    (80, 8, 1, [290, 5], True),                                # group 8 at 128 rows per block, causal
])
def test_attn_self(ops, D, H, Hkv, lens, causal):
    T = sum(lens)
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    qkv = rnd(T, (H + 2 * Hkv) * D, seed=20)
    q, k, v = qkv[:, : H * D], qkv[:, H * D: (H + Hkv) * D], qkv[:, (H + Hkv) * D:]
    out = torch.zeros(T, H * D, device="cuda", dtype=BF)
    cu_t = torch.tensor(cu, dtype=torch.int32, device="cuda")
    ops.attn_varlen(q, k, v, out, cu_t, cu_t, max(lens), H, Hkv, D, causal=causal)
    close_bf16(out, ref_attn(q, k, v, cu, cu, H, Hkv, D, causal), f"attn D={D} lens={lens[:3]}..", ulps=6)


@pytest.mark.parametrize("lq,lk", [([8, 5, 10], [529, 345, 16]), ([2116, 64], [8, 7]), ([3, 0, 4], [12, 9, 130])])
def test_attn_cross_decoder_shapes(ops, lq, lk):
    D, H = 80, 16
    cq, ck = [0], [0]
    for a, b in zip(lq, lk):
        cq.append(cq[-1] + a)
        ck.append(ck[-1] + b)
    q, k, v = rnd(cq[-1], H * D, seed=21), rnd(ck[-1], H * D, seed=22), rnd(ck[-1], H * D, seed=23)
    out = torch.zeros(cq[-1], H * D, device="cuda", dtype=BF)
    ops.attn_varlen(q, k, v, out, torch.tensor(cq, dtype=torch.int32, device="cuda"),
                    torch.tensor(ck, dtype=torch.int32, device="cuda"), max(lq), H, H, D)
    close_bf16(out, ref_attn(q, k, v, cq, ck, H, H, D, False), f"cross {lq}x{lk}", ulps=6)


def test_attn_online_softmax_rescale_branch(ops):
    """One key far above the rest in a late tile forces the running-max rescale (guide §5.4 rule 26)."""
    D, H, L = 128, 2, 200
    q, k, v = rnd(L, H * D, seed=24), rnd(L, H * D, seed=25), rnd(L, H * D, seed=26)
    k[150] = (q[10].float() * 4).to(BF)
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    out = torch.zeros(L, H * D, device="cuda", dtype=BF)
    ops.attn_varlen(q, k, v, out, cu, cu, L, H, H, D)
    close_bf16(out, ref_attn(q, k, v, [0, L], [0, L], H, H, D, False), "rescale", ulps=6)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 16, 2), (32, 4, 2), (128, 28, 4)])
def test_decode_attn(ops, D, Hq, Hkv):
    B, S_max = 3, 640
    lens = [578, 130, 64]
    q = rnd(B, Hq * D, seed=30)
    kc = rnd(B, Hkv, S_max, D, seed=31)
    v = rnd(B, Hkv, S_max, D, seed=32)
    vt = v.transpose(2, 3).contiguous()
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    ws = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
    out = torch.zeros(B, Hq * D, device="cuda", dtype=BF)
    ops.decode_attn(q, kc, vt, lens_t, out, ws, Hq, Hkv, D, S_max, max(lens))
    rep = Hq // Hkv
    ref = torch.zeros(B, Hq * D, device="cuda")
    for b in range(B):
        kk = kc[b, :, : lens[b]].float().repeat_interleave(rep, 0)      # Hq,L,D
        vv = v[b, :, : lens[b]].float().repeat_interleave(rep, 0)
        sc = torch.einsum("hd,hld->hl", q[b].float().view(Hq, D), kk) * D ** -0.5
        ref[b] = torch.einsum("hl,hld->hd", torch.softmax(sc, -1), vv).reshape(-1)
    close_bf16(out, ref, "decode attn", ulps=6)


@pytest.mark.parametrize("M,N,K", [(8, 2048, 2048), (16, 2560, 2048), (32, 2048, 11008), (21, 704, 512), (40, 640, 256)])
def test_gemm_packed_activations(ops, M, N, K):
    """Fragment-packed activation layout (A and/or C, R): bit-identical to the row-major path on the same operands."""
    x = rnd(M, K, seed=61)
    w, b, r = rnd(N, K, scale=0.05, seed=62), rnd(N, seed=63), rnd(M, N, seed=64)
    wp = ops.pack_weight(w)
    M16 = (M + 15) // 16 * 16
    xp = torch.zeros(M16, K, device="cuda", dtype=BF)
    ops.pack_rows(x, xp, M, to_packed=True)
    back = torch.zeros_like(x)
    ops.pack_rows(xp, back, M, to_packed=False)
    assert torch.equal(back, x), "pack/unpack round trip"
    ref_plain = ops.gemm_packed(x, wp, N, b, norm_eps=1e-6)
    got = ops.gemm_packed(xp, wp, N, b, norm_eps=1e-6, a_packed=True, rows=M)
    assert torch.equal(got, ref_plain), "A packed, C row-major"
    ref_res = r.clone()
    ops.gemm_packed(x, wp, N, out=ref_res, epilogue=ops.EPI_RESID, residual=ref_res)
    rp = torch.zeros(M16, N, device="cuda", dtype=BF)
    ops.pack_rows(r, rp, M, to_packed=True)
    ws = ops.new_splitk_workspace(N, 2, "cuda")
    for split in (1, 2):
        cp = rp.clone()
        ops.gemm_packed(xp, wp, N, out=cp, epilogue=ops.EPI_RESID, residual=cp, a_packed=True, c_packed=True, rows=M,
                        split_k=split, workspace=ws)
        un = torch.zeros(M, N, device="cuda", dtype=BF)
        ops.pack_rows(cp, un, M, to_packed=False)
        if split == 1:
            assert torch.equal(un, ref_res), "A, C, R packed"
        else:
            close_bf16(un, x.float() @ w.float().T + r.float(), "packed split-K")
    if N % 32 == 0:
        ref_sw = ops.gemm_packed(x, wp, N, b, epilogue=ops.EPI_SWIGLU, norm_eps=1e-6)
        hp = torch.zeros(M16, N // 2, device="cuda", dtype=BF)
        ops.gemm_packed(xp, wp, N, b, out=hp, epilogue=ops.EPI_SWIGLU, norm_eps=1e-6, a_packed=True, c_packed=True, rows=M)
        un = torch.zeros(M, N // 2, device="cuda", dtype=BF)
        ops.pack_rows(hp, un, M, to_packed=False)
        assert torch.equal(un, ref_sw), "SwiGLU into a packed buffer"


@pytest.mark.parametrize("D,Hq,Hkv,sec", [(128, 16, 2, (16, 24, 24)), (32, 4, 2, (4, 6, 6)), (128, 28, 4, (16, 24, 24))])
def test_decode_attn_rope_matches_unfused_pipeline(ops, D, Hq, Hkv, sec):
    """rope table + (rope, append, split attention) + merge == llm_qkv_post → decode_attn, and the fp32 reference."""
    B, S_max = 3, 1344
    slots = [577, 63, 1290]
    qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=33)
    kc = rnd(B, Hkv, S_max, D, seed=34)
    v = rnd(B, Hkv, S_max, D, seed=35)
    vt = v.transpose(2, 3).contiguous()
    slot_t = torch.tensor(slots, dtype=torch.int32, device="cuda")
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
    # the prompt path (llm_qkv_post) and the decode path (decode_attn_rope) must rotate a token bit-identically at EVERY position: a
    # draw of positions where hipcc's differing fma contraction of x1*c - x2*s flipped one bf16 was found by an unseeded run of this test
    for seed in range(1, 12):
        gpos = torch.randint(0, 4000, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(seed)).cuda()
        qa, qb = torch.zeros(B, Hq * D, device="cuda", dtype=BF), torch.zeros(B, Hq * D, device="cuda", dtype=BF)
        ka, va, kb, vb = kc.clone(), vt.clone(), kc.clone(), vt.clone()
        csx = torch.zeros(B, D // 2, 2, device="cuda")
        ops.rope_table(gpos, inv, csx, D, sec)
        ops.decode_attn_rope(qkv, csx, slot_t, ka, va, qa, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max)
        ops.llm_qkv_post(qkv, gpos, inv, qb, kb, vb, Hq, Hkv, D, S_max, sec, slot=slot_t)
        assert torch.equal(ka, kb) and torch.equal(va, vb), f"rotated K / V differ between the decode and the prompt path (positions seed {seed})"
        ob = torch.zeros_like(qa)
        ops.decode_attn(qb, kb, vb, slot_t + 1, ob, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, max(slots) + 1)
        assert torch.equal(qa, ob), f"attention over the rotated q differs between the two paths (positions seed {seed})"
    pos = torch.randint(0, 900, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(0)).cuda()
    kc1, vt1, kc2, vt2 = kc.clone(), vt.clone(), kc.clone(), vt.clone()
    cs = torch.zeros(B, D // 2, 2, device="cuda")
    ops.rope_table(pos, inv, cs, D, sec)
    ws = ops.new_decode_workspace(B, Hkv, D, S_max, "cuda")
    out1 = torch.zeros(B, Hq * D, device="cuda", dtype=BF)
    ops.decode_attn_rope(qkv, cs, slot_t, kc1, vt1, out1, ws, Hq, Hkv, D, S_max, S_max)
    q2 = torch.zeros(B, Hq * D, device="cuda", dtype=BF)
    ops.llm_qkv_post(qkv, pos, inv, q2, kc2, vt2, Hq, Hkv, D, S_max, sec, slot=slot_t)
    assert torch.equal(kc1, kc2) and torch.equal(vt1, vt2), "cache append differs"
    rep = Hq // Hkv
    ref = torch.zeros(B, Hq * D, device="cuda")
    for b in range(B):
        L = slots[b] + 1
        kk = kc2[b, :, :L].float().repeat_interleave(rep, 0)
        vv = vt2[b, :, :, :L].float().transpose(1, 2).repeat_interleave(rep, 0)
        sc = torch.einsum("hd,hld->hl", q2[b].float().view(Hq, D), kk) * D ** -0.5
        ref[b] = torch.einsum("hl,hld->hd", torch.softmax(sc, -1), vv).reshape(-1)
    close_bf16(out1, ref, "decode attn + rope", ulps=6)
    out2 = torch.zeros_like(out1)
    ops.decode_attn(q2, kc2, vt2, slot_t + 1, out2, ws, Hq, Hkv, D, S_max, max(slots) + 1)
    assert torch.equal(out1, out2), "fused rope/append path differs from llm_qkv_post + decode_attn"
    outp = torch.zeros(16, Hq * D, device="cuda", dtype=BF)          # same call writing the fragment-packed layout
    ops.decode_attn_rope(qkv, cs, slot_t, kc.clone(), vt.clone(), outp, ws, Hq, Hkv, D, S_max, S_max, out_packed=True)
    un = torch.zeros_like(out1)
    ops.pack_rows(outp, un, B, to_packed=False)
    assert torch.equal(un, out1)


@pytest.mark.parametrize("Hq,Hkv", [(16, 2), (28, 4)])
@pytest.mark.parametrize("dt", [BF, torch.float16], ids=["bf16", "fp16"])
def test_decode_attn_rope_on_fragment_packed_caches(ops, Hq, Hkv, dt):
    """Round 6: the one-launch decode attention over FRAGMENT-PACKED caches (padt_decode_attn_rope, cache_packed = 1) against the two-launch
    form on row-major caches, in both operand types:
      * the caches it appends to, un-packed, are the row-major call's bit for bit — and the prompt pass (llm_qkv_post, cache_packed = 1) writes
        the same images, at every position draw (same rotation roundings at all three sites);
      * outputs are the row-major call's up to two 16-bit roundings (different merge order) and sit at the same distance from the fp32 statement;
      * a sample's output does not depend on what else is in the batch (a block sees one sample) nor on the capacity S_max of the cache
        (splits sit at absolute key positions): merged decode groups == batch-at-a-time;
      * the fragment-packed output layout (out_packed) holds the same rows."""
    D, sec = 128, (16, 24, 24)
    B, S_max = 5, 1344
    slots = [577, 63, 1290, 0, 64]
    g = torch.Generator().manual_seed(91)
    mk = lambda *shape: torch.randn(*shape, generator=g).cuda().to(dt)
    qkv, kc, vt = mk(B, (Hq + 2 * Hkv) * D), mk(B, Hkv, S_max, D), mk(B, Hkv, D, S_max)
    slot_t = torch.tensor(slots, dtype=torch.int32, device="cuda")
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
    assert torch.equal(ops.unpack_k_cache(ops.pack_k_cache(kc)), kc) and torch.equal(ops.unpack_vt_cache(ops.pack_vt_cache(vt)), vt)
    for seed in range(1, 8):
        gpos = torch.randint(0, 4000, (3, B), dtype=torch.int32, generator=torch.Generator().manual_seed(seed)).cuda()
        cs = torch.zeros(B, D // 2, 2, device="cuda")
        ops.rope_table(gpos, inv, cs, D, sec)
        k1, v1 = kc.clone(), vt.clone()
        o1 = torch.zeros(B, Hq * D, device="cuda", dtype=dt)
        ops.decode_attn_rope(qkv, cs, slot_t, k1, v1, o1, ops.new_decode_workspace(B, Hkv, D, S_max, "cuda"), Hq, Hkv, D, S_max, S_max)
        kp, vp = ops.pack_k_cache(kc), ops.pack_vt_cache(vt)
        o2 = torch.zeros_like(o1)
        ops.decode_attn_rope(qkv, cs, slot_t, kp, vp, o2, None, Hq, Hkv, D, S_max, S_max, cache_packed=True)
        assert torch.equal(ops.unpack_k_cache(kp), k1) and torch.equal(ops.unpack_vt_cache(vp), v1), f"packed append differs (positions seed {seed})"
        kq, vq = ops.pack_k_cache(kc), ops.pack_vt_cache(vt)
        qd = torch.zeros_like(o1)
        ops.llm_qkv_post(qkv, gpos, inv, qd, kq, vq, Hq, Hkv, D, S_max, sec, slot=slot_t, cache_packed=True)
        assert torch.equal(kq, kp) and torch.equal(vq, vp), f"prompt-pass append into the packed images differs (positions seed {seed})"
        assert torch.isfinite(o2.float()).all()
        # one block per (kv head, sample) (cache_packed = 2) and two blocks, each with half of the d-tiles (3): the same bits, outputs and appends
        for force in (2, 3):
            kf_, vf_ = ops.pack_k_cache(kc), ops.pack_vt_cache(vt)
            of_ = torch.zeros_like(o1)
            ops.decode_attn_rope(qkv, cs, slot_t, kf_, vf_, of_, None, Hq, Hkv, D, S_max, S_max, cache_packed=force)
            assert torch.equal(of_, o2) and torch.equal(kf_, kp) and torch.equal(vf_, vp), f"decode attention with cache_packed = {force} differs (positions seed {seed})"
        # at most two 16-bit roundings apart (8 / 11 mantissa bits) + fp32 noise on outputs that cancel to ~0: the probabilities are rounded to
        # 16 bits relative to the RUNNING maximum here and to each split's own maximum in the two-launch form, then merged in a different order
        # (an output that cancels to a small value carries the P-rounding noise of the whole row: an absolute term of one 16-bit step of the row scale)
        eps16 = 2.0 ** -8 if dt == BF else 2.0 ** -11
        lim = torch.maximum(o1.float().abs(), o2.float().abs()) * 4 * eps16 + 2 * eps16 * o1.float().pow(2).mean().sqrt()
        assert bool(((o1.float() - o2.float()).abs() <= lim).all()), f"one-launch and two-launch outputs further apart than the operand type's P rounding allows (positions seed {seed})"
    # fp32 statement from the rotated q (prompt kernel) and the appended caches
    rep = Hq // Hkv
    ref = torch.zeros(B, Hq * D, device="cuda")
    for b in range(B):
        L = slots[b] + 1
        kk = k1[b, :, :L].float().repeat_interleave(rep, 0)
        vv = v1[b, :, :, :L].float().transpose(1, 2).repeat_interleave(rep, 0)
        sc = torch.einsum("hd,hld->hl", qd[b].float().view(Hq, D), kk) * D ** -0.5
        ref[b] = torch.einsum("hl,hld->hd", torch.softmax(sc, -1), vv).reshape(-1)
    e1, e2 = (o1.float() - ref).abs().max().item(), (o2.float() - ref).abs().max().item()
    print(f"\n[decode attention, packed caches, {Hq}:{Hkv} {dt}] |two launches - fp32| {e1:.3e}, |one launch - fp32| {e2:.3e}, differing outputs {(o1 != o2).sum().item()} of {o1.numel()}")
    assert e2 <= 1.25 * e1 + 1e-4
    # batch- and capacity-invariance: sample 2 alone, in a smaller cache
    S2 = 1344 - 64 * 0
    for rows, S_small in (([2], S_max), ([2, 0], S_max), ([1, 4], 128)):
        kk = ops.pack_k_cache(kc[rows, :, :S_small].contiguous())
        vv = ops.pack_vt_cache(vt[rows, :, :, :S_small].contiguous())
        oo = torch.zeros(len(rows), Hq * D, device="cuda", dtype=dt)
        ops.decode_attn_rope(qkv[rows].contiguous(), cs[rows].contiguous(), slot_t[rows].contiguous(), kk, vv, oo, None, Hq, Hkv, D, S_small, S_small,
                             cache_packed=True)
        assert torch.equal(oo, o2[rows]), f"rows {rows} at capacity {S_small}: the output depends on the batch or on S_max"
    outp = torch.zeros(16, Hq * D, device="cuda", dtype=dt)
    ops.decode_attn_rope(qkv, cs, slot_t, ops.pack_k_cache(kc), ops.pack_vt_cache(vt), outp, None, Hq, Hkv, D, S_max, S_max, out_packed=True, cache_packed=True)
    un = torch.zeros_like(o2)
    ops.pack_rows(outp, un, B, to_packed=False)
    assert torch.equal(un, o2)


# ------------------------------------------------------------------------------------------------------------ row kernels
def test_rmsnorm_layernorm(ops):
    g = ops.rmsnorm(rnd(5, 1280, seed=47), (1 + 0.1 * rnd(1280, seed=41).float()).to(BF), gelu=True)
    gx = rnd(5, 1280, seed=47).float()
    close_bf16(g, torch.nn.functional.gelu((1 + 0.1 * rnd(1280, seed=41).float()).to(BF).float() * gx *
                                           torch.rsqrt(gx.pow(2).mean(-1, keepdim=True) + 1e-6)), "rmsnorm+gelu")
    x, w, b = rnd(37, 1280, seed=40), (1 + 0.1 * rnd(1280, seed=41).float()).to(BF), rnd(1280, seed=42)
    xf = x.float()
    close_bf16(ops.rmsnorm(x, w), w.float() * xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6), "rmsnorm")
    low = rnd(10, 1280, seed=43)
    y = ops.rmsnorm(x[:37], w, add=low, add_div=4)
    s = xf + low.float().repeat_interleave(4, 0)[:37]
    close_bf16(y, w.float() * s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6), "rmsnorm+repeat4 add")
    x2 = rnd(9, 2048, seed=44)
    close_bf16(ops.layernorm(x2, rnd(2048, seed=45), rnd(2048, seed=46)),
               torch.nn.functional.layer_norm(x2.float(), (2048,), rnd(2048, seed=45).float(), rnd(2048, seed=46).float(), 1e-5), "layernorm")


def test_rope_half_gather_add_cast(ops):
    T, H, D = 50, 32, 80
    x = rnd(T, 3 * 16 * 80, seed=50)
    ang = torch.rand(T, 40, device="cuda") * 50
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().contiguous(), emb.sin().contiguous()
    ref = x.float().clone()
    v = ref[:, : H * D].view(T, H, D)
    rot = torch.cat([-v[..., 40:], v[..., :40]], -1)
    ref[:, : H * D] = (v * cos[:, None] + rot * sin[:, None]).reshape(T, -1)
    y = x.clone()
    ops.rope_half_(y, cos, sin, H, D)
    close_bf16(y, ref, "rope_half")
    idx = torch.randint(0, T, (77,), device="cuda", dtype=torch.int32)
    assert torch.equal(ops.gather_rows(x, idx), x[idx.long()])
    assert torch.equal(ops.gather_rows(cos, idx), cos[idx.long()])
    a, b = rnd(24, 1280, seed=51), rnd(8, 1280, seed=52)
    close_bf16(ops.add_rows(a, b), a.float() + b.float().repeat(3, 1), "add_rows")
    f = torch.randn(33, 1176, device="cuda")
    c = ops.cast_f32_bf16(f, 1184)
    assert torch.equal(c[:, :1176], f.to(BF)) and (c[:, 1176:] == 0).all()
    s = torch.randn(12, device="cuda")
    close_f32(ops.sigmoid_f32_(s.clone()), torch.sigmoid(s), "sigmoid")


def test_embed_tokens(ops):
    V, NP, D, T = 1000, 37, 256, 64
    E, P, I = rnd(V, D, seed=60), rnd(NP, D, seed=61), rnd(20, D, seed=62)
    ids = torch.randint(0, V + NP, (T,), device="cuda")
    img = torch.full((T,), -1, dtype=torch.int32, device="cuda")
    img[5:25] = torch.arange(20, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = ops.embed_tokens(ids, img, E, P, I, err_flag=err)
    ref = torch.cat([E, P])[ids]
    ref[5:25] = I
    assert torch.equal(out, ref) and int(err) == 0
    ids[0] = V + NP
    ops.embed_tokens(ids, img, E, P, I, err_flag=err)
    assert int(err) == 1


def test_llm_qkv_post(ops):
    Hq, Hkv, D, S = 16, 2, 128, 128
    B, T = 2, 11
    qkv = rnd(T, (Hq + 2 * Hkv) * D, seed=70)
    pos = torch.randint(0, 600, (3, T), dtype=torch.int32, device="cuda")
    sample = torch.tensor([0] * 6 + [1] * 5, dtype=torch.int32, device="cuda")
    slot = torch.tensor(list(range(6)) + list(range(5)), dtype=torch.int32, device="cuda")
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float) / D))).cuda()
    q_out = torch.zeros(T, Hq * D, device="cuda", dtype=BF)
    kp = torch.zeros(T, Hkv * D, device="cuda", dtype=BF)
    kc = torch.zeros(B, Hkv, S, D, device="cuda", dtype=BF)
    vt = torch.zeros(B, Hkv, D, S, device="cuda", dtype=BF)
    ops.llm_qkv_post(qkv, pos, inv, q_out, kc, vt, Hq, Hkv, D, S, (16, 24, 24), sample=sample, slot=slot, k_pack=kp)
    fr = pos.float()[..., None] * inv                                  # 3,T,64
    emb = torch.cat([fr, fr], -1)
    sec = [16, 24, 24] * 2
    cos = torch.cat([c[i % 3] for i, c in enumerate(emb.cos().split(sec, -1))], -1)   # T,128
    sin = torch.cat([c[i % 3] for i, c in enumerate(emb.sin().split(sec, -1))], -1)

    def rope(x):
        x = x.float()
        return x * cos[:, None] + torch.cat([-x[..., 64:], x[..., :64]], -1) * sin[:, None]
    qr = rope(qkv[:, : Hq * D].view(T, Hq, D))
    kr = rope(qkv[:, Hq * D: (Hq + Hkv) * D].view(T, Hkv, D))
    vv = qkv[:, (Hq + Hkv) * D:].view(T, Hkv, D)
    close_bf16(q_out, qr.reshape(T, -1), "mrope q")
    close_bf16(kp, kr.reshape(T, -1), "mrope k")
    for t in range(T):
        b, s = int(sample[t]), int(slot[t])
        assert torch.equal(kc[b, :, s], kp[t].view(Hkv, D))
        assert torch.equal(vt[b, :, :, s], vv[t])


def test_mask_scatter(ops):
    dm, n_obj = 80, 3
    grids = [(6, 8), (6, 8), (4, 4)]
    pn = [h * w for h, w in grids]
    cu = [0]
    for p in pn:
        cu.append(cu[-1] + p)
    N = cu[-1]
    e2 = rnd(4 * N, 4 * dm, seed=80)
    tok = rnd(n_obj, dm, seed=81)
    Hm, Wm = 6, 8
    masks = torch.zeros(n_obj, 4 * Hm, 4 * Wm, device="cuda")
    ops.mask_scatter(e2, tok, torch.tensor(cu, dtype=torch.int32, device="cuda"),
                     torch.tensor([w for _, w in grids], dtype=torch.int32, device="cuda"), masks, n_obj, N, dm)
    ref = torch.zeros_like(masks)
    e = e2.float().view(N, 2, 2, 2, 2, dm)                           # n, a, b, c, d, k
    for o in range(n_obj):
        W = grids[o][1]
        for pi in range(pn[o]):
            n = cu[o] + pi
            r, c_ = pi // W, pi % W
            lg = (e[n] * tok[o].float()).sum(-1)                     # a,b,c,d
            blk = lg.permute(0, 2, 1, 3).reshape(4, 4)               # (a,c),(b,d)
            ref[o, 4 * r: 4 * r + 4, 4 * c_: 4 * c_ + 4] = blk
    close_f32(masks, ref, "mask_scatter", rel=1e-5)


@pytest.mark.parametrize("B", [1, 8, 20])
def test_vrt_head_and_greedy(ops, B):
    V, D, per = 3001, 256, 37
    NP = B * per
    E, P, h = rnd(V, D, seed=90), rnd(NP, D, seed=91), rnd(B, D, seed=92)
    off = torch.arange(0, NP + 1, per, dtype=torch.int32, device="cuda")
    nblk = ops.vrt_head_nblk(V, NP)
    pv = torch.empty(nblk * B, device="cuda")
    pi = torch.empty(nblk * B, dtype=torch.int32, device="cuda")
    logits = torch.zeros(B, V + NP, device="cuda")
    T_max, eos, pad = 6, 17, 3
    modes = torch.tensor([0, 1, 2, 3, 0, 0], dtype=torch.int32, device="cuda")
    full = h.float() @ torch.cat([E, P]).float().T
    allow = torch.zeros(B, V + NP, dtype=torch.bool, device="cuda")
    allow[:, :V] = True
    for b in range(B):
        allow[b, V + b * per: V + (b + 1) * per] = True
    unfinished = torch.ones(B, dtype=torch.int32, device="cuda")
    tokens = torch.zeros(B, T_max, dtype=torch.int64, device="cuda")
    cur = torch.zeros(B, dtype=torch.int64, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    slot = torch.full((B,), 5, dtype=torch.int32, device="cuda")
    lens = torch.full((B,), 6, dtype=torch.int32, device="cuda")
    pos3 = torch.full((3, B), 9, dtype=torch.int32, device="cuda")
    hbuf = torch.zeros(T_max, B, D, device="cuda", dtype=BF)
    exp_unf = torch.ones(B, dtype=torch.bool, device="cuda")
    for s in range(4):
        ops.vrt_head(h, E, P, off, pv, pi, eos, mode_table=modes, step=step, logits=logits)
        m = allow.clone()
        if s == 1:
            m[:, V:] = False
        elif s == 2:
            m[:, :V] = False
        elif s == 3:
            m[:] = False
            m[:, eos] = True
        ref = full.masked_fill(~m, float("-inf"))
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isfinite(logits), fin), f"mask pattern step {s}"
        close_f32(logits[fin], ref[fin], "head logits", rel=2e-5)
        ops.greedy_step(pv, pi, nblk, h, hbuf, unfinished, tokens, cur, step, slot, lens, pos3, eos, pad)
        want = logits.argmax(-1)                                     # torch.argmax: first max
        want = torch.where(exp_unf, want, torch.full_like(want, pad))
        exp_unf = exp_unf & (want != eos)
        assert torch.equal(tokens[:, s], want), f"tokens step {s}"
        assert torch.equal(cur, want) and int(step) == s + 1
        assert torch.equal(unfinished.bool(), exp_unf)
        assert torch.equal(hbuf[s], h)
    assert int(slot[0]) == 9 and int(lens[0]) == 10 and int(pos3[2, B - 1]) == 13
    assert not exp_unf.any()                                        # step 3 forced EOS everywhere


@pytest.mark.parametrize("B", [5, 20, 40])
def test_vrt_head_packed_table_is_bit_identical(ops, B):
    """Fragment-packed text table + packed hidden rows == the row-major call: same partial (max, argmax) and logits."""
    V, D, per = 3008, 256, 23                                      # V % 16 == 0 (the packed path's requirement)
    NP = B * per
    E, P, h = rnd(V, D, seed=95), rnd(NP, D, seed=96), rnd(B, D, seed=97)
    off = torch.arange(0, NP + 1, per, dtype=torch.int32, device="cuda")
    nblk = ops.vrt_head_nblk(V, NP)
    res = []
    B16 = (B + 15) // 16 * 16
    hp = torch.zeros(B16, D, device="cuda", dtype=BF)
    ops.pack_rows(h, hp, B, to_packed=True)
    Ep = ops.pack_weight(E)
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
    close_f32(res[1][2][fin], full[fin], "packed head logits", rel=2e-5)


def test_vrt_head_tie_breaks_to_lowest_index(ops):
    V, D = 64, 64
    E = torch.zeros(V, D, device="cuda", dtype=BF)
    E[5, 0] = 1.0
    E[40, 0] = 1.0                                                  # exact tie between rows 5 and 40 (different blocks)
    P = torch.zeros(16, D, device="cuda", dtype=BF)
    h = torch.zeros(1, D, device="cuda", dtype=BF)
    h[0, 0] = 2.0
    off = torch.tensor([0, 16], dtype=torch.int32, device="cuda")
    nblk = ops.vrt_head_nblk(V, 16)
    pv, pi = torch.empty(nblk, device="cuda"), torch.empty(nblk, dtype=torch.int32, device="cuda")
    ops.vrt_head(h, E, P, off, pv, pi, eos=1)
    st = [torch.ones(1, dtype=torch.int32, device="cuda"), torch.zeros(1, 2, dtype=torch.int64, device="cuda"),
          torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")]
    z = torch.zeros(3, dtype=torch.int32, device="cuda")
    ops.greedy_step(pv, pi, nblk, h, torch.zeros(2, 1, D, device="cuda", dtype=BF), st[0], st[1], st[2], st[3],
                    z[:1].clone(), z[:1].clone(), z.clone(), 1, 0)
    assert int(st[2]) == 5


@pytest.mark.parametrize("D,H", [(80, 4), (128, 2), (32, 3)])
def test_attn_varlen_fused_rope_equals_rope_then_attention(ops, D, H):
    """RoPE fused into the (window) attention kernel == rope_half on q,k followed by the plain kernel; ragged windows."""
    lens = [64, 48, 36, 64, 7, 130]
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    qkv = rnd(T, 3 * H * D, seed=91)
    ang = torch.rand(T, D // 2, device="cuda") * 40
    emb = torch.cat([ang, ang], -1)
    cos, sin = emb.cos().contiguous(), emb.sin().contiguous()
    vh = H * D
    ref_in = qkv.clone()
    ops.rope_half_(ref_in, cos, sin, 2 * H, D)
    ref = torch.zeros(T, vh, device="cuda", dtype=BF)
    ops.attn_varlen(ref_in[:, :vh], ref_in[:, vh:2 * vh], ref_in[:, 2 * vh:], ref, cu, cu, max(lens), H, H, D)
    got = torch.zeros_like(ref)
    ops.attn_varlen(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], got, cu, cu, max(lens), H, H, D, rope=(cos, sin))
    assert torch.equal(got, ref), f"fused rope differs: max |d| {(got.float() - ref.float()).abs().max().item():.3e}"
    with pytest.raises(Exception, match="fused RoPE"):
        ops.attn_varlen(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], got, cu, cu, 300, H, H, D, rope=(cos, sin))


def test_mask_upsample_binarize_against_reference_expression(ops):
    """padt_mask_upsample_binarize vs F.interpolate(bilinear).sigmoid() > 0.5 on the golden inputs + a ragged extra case:
    up-sampled logits to fp32 rounding, binary masks identical wherever the logit is not within rounding of the threshold."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.npz"))
    masks = torch.from_numpy(z["masks"]).cuda()
    sizes = [tuple(int(v) for v in r) for r in z["image_sizes"]]
    sidx = z["sample_idx"].tolist()
    hs = torch.from_numpy(z["valid_h"]).to(torch.int32).cuda() * 4
    ws = torch.from_numpy(z["valid_w"]).to(torch.int32).cuda() * 4
    dh = torch.tensor([sizes[s][1] for s in sidx], dtype=torch.int32, device="cuda")
    dw = torch.tensor([sizes[s][0] for s in sidx], dtype=torch.int32, device="cuda")
    out, up = ops.mask_upsample_binarize(masks, hs, ws, dh, dw, int(dh.max()), int(dw.max()), want_logits=True)
    o, n_flip = 0, 0
    for i, s in enumerate(sidx):
        w, h = sizes[s]
        ref_up = torch.nn.functional.interpolate(masks[i][None, None, : int(hs[i]), : int(ws[i])].cpu(), size=(h, w), mode="bilinear")[0, 0]
        exp_up = torch.from_numpy(z["exp_up"][o:o + h * w]).view(h, w)
        o += h * w
        assert torch.equal(ref_up, exp_up)                             # same torch → the fixture is what the expression gives
        got_up = up[i, :h, :w].cpu()
        assert (got_up - ref_up).abs().max().item() <= 2e-6 * ref_up.abs().max().item() + 1e-6
        ref_bin = (ref_up.sigmoid() > 0.5)
        diff = (out[i, :h, :w].cpu().bool() != ref_bin)
        assert (ref_up.abs()[diff] < 1e-5).all(), "binary mask differs away from the threshold"
        n_flip += int(diff.sum())
        assert (out[i, h:, :] == 0).all() and (out[i, :, w:] == 0).all()
    assert n_flip <= 2
    # down-sampling and non-square, single object
    m = torch.randn(1, 40, 24, device="cuda")
    one = lambda v: torch.tensor([v], dtype=torch.int32, device="cuda")
    o2, u2 = ops.mask_upsample_binarize(m, one(37), one(21), one(19), one(50), 19, 50, want_logits=True)
    r2 = torch.nn.functional.interpolate(m[:, None, :37, :21].cpu(), size=(19, 50), mode="bilinear")[0, 0]
    assert (u2[0].cpu() - r2).abs().max().item() < 1e-5
    d2 = o2[0].cpu().bool() != (r2.sigmoid() > 0.5)
    assert (r2.abs()[d2] < 1e-5).all()


def test_patchify_normalize_matches_hf_processor_bit_exact(ops):
    """padt_patchify_normalize vs the HF Qwen2-VL PIL processor's pixel_values (fixture): fp32 bit-exact, bf16 = its rounding;
    ImageFrontEnd end to end (host resize skipped: inputs are already at their smart_resize size)."""
    import numpy as np
    from padt_amd import preprocess as P
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.npz"))
    lut = torch.from_numpy(P.normalize_lut()).cuda()
    exp = torch.from_numpy(z["pix"])
    o = 0
    for k in ("img0", "img1"):
        img = torch.from_numpy(z[k]).cuda()
        n = (img.shape[0] // 14) * (img.shape[1] // 14)
        out32 = torch.zeros(n, 1176, device="cuda")
        ops.patchify_normalize(img, lut, out32)
        assert torch.equal(out32.cpu(), exp[o:o + n])
        out16 = torch.zeros(n, 1176, device="cuda", dtype=BF)
        ops.patchify_normalize(img, lut, out16)
        assert torch.equal(out16.cpu(), exp[o:o + n].to(BF))
        o += n
    fe = P.ImageFrontEnd("cuda", dtype=torch.float32)
    pix, grid = fe([z["img0"], z["img1"]])
    assert torch.equal(pix.cpu(), exp) and grid.tolist() == z["grid"].tolist()
    with pytest.raises(Exception, match="multiples of patch"):
        ops.patchify_normalize(torch.zeros(30, 28, 3, dtype=torch.uint8, device="cuda"), lut, torch.zeros(4, 1176, device="cuda"))


def test_gpu_resize_is_byte_exact_with_pillow_and_front_end_matches_pil_path(ops):
    """csrc/resize.hip = Pillow's ImagingResample on the device: byte-identical to PIL.Image.resize (BICUBIC as the HF processor calls
    it, LANCZOS as eval/test_demo.py:73 / utils.py:217 do), down- and up-scaling; the whole front-end (callers' LANCZOS rule →
    smart_resize → bicubic → rescale/normalize/patchify) on the GPU gives bit-identical pixel_values to the host-PIL path, for RGB,
    gray ('L'), palette and RGBA inputs of ragged sizes."""
    import numpy as np
    from PIL import Image
    from padt_amd import preprocess as P
    rng = np.random.default_rng(5)
    fe = P.ImageFrontEnd("cuda", dtype=torch.float32)
    for (H, W, oh, ow) in [(480, 640, 476, 644), (333, 500, 644, 448), (100, 37, 28, 56), (1200, 900, 644, 476), (20, 300, 28, 420)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for name, pf in (("bicubic", Image.BICUBIC), ("lanczos", Image.LANCZOS)):
            ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=pf))
            got = fe.resize_device(torch.from_numpy(img).cuda(), ow, oh, name).cpu().numpy()
            assert np.array_equal(got, ref), (H, W, oh, ow, name, int((got != ref).sum()))
    images = [Image.fromarray(rng.integers(0, 256, (427, 640, 3), dtype=np.uint8)),
              Image.fromarray(rng.integers(0, 256, (500, 333), dtype=np.uint8), mode="L"),
              Image.fromarray(rng.integers(0, 256, (96, 20, 3), dtype=np.uint8)).convert("P"),
              Image.fromarray(rng.integers(0, 256, (640, 480, 4), dtype=np.uint8), mode="RGBA"),
              rng.integers(0, 256, (1100, 1700, 3), dtype=np.uint8)]
    for pre in (None, "demo644", "min28", "demo"):
        gpu = P.ImageFrontEnd("cuda", dtype=torch.float32, resize="gpu", pre_resize=pre)
        pil = P.ImageFrontEnd("cuda", dtype=torch.float32, resize="pil", pre_resize=pre)
        pg, gg = gpu(images)
        pp, gp = pil(images)
        assert gg.tolist() == gp.tolist() and torch.equal(pg, pp), pre
    # and against the HF processor's own output (fixture generated from the installed transformers PIL processor, which resizes itself)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.npz"))
    if "raw0" in z.files:
        pix, grid = fe([z["raw0"], z["raw1"]])
        assert torch.equal(pix.cpu(), torch.from_numpy(z["pix_raw"])) and grid.tolist() == z["grid_raw"].tolist()


@pytest.mark.parametrize("T,top_k,top_p", [(1.0, 0, 1.0), (0.7, 40, 1.0), (1.3, 50, 0.9), (1.0, 3, 0.5)])
def test_sample_token_distribution(ops, T, top_k, top_p):
    """Sampling branch (padt.py:740-743 + HF warpers): the device draws are not torch.multinomial's, so the test is distribution-level —
    131 072 draws from one logit row; support == the oracle's warped support exactly (top-k ties and the nucleus rule included),
    empirical frequencies within 5 sigma of softmax(warped) on every kept token."""
    import padt_oracle as O
    n, B, reps = 3000, 64, 2048
    g = torch.Generator().manual_seed(7)
    row = torch.randn(n, generator=g) * 2.0
    row[100] = row[7]                                             # a tie inside the interesting range
    row[5] = float("-inf")                                        # masked rows stay impossible
    logits = row[None, :].repeat(B, 1).cuda()
    exp = O.warp_logits(row[None, :], T, top_k, top_p)[0]
    p_ref = exp.softmax(-1)
    counts = torch.zeros(n, dtype=torch.long, device="cuda")
    pv = torch.zeros(B, device="cuda")
    pi = torch.zeros(B, dtype=torch.int32, device="cuda")
    cfg = ops.gen_cfg_tensor(1.0, (), "cuda", do_sample=True, seed=1234, temperature=T, top_k=top_k, top_p=top_p)
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    for r in range(reps):
        step.fill_(r)
        ops.sample_token(logits, n, cfg, step, pv, pi, B)
        counts += torch.bincount(pi.long(), minlength=n)
    N = B * reps
    freq = counts.cpu().double() / N
    kept = torch.isfinite(exp)
    assert int(counts.cpu()[~kept].sum()) == 0, "a token outside the warped support was drawn"
    pr = p_ref.double()
    common = kept & (pr * N >= 50)                                # normal approximation holds: per-token 5-sigma test
    sigma = (pr * (1 - pr) / N).sqrt()
    dev = ((freq - pr).abs() / (sigma + 1e-12))[common]
    assert dev.max().item() < 5.0, f"worst deviation {dev.max().item():.2f} sigma"
    rare = kept & ~common                                         # rare tokens: their total mass, same bound
    if bool(rare.any()):
        pm, fm = pr[rare].sum().item(), freq[rare].sum().item()
        assert abs(fm - pm) < 5.0 * math.sqrt(pm * (1 - pm) / N) + 1e-9, f"rare-token mass {fm:.5f} vs {pm:.5f}"
    if top_k == 0:
        assert int((counts > 0).sum()) > 500                      # the tail is really sampled
    # same (seed, step) → same draw; another seed → another sequence of draws
    step.fill_(3)
    ops.sample_token(logits, n, cfg, step, pv, pi, B)
    a = pi.clone()
    ops.sample_token(logits, n, cfg, step, pv, pi, B)
    assert torch.equal(a, pi)
    cfg2 = ops.gen_cfg_tensor(1.0, (), "cuda", do_sample=True, seed=99, temperature=T, top_k=top_k, top_p=top_p)
    ops.sample_token(logits, n, cfg2, step, pv, pi, B)
    if top_k != 3:
        assert not torch.equal(a, pi)


@pytest.mark.parametrize("M,N,K", [(8, 2560, 2048), (1, 2048, 3584), (40, 704, 512), (64, 2048, 2080), (16, 22016, 2048), (33, 96, 256)])
def test_gemm_packed_fp8_weights(ops, M, N, K):
    """fp8 (OCP e4m3, power-of-two row scales) decode projections: the kernel converts the bytes to bf16 fragments in registers
    (exact) and scales the fp32 accumulator — against fp32 statements on the DEQUANTISED matrix: same tolerance as the bf16 kernel,
    and bit-identical to the bf16 packed kernel run on the dequantised matrix (same fragments, same accumulation order)."""
    x = rnd(M, K, seed=81)
    w, b, r = rnd(N, K, scale=0.05, seed=82), rnd(N, seed=83), rnd(M, N, seed=84)
    w[3] *= 40.0                                                   # rows of very different magnitude → different scales
    w[5] *= 0.001
    q, sc, deq = ops.quantize_fp8_rows(w)
    assert torch.equal((q.view(torch.float8_e4m3fn).float() * sc[:, None]).to(BF), deq)
    assert torch.equal(torch.exp2(torch.round(torch.log2(sc))), sc)                # powers of two
    rel_q = ((deq.float() - w.float()).abs() / (w.float().abs().amax(1, keepdim=True) + 1e-30)).max().item()
    assert rel_q < 2 ** -4                                                          # 3 mantissa bits, scale wastes < 1 bit of range
    wq = ops.pack_weight_fp8(q)
    wp = ops.pack_weight(deq)
    xf = x.float()
    rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    lin = xf @ deq.float().T
    got = ops.gemm_packed_fp8(x, wq, sc, N, b, norm_eps=1e-6)
    close_bf16(got, lin * rstd + b.float(), f"fp8 norm+bias {M}x{N}x{K}")
    assert torch.equal(got, ops.gemm_packed(x, wp, N, b, norm_eps=1e-6)), "fp8 path != bf16 path on the dequantised matrix"
    out = r.clone()
    ops.gemm_packed_fp8(x, wq, sc, N, out=out, epilogue=ops.EPI_RESID, residual=out)
    close_bf16(out, lin + r.float(), f"fp8 resid {M}x{N}x{K}")
    if N >= 256:
        ws = ops.new_splitk_workspace(N, 2, "cuda")
        close_bf16(ops.gemm_packed_fp8(x, wq, sc, N, b, split_k=2, workspace=ws), lin + b.float(), "fp8 split-K")
    M16 = (M + 15) // 16 * 16
    xp = torch.zeros(M16, K, device="cuda", dtype=BF)
    ops.pack_rows(x, xp, M, to_packed=True)
    if N % 32 == 0:
        hp = torch.zeros(M16, N // 2, device="cuda", dtype=BF)
        ops.gemm_packed_fp8(xp, wq, sc, N, b, out=hp, epilogue=ops.EPI_SWIGLU, norm_eps=1e-6, a_packed=True, c_packed=True, rows=M)
        un = torch.zeros(M, N // 2, device="cuda", dtype=BF)
        ops.pack_rows(hp, un, M, to_packed=False)
        y = (lin * rstd + b.float()).view(M, N // 32, 2, 16)
        close_bf16(un, (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, N // 2), f"fp8 SwiGLU {M}x{N}x{K}")


# ------------------------------------------------------------------------------------------------ fp32 residual stream
@pytest.mark.parametrize("M,N,K", [(1000, 1280, 1280), (16928 // 8, 1280, 3456), (577, 2048, 2048), (300, 200, 136), (40, 640, 512), (2 * 256 + 24, 768, 640)])
def test_gemm_resid32_stream_and_mirror(ops, M, N, K, knobs):
    """padt_gemm_resid32 on every kernel family (256-row phase kernel incl. the wide mirror write-out, 128^2 kernel with ragged tiles,
    skinny kernel, peeled tail): x32 += a w^T + b within fp32 accumulation noise, the mirror is EXACTLY bf16(x32) (one rounding)."""
    if M == 2 * 256 + 24:
        knobs(peel=2, mf=4)
    a, w, b = rnd(M, K, seed=91), rnd(N, K, scale=0.05, seed=92), rnd(N, seed=93)
    x0 = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(94)) * 3
    ld = (N + 7) // 8 * 8
    x32 = torch.zeros(M, ld, device="cuda")
    x32[:, :N] = x0
    xb = torch.full((M, ld), 7.0, device="cuda", dtype=BF)
    ops.gemm_resid32(a, w, b, x32[:, :N], xb[:, :N])
    ref = x0 + a.float() @ w.float().T + b.float()
    close_f32(x32[:, :N], ref, f"resid32 {M}x{N}x{K}", rel=3e-5)
    assert torch.equal(xb[:, :N], x32[:, :N].to(BF)), "mirror != bf16(stream)"
    if ld != N:
        assert (xb[:, N:] == 7.0).all() and (x32[:, N:] == 0).all(), "wrote outside N"
    y = x32.clone()
    ops.gemm_resid32(a, w, None, y[:, :N], None)                       # no mirror, no bias
    close_f32(y[:, :N], x32[:, :N] + a.float() @ w.float().T, "resid32 without mirror", rel=3e-5)


@pytest.mark.parametrize("M,N,K", [(8, 2048, 2048), (64, 2048, 11008), (21, 704, 512), (128, 3584, 3584)])
def test_gemm_packed_resid32(ops, M, N, K):
    """Decode-step residual projection over the fp32 stream: same accumulator bits as the bf16-stream kernel (the fp32 result rounds to
    what that kernel stores when the residual is bf16-representable), packed mirror == bf16(stream), bf16 and fp8 weights, split-K."""
    x = rnd(M, K, seed=95)
    w, r = rnd(N, K, scale=0.05, seed=96), rnd(M, N, seed=97)
    wp = ops.pack_weight(w)
    M16 = (M + 15) // 16 * 16
    xp = torch.zeros(M16, K, device="cuda", dtype=BF)
    ops.pack_rows(x, xp, M, to_packed=True)
    ws = ops.new_splitk_workspace(N, 2, "cuda")
    old = r.clone()
    ops.gemm_packed(x, wp, N, out=old, epilogue=ops.EPI_RESID, residual=old)
    for split in (1, 2):
        x32 = r.float().contiguous()
        mir = torch.zeros(M16, N, device="cuda", dtype=BF)
        ops.gemm_packed_resid32(xp, wp, N, x32, mir, split_k=split, workspace=ws, rows=M)
        close_f32(x32, r.float() + x.float() @ w.float().T, f"packed resid32 split {split}", rel=3e-5)
        un = torch.zeros(M, N, device="cuda", dtype=BF)
        ops.pack_rows(mir, un, M, to_packed=False)
        assert torch.equal(un, x32.to(BF)), "packed mirror != bf16(stream)"
        if split == 1:
            assert torch.equal(un, old), "fp32-stream kernel and bf16-stream kernel disagree on the same operands"
    q, sc, deq = ops.quantize_fp8_rows(w)
    if N % 16 == 0 and K % 8 == 0:
        x32 = r.float().contiguous()
        mir = torch.zeros(M16, N, device="cuda", dtype=BF)
        ops.gemm_packed_resid32(xp, ops.pack_weight_fp8(q), N, x32, mir, scales=sc, rows=M)
        close_f32(x32, r.float() + x.float() @ deq.float().T, "packed resid32 fp8", rel=3e-5)


def test_cast_bf16_f32_and_rmsnorm_f32(ops):
    x = rnd(77, 2048, seed=98)
    assert torch.equal(ops.cast_bf16_f32(x), x.float())
    x32 = torch.randn(77, 1280, device="cuda") * 2
    w = rnd(1280, seed=99) * 0.1 + 1
    ref = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
    close_bf16(ops.rmsnorm_f32(x32, w), ref, "rmsnorm_f32")


# ------------------------------------------------------------------------------------------------ fp8 x fp8 MFMA GEMM
def _dequant8(q, scale):
    return q.view(torch.float8_e4m3fn).float() * scale[:, None]


@pytest.mark.parametrize("M,K", [(77, 2048), (300, 18944), (16, 128)])
def test_quant_rows_fp8_matches_the_host_statement(ops, M, K):
    """padt_quant_rows_fp8 == ops.quantize_fp8_rows's rule applied to activation rows: power-of-two scale 2^ceil(log2(amax / 448)),
    e4m3 codes bit for bit; with norm_eps the row scale carries rsqrt(mean(x^2) + eps)."""
    x = rnd(M, K, seed=201)
    x[3] *= 50
    x[5] = 0
    q_ref, sc_ref, _ = ops.quantize_fp8_rows(x)
    sc_ref = torch.where(x.float().abs().amax(1) > 0, sc_ref, torch.ones_like(sc_ref))
    q, rs = ops.quant_rows_fp8(x)
    assert torch.equal(rs, sc_ref)
    assert torch.equal(q[x.float().abs().amax(1) > 0], q_ref[x.float().abs().amax(1) > 0]) and ((q[5] & 0x7F) == 0).all()
    q2, rs2 = ops.quant_rows_fp8(x, norm_eps=1e-6)
    assert torch.equal(q2, q)
    rstd = torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6)
    assert ((rs2 - sc_ref * rstd).abs() <= 2e-6 * (sc_ref * rstd).abs()).all()


@pytest.mark.parametrize("M,N,K", [(1000, 512, 256), (4616, 2560, 2048), (577, 4608, 3584), (300, 768, 1280)])
@pytest.mark.parametrize("mf", [0, 2, 3, 4])
def test_gemm_fp8_mfma_against_fp32_on_the_dequantised_operands(ops, M, N, K, mf, knobs):
    """padt_gemm_fp8 (v_mfma_f32_16x16x128_f8f6f4 in the 256-row tile kernel, every tile height) against the fp32 GEMM of the DEQUANTISED
    operands — plain (+bias), SwiGLU, and the fp32 residual stream with its bf16 mirror.  The fp8 MFMA does not sum its 128 products in full
    fp32: measured ≈2^-15 of the magnitude sum of a dot product (tools/ubench/f8probe.hip: 4e-4 at |c| = 11), so the fp32 output is held to
    3e-5 x (|A| · |W|^T) element-wise; bf16 outputs to the usual one-rounding tolerance."""
    if mf:
        knobs(mf=mf)
    x = rnd(M, K, seed=211)
    w = rnd(N, K, scale=0.05, seed=212)
    b = rnd(N, seed=213)
    w[7] *= 30.0
    a8, rs = ops.quant_rows_fp8(x)
    w8, ws, _ = ops.quantize_fp8_rows(w)
    ref = _dequant8(a8, rs) @ _dequant8(w8, ws).T
    out = ops.gemm_fp8(a8, w8, ws, rs, bias=b)
    close_bf16(out, ref + b.float(), f"fp8 plain {M}x{N}x{K}")
    x32 = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    x0 = x32.clone()
    xb = torch.zeros(M, N, device="cuda", dtype=BF)
    ops.gemm_fp8(a8, w8, ws, rs, epilogue=ops.EPI_RESID, x32=x32, xb=xb)
    mag = _dequant8(a8, rs).abs() @ _dequant8(w8, ws).abs().T
    bad = (x32 - (x0 + ref)).abs() > 3e-5 * mag + 1e-6
    assert not bad.any(), f"fp8 resid32 {M}x{N}x{K}: {int(bad.sum())} outside 3e-5 of the magnitude sum, worst {((x32 - x0 - ref).abs() / (mag + 1e-9)).max().item():.2e}"
    assert torch.equal(xb, x32.to(BF))
    wi = interleave_gate_up(w[: N // 2].contiguous(), w[N // 2:].contiguous())
    bi = interleave_gate_up(b[: N // 2].reshape(-1, 1), b[N // 2:].reshape(-1, 1)).view(-1)
    wi8, wis, _ = ops.quantize_fp8_rows(wi)
    a8n, rsn = ops.quant_rows_fp8(x, norm_eps=1e-6)
    lin = (_dequant8(a8n, torch.ones_like(rsn)) @ _dequant8(wi8, wis).T) * rsn[:, None] + bi.float()
    y = lin.view(M, N // 32, 2, 16)
    close_bf16(ops.gemm_fp8(a8n, wi8, wis, rsn, bias=bi, epilogue=ops.EPI_SWIGLU), (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, N // 2),
               f"fp8 swiglu {M}x{N}x{K}")
    with pytest.raises(Exception, match="padt_gemm_fp8"):
        ops.gemm_fp8(a8[:, :64], w8[:, :64], ws, rs)


def test_mask_rle_on_the_device_is_the_host_statement_byte_for_byte(ops):
    """padt_mask_rle (column-major run lengths + COCO rleToString in one block per object) against padt_amd.postprocess.rle_counts /
    rle_string — the restatement of cocoapi's rleEncode / rleToString that tests/golden/postprocess.npz pins (make_golden_post.py) — on the
    golden masks, on noise-like and blob-like masks of ragged sizes in ONE padded batch, and on the edge cases: all zeros, all ones, a leading
    one (zero-length first run), a single pixel, one column, one row, heights that need the narrower LDS strips."""
    import numpy as np
    from padt_amd import postprocess as P
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.npz"))
    sizes = [tuple(int(v) for v in r) for r in z["image_sizes"]]
    sidx = z["sample_idx"].tolist()
    lens = z["exp_bits_len"].tolist()
    cases, exp_golden, o = [], [], 0
    for i, s in enumerate(sidx):
        w, h = sizes[s]
        cases.append(np.unpackbits(z["exp_bits"][o:o + lens[i]])[: h * w].reshape(h, w))
        o += lens[i]
        exp_golden.append(z["exp_str"].tolist()[i])
    rng = np.random.default_rng(7)
    for (h, w, p) in [(640, 640, 0.5), (640, 640, 0.03), (480, 640, 0.3), (37, 5, 0.5), (1, 97, 0.4), (97, 1, 0.4), (1, 1, 1.0), (1000, 130, 0.2), (2000, 40, 0.5)]:
        cases.append((rng.random((h, w)) < p).astype(np.uint8))
    yy, xx = np.mgrid[0:427, 0:640]
    cases.append((((yy - 200) ** 2 + (xx - 300) ** 2) < 150 ** 2).astype(np.uint8))        # a blob: long runs, multi-byte counts, negative differences
    cases += [np.zeros((33, 65), np.uint8), np.ones((33, 65), np.uint8), np.ones((640, 640), np.uint8)]
    lead = np.zeros((64, 64), np.uint8)
    lead[0, 0] = 1
    cases.append(lead)
    # groups of similar height share a launch (the LDS strip width follows the tallest mask of the launch)
    groups = {}
    for k, m in enumerate(cases):
        groups.setdefault(0 if m.shape[0] <= 700 else (1 if m.shape[0] <= 1200 else 2), []).append(k)
    got_s, got_c = {}, {}
    for ks in groups.values():
        mh, mw = max(cases[k].shape[0] for k in ks), max(cases[k].shape[1] for k in ks)
        buf = torch.zeros((len(ks), mh, mw), dtype=torch.uint8)
        for j, k in enumerate(ks):
            buf[j, : cases[k].shape[0], : cases[k].shape[1]] = torch.from_numpy(cases[k])
            buf[j, cases[k].shape[0]:, :] = 1                          # padding must not be read
            buf[j, :, cases[k].shape[1]:] = 1
        dh = torch.tensor([cases[k].shape[0] for k in ks], dtype=torch.int32, device="cuda")
        dw = torch.tensor([cases[k].shape[1] for k in ks], dtype=torch.int32, device="cuda")
        ss, cc = ops.mask_rle(buf.cuda(), dh, dw, want_counts=True)
        for j, k in enumerate(ks):
            got_s[k], got_c[k] = ss[j], cc[j]
    for k, m in enumerate(cases):
        counts = P.rle_counts(m)
        assert got_c[k] == counts, f"case {k} {m.shape}: counts differ ({len(got_c[k])} vs {len(counts)})"
        assert got_s[k] == P.rle_string(counts), f"case {k} {m.shape}: string differs"
        assert sum(got_c[k]) == m.size
    for k, s in enumerate(exp_golden):
        assert got_s[k] == s                                          # the reference-side fixture itself
    assert ops.mask_rle(torch.zeros((0, 4, 4), dtype=torch.uint8, device="cuda"), None, None) == []


def test_mask_rle_bounded_scratch_overflow_and_oversize_objects_are_reported(ops):
    """the per-call scratch is bounded (128 runs per column) and cached per shape — blob masks fit; a noise mask beyond it is REPORTED
    (fetch → None / PaDTHipError) and ops.mask_rle answers it with the worst-case capacities; an object taller than the launch's max_h or wider than a
    mask row is reported by the kernel instead of overrunning the LDS strip."""
    import numpy as np
    from padt_amd import _lib, postprocess as P
    yy, xx = np.mgrid[0:320, 0:400]
    blob = (((yy - 150) ** 2 + (xx - 200) ** 2) < 100 ** 2).astype(np.uint8)
    noise = (np.random.default_rng(3).random((320, 400)) < 0.5).astype(np.uint8)
    buf = torch.from_numpy(np.stack([blob, blob.T.copy().T])).cuda()
    dh = torch.tensor([320, 320], dtype=torch.int32, device="cuda")
    dw = torch.tensor([400, 400], dtype=torch.int32, device="cuda")
    h1 = ops.mask_rle_launch(buf, dh, dw)
    assert h1["counts"].shape[1] == 128 * 400 + 2                    # bounded, not 320 * 400 + 2
    s1 = ops.mask_rle_fetch(h1, on_overflow="none")
    assert s1 is not None and s1[0] == P.rle_string(P.rle_counts(blob))
    h2 = ops.mask_rle_launch(buf, dh, dw)
    assert h2["counts"].data_ptr() == h1["counts"].data_ptr()        # the cached scratch of this shape
    nb = torch.from_numpy(np.stack([blob, noise])).cuda()
    h3 = ops.mask_rle_launch(nb, dh, dw)
    assert ops.mask_rle_fetch(h3, on_overflow="none") is None
    with pytest.raises(_lib.PaDTHipError):
        ops.mask_rle_fetch(ops.mask_rle_launch(nb, dh, dw))
    assert ops.mask_rle(nb, dh, dw)[1] == P.rle_string(P.rle_counts(noise))      # worst-case capacities on the second attempt
    too_tall = torch.tensor([320, 321], dtype=torch.int32, device="cuda")
    h4 = ops.mask_rle_launch(buf, too_tall, dw)
    assert ops.mask_rle_fetch(h4, on_overflow="none") is None and h4["n_counts"].cpu().tolist()[1] == -1
    too_wide = torch.tensor([400, 401], dtype=torch.int32, device="cuda")
    h5 = ops.mask_rle_launch(buf, dh, too_wide)
    assert ops.mask_rle_fetch(h5, on_overflow="none") is None and h5["n_counts"].cpu().tolist()[1] == -1
