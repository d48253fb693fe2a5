// bf16 MFMA GEMMs for gfx950:  C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N])     (A, W K-contiguous, nn.Linear layout)
//
// Two kernels behind one C-ABI entry (padt_gemm_bf16):
//   * gemm_tile_kernel   — M > 64.  128x128x64 tiles, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 fragments.
//                          Both operands are staged HBM→LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//                          instruction), double-buffered, one barrier per K-step.  The LDS image is lane-linear, so the
//                          bank-conflict XOR swizzle is applied on the SOURCE address (16-byte chunk c of row r is
//                          stored in slot c^(r&7)) and again on the ds_read_b128 side.  K tails (K % 64 != 0) are read
//                          from a zero page.  Block ids are remapped so each XCD works on a contiguous chunk of tiles.
//   * gemm_skinny_kernel — M <= 64 (decode steps, decoder queries, heads): HBM-bound weight streaming.  Each block owns
//                          16 (or 32, SwiGLU) weight rows, its 4 waves interleave over K, every lane streams 16-byte
//                          pieces of W straight to VGPRs (no LDS round trip: W is read once), x comes from L1/L2.
// The MFMA is issued "swapped" (W fragment as the A operand) so a lane ends up with 4 CONSECUTIVE output columns of one
// row: 8-byte bf16 / 16-byte f32 stores, and bias / residual / SwiGLU pairs are lane-local.
//
// Replaces: every nn.Linear / Conv3d-as-GEMM on the path — HF ViT qkv/proj/MLP/merger, LLM q/k/v/o/gate/up/down,
// vis_proj (padt.py:189), PaDT decoder projections and heads (padt_decoder.py:15-18,82-86,142-184).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

extern "C" void padt_set_error(const char* msg);

// Measurement surface: while a slot array is registered, every tile-GEMM call (M > 64) of this process — of either operand-type
// instantiation — takes the next {start, end} slot and its kernels record their first block start / last block end in 100 MHz ticks.  The
// caller initialises the slots to {~0, 0}.  The call counter is atomic (lanes of a pipelined runner launch GEMMs from one host thread today,
// but nothing in the ABI says so).  Do not register while a graph is being captured: the slot pointer would be baked into it (the
// captured decode step has no tile GEMM).  bench.py's in-situ roofline leg only.
#include <atomic>
struct GemmProfState { std::atomic<unsigned long long*> slots{nullptr}; std::atomic<long> cap{0}, n{0}; };
#if !PADT_OP16_F16
GemmProfState padt_g_gemm_prof;
extern "C" long padt_gemm_profile(void* slots_u64, long capacity) {
    const long used = padt_g_gemm_prof.n.exchange(0);
    padt_g_gemm_prof.cap.store(slots_u64 ? capacity : 0);
    padt_g_gemm_prof.slots.store((unsigned long long*)slots_u64);
    return used;                                                  // calls recorded since the previous registration
}
extern "C" long padt_gemm_splitk_workspace(long N, int split_k) {
    const long ticket_bytes = (((N + 15) / 16 * 4 + 255) / 256) * 256;
    return ticket_bytes + (N + 15) / 16 * (long)split_k * 8 * 64 * 16;   // up to 8 row blocks (128 rows) of fp32 fragments
}
#else
extern GemmProfState padt_g_gemm_prof;
#endif

namespace PADT_NS {

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_SWIGLU = 3 };

struct GemmArgs {
    const x16_t* A; long lda;
    const x16_t* W; long ldw;
    const x16_t* bias;          // [N] or null
    void* C; long ldc;           // bf16 or f32
    const x16_t* R; long ldr;   // residual (EPI_RESID)
    int M, N, K;
    float* ws = nullptr;         // split-K (skinny kernel, gridDim.y > 1): partial accumulators [nb][split][frags][64][4]
    int* ticket = nullptr;       //          and one completion ticket per n-block (zero between launches)
    int split = 1;
    const float* rs = nullptr;   // optional per-row scale applied to the accumulator before bias (fused RMSNorm: rstd[m])
    RopeEpi rope = {nullptr, nullptr, 0, 0, 0};   // optional fused RoPE of the leading output columns (EPI_NONE only)
    int a_pack = 0;              // skinny kernel: A / (C and R) stored in the 16-row fragment-packed activation layout
    int c_pack = 0;              //   element (m, k) at (m/16)*16*ld + ((k/8)*16 + m%16)*8 + k%8   (see padt_hip.h)
    const float* cs = nullptr;   // optional per-output-column scale applied to the accumulator before bias (fp8 weights: dequantisation scale of weight row n)
    int r_f32 = 0;               // EPI_RESID: R is fp32 [M][ldr] (fp32 residual stream; with OUT_F32)
    long lo_off = 0;             // bf16 output: also store lo = bf16(x - hi) at C + lo_off (split-precision pair, padt_gemm_bf16_ex)
    unsigned long long* prof = nullptr;   // tile kernels: optional in-kernel launch timing slot (padt_gemm_profile)
    x16_t* C2 = nullptr;        // fp32 output: optional bf16 mirror of C (fp32 residual stream → the next projection's A operand), row-major
    long ldc2 = 0;               //   or, with c_pack, in the fragment-packed activation layout (decode steps)
    long a_lo_off = 0;           // skinny kernel, WQ = 2: A rows are (hi | lo) pairs, the lo half starts a_lo_off elements into the row (padt_gemm_split_rows)
};

// bf16 split pair of 4 fp32 values: hi = bf16(x), lo = bf16(x - hi)  (hi + lo carries 16 mantissa bits)
PADT_DEV void split4(const float* o, u32x2& hi, u32x2& lo) {
    hi = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
    float h[4];
    unpack4x(hi, h);
    lo = u32x2{pack2x(o[0] - h[0], o[1] - h[1]), pack2x(o[2] - h[2], o[3] - h[3])};
}

// offset of element (m, n) of a row-major or fragment-packed [rows][ld] activation matrix (n % 4 == 0 keeps 4 elements together)
PADT_DEV long act_index(int m, int n, long ld, int packed) {
    return packed ? (long)(m >> 4) * 16 * ld + ((long)(n >> 3) * 16 + (m & 15)) * 8 + (n & 7) : (long)m * ld + n;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];   // 256 B of zeros (K-tail source)

// ---------------------------------------------------------------------------------------------------------------------
// epilogue for one 16x16 fragment held "swapped": lane has row m, columns n..n+3 in v[0..3]
PADT_DEV void unpack4(u32x2 v, float* f) { unpack4x(v, f); }

template <int EPI, bool OUT_F32>
PADT_DEV void store_frag(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M || n >= p.N) return;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (p.rs) {
        const float sc = p.rs[m];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= sc;
    }
    if (p.cs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= (n + r < p.N) ? p.cs[n + r] : 1.f;
    }
    if (n + 3 < p.N) {                                   // full fragment: 8-byte bias / residual loads, one vector store
        const x16_t* bp = p.bias ? p.bias + n : reinterpret_cast<const x16_t*>(g_zero_page);
        const u32x2 braw = *reinterpret_cast<const u32x2*>(bp);     // both loads are issued before either is consumed
        u32x2 rraw = u32x2{0u, 0u};
        f32x4 rf = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_RESID) {
            if (p.r_f32) rf = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (long)m * p.ldr + n);
            else rraw = *reinterpret_cast<const u32x2*>(p.R + act_index(m, n, p.ldr, p.c_pack));
        }
        {
            float bv[4];
            unpack4(braw, bv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += bv[r];
        }
        if (EPI == EPI_NONE) rope_pairs(o, m, n, p.rope);
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
        }
        if (EPI == EPI_RESID) {
            float rv[4];
            unpack4(rraw, rv);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += p.r_f32 ? rf[r] : rv[r];
        }
        if (OUT_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
            if (p.C2) *reinterpret_cast<u32x2*>(p.C2 + act_index(m, n, p.ldc2, p.c_pack)) =
                u32x2{pack2x(o[0] * PADT_STREAM_SCALE, o[1] * PADT_STREAM_SCALE), pack2x(o[2] * PADT_STREAM_SCALE, o[3] * PADT_STREAM_SCALE)};
        }
        else if (p.lo_off) {
            u32x2 hi, lo;
            split4(o, hi, lo);
            x16_t* c = reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + n;
            *reinterpret_cast<u32x2*>(c) = hi;
            *reinterpret_cast<u32x2*>(c + p.lo_off) = lo;
        }
        else *reinterpret_cast<u32x2*>(reinterpret_cast<x16_t*>(p.C) + act_index(m, n, p.ldc, p.c_pack)) = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
        return;
    }
    for (int r = 0; r < 4 && n + r < p.N; ++r) {         // ragged N tail: scalar
        float x = o[r];
        if (p.bias) x += x2f(p.bias[n + r]);
        if (EPI == EPI_GELU) x = gelu_erf(x);
        if (EPI == EPI_RESID) x += p.r_f32 ? reinterpret_cast<const float*>(p.R)[(long)m * p.ldr + n + r] : x2f(p.R[(long)m * p.ldr + n + r]);
        if (OUT_F32) {
            reinterpret_cast<float*>(p.C)[(long)m * p.ldc + n + r] = x;
            if (p.C2) p.C2[act_index(m, n + r, p.ldc2, p.c_pack)] = f2x(x * PADT_STREAM_SCALE);
        } else {
            const x16_t h = f2x(x);
            reinterpret_cast<x16_t*>(p.C)[(long)m * p.ldc + n + r] = h;
            if (p.lo_off) reinterpret_cast<x16_t*>(p.C)[(long)m * p.ldc + n + r + p.lo_off] = f2x(x - x2f(h));
        }
    }
}

// SwiGLU pair: g/u fragments of weight rows [32q,32q+16) / [32q+16,32q+32) → output columns 16q + ...
PADT_DEV void store_swiglu(const GemmArgs& p, int m, int n_gate, f32x4 g, f32x4 u) {
    // n_gate = interleaved row index of the first gate element held by this lane (multiple of 4, inside a gate block)
    if (m >= p.M || n_gate >= p.N) return;
    const int blk = n_gate >> 5, in = n_gate & 15;
    const int no = blk * 16 + in;                       // output column
    float gb[4], ub[4];
    const x16_t* bp = p.bias ? p.bias + n_gate : reinterpret_cast<const x16_t*>(g_zero_page);
    unpack4(*reinterpret_cast<const u32x2*>(bp), gb);
    unpack4(*reinterpret_cast<const u32x2*>(bp + 16), ub);
    float o[4];
    const float sc = p.rs ? p.rs[m] : 1.0f;
    f32x4 gs = f32x4{1.f, 1.f, 1.f, 1.f}, us = gs;
    if (p.cs) { gs = *reinterpret_cast<const f32x4*>(p.cs + n_gate); us = *reinterpret_cast<const f32x4*>(p.cs + n_gate + 16); }
    if (p.lo_off) {
        // split SwiGLU (precision="reference", round 6): exact expf + IEEE division (what padt_swiglu_split computes), (hi, lo) pair with lo at lo_off
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gv = g[r] * sc * gs[r] + gb[r];
            o[r] = gv / (1.0f + expf(-gv)) * (u[r] * sc * us[r] + ub[r]);
        }
        u32x2 hi, lo;
        split4(o, hi, lo);
        x16_t* c = reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + no;
        *reinterpret_cast<u32x2*>(c) = hi;
        *reinterpret_cast<u32x2*>(c + p.lo_off) = lo;
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = silu(g[r] * sc * gs[r] + gb[r]) * (u[r] * sc * us[r] + ub[r]);
    x16_t* c = reinterpret_cast<x16_t*>(p.C) + act_index(m, no, p.ldc, p.c_pack);
    *reinterpret_cast<u32x2*>(c) = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile kernel
constexpr int BM = 128, BN = 128;
template <int BK> struct TileCfg {
    static constexpr int ROW_BYTES = BK * 2;                 // bytes of one tile row in LDS (128 or 64)
    static constexpr int CPR = BK / 8;                       // 16-byte chunks per row (8 or 4)
    static constexpr int TILE_BYTES = BM * ROW_BYTES;        // one operand tile
    static constexpr int LDS = 4 * TILE_BYTES;               // 2 buffers x (A, W)
    static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;    // rows covered by one 1-KiB wave DMA (8 or 16)
    static constexpr int DMA_PER_WAVE = BM / ROWS_PER_DMA / 4;
    // bank-conflict-free XOR swizzle of the 16-byte chunk index for ds_read_b128 (MI355X lane groups):
    //   128-byte rows: chunk ^ (row & 7);  64-byte rows: chunk ^ ((row >> 1) & 3)   (searched exhaustively, 0 conflicts)
    PADT_DEV static int swz(int row, int chunk) { return BK == 64 ? (chunk ^ (row & 7)) : (chunk ^ ((row >> 1) & 3)); }
};

template <int BK>
PADT_DEV void stage_tile(const x16_t* __restrict__ base, long ld, int row0, int nrows, int k0, int K,
                         char* lds_tile, int wave, int lane) {
    using T = TileCfg<BK>;
    // LDS slot (r, s) holds global chunk swz(r, s): the DMA image is lane-linear, so the swizzle goes on the SOURCE address
#pragma unroll
    for (int i = 0; i < T::DMA_PER_WAVE; ++i) {
        const int c = wave * T::DMA_PER_WAVE + i;
        const int r = c * T::ROWS_PER_DMA + lane / T::CPR;
        const int sl = lane % T::CPR;
        const int j = T::swz(r, sl);
        int row = row0 + r;
        row = row < nrows ? row : nrows - 1;
        const int k = k0 + j * 8;
        unsigned long long src = reinterpret_cast<unsigned long long>(base + (long)row * ld + k);
        unsigned long long zp = reinterpret_cast<unsigned long long>(g_zero_page);
        asm volatile("" : "+v"(zp));                       // keep both candidates in VGPRs: one straight-line DMA
        src = (k < K) ? src : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + c * 1024), 16, 0, 0);
    }
}

// Measured dead ends on this structure (profiles/r01_gemm_tile_experiments.md): BK = 32 with 3 blocks/CU (-15 %),
// precomputed per-lane DMA pointers (+50 VGPRs, -9 %), DMA pieces spread between the MFMA groups (-10 %).
template <int EPI, bool OUT_F32, int BK>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmArgs p) {
    using T = TileCfg<BK>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ProfScope prof_scope(p.prof, tid);
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int id = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = id / ntn, tn = id % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = (p.K + BK - 1) / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS: [buf0: A | W][buf1: A | W]
    stage_tile<BK>(p.A, p.lda, m0, p.M, 0, p.K, smem, wave, lane);
    stage_tile<BK>(p.W, p.ldw, n0, p.N, 0, p.K, smem + T::TILE_BYTES, wave, lane);

    const int frow = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile t landed everywhere; everyone is done reading buf[(t+1)&1]
        const int cur = t & 1;
        if (t + 1 < nk) {
            char* nxt = smem + (cur ^ 1) * 2 * T::TILE_BYTES;
            stage_tile<BK>(p.A, p.lda, m0, p.M, (t + 1) * BK, p.K, nxt, wave, lane);
            stage_tile<BK>(p.W, p.ldw, n0, p.N, (t + 1) * BK, p.K, nxt + T::TILE_BYTES, wave, lane);
        }
        const char* a_t = smem + cur * 2 * T::TILE_BYTES;
        const char* w_t = a_t + T::TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int j = kk * 4 + fq;
            x16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + frow;
                af[i] = ld_frag(a_t + ra * T::ROW_BYTES + (T::swz(ra, j) << 4));
                const int rw = wn * 64 + i * 16 + frow;
                wf[i] = ld_frag(w_t + rw * T::ROW_BYTES + (T::swz(rw, j) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(wf[ni], af[mi], acc[mi][ni]);
        }
    }

    // epilogue: acc[mi][ni][r] = C[m0 + wm*64 + mi*16 + (lane&15)][n0 + wn*64 + ni*16 + (lane>>4)*4 + r]
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
    if (interior && EPI != EPI_SWIGLU && !p.r_f32 && !p.lo_off && !p.cs && !p.C2) {
        // block-uniform fast path: no per-fragment bounds checks; every bias / residual load is issued up front
        const int mb = m0 + wm * 64 + frow, nb = n0 + wn * 64 + fq * 4;
        u32x2 braw[4], rraw[4][4];
        const x16_t* bp = p.bias ? p.bias + nb : reinterpret_cast<const x16_t*>(g_zero_page);
        const int bstep = p.bias ? 16 : 0;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) braw[ni] = *reinterpret_cast<const u32x2*>(bp + ni * bstep);
        if (EPI == EPI_RESID) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    rraw[mi][ni] = *reinterpret_cast<const u32x2*>(p.R + (long)(mb + mi * 16) * p.ldr + nb + ni * 16);
        }
        float rsc[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) rsc[mi] = p.rs ? p.rs[mb + mi * 16] : 1.0f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                float bv[4], o[4];
                unpack4(braw[ni], bv);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[mi][ni][r] * rsc[mi] + bv[r];
                if (EPI == EPI_NONE) rope_pairs(o, mb + mi * 16, nb + ni * 16, p.rope);
                if (EPI == EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
                }
                if (EPI == EPI_RESID) {
                    float rv[4];
                    unpack4(rraw[mi][ni], rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += rv[r];
                }
                const long off = (long)(mb + mi * 16) * p.ldc + nb + ni * 16;
                if (OUT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = f32x4{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<u32x2*>(reinterpret_cast<x16_t*>(p.C) + off) = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
            }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + frow;
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int n = n0 + wn * 64 + ni * 16 + fq * 4;
                store_swiglu(p, m, n, acc[mi][ni], acc[mi][ni + 1]);
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wn * 64 + ni * 16 + fq * 4;
                store_frag<EPI, OUT_F32>(p, m, n, acc[mi][ni]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Skinny kernel: M <= 16*MT.  Block = NW waves, owns NT*16 weight rows; wave w handles K-steps w, w+NW, ...
// Weights are streamed with non-temporal 16-byte loads (read exactly once per launch); U K-steps are in flight per wave.
// NORM: the RMSNorm that precedes the projection in the reference (HF:727,744 input/post-attention layernorm) is fused:
//   y = (x * rsqrt(mean(x^2)+eps) * g) @ W^T  ==  rstd[m] * (x @ (W·diag(g))^T)[m]  — the norm weight g is folded into the
//   weight matrix once at load time (weights.py), the per-row sum of squares is accumulated from the x fragments the
//   MFMA consumes anyway, and rstd scales the fp32 accumulator.
// WQ = 2 (round 6, padt_gemm_split_rows): split-precision A rows [hi(K) | lo(K)] against the PLAIN weight matrix — every weight fragment is
//   loaded ONCE and multiplied with the row block's hi and lo fragments (acc += W·hi + W·lo per K-step).  The decode steps of
//   precision="reference" streamed the doubled image [W | W] through the K' = 2K loop: twice the HBM bytes of a launch that is bound by them.
// WQ = 1: fp8 weights (OCP e4m3, per-output-row scale in p.cs) in the fp8 fragment-packed image [N/16][Kp/64][64 lanes][16 B]: a lane's
//   16 bytes hold its 8 elements of K-step 2t and its 8 elements of K-step 2t + 1 — one 1-KiB wave load feeds two MFMA K-steps, the
//   weight stream is half the bf16 bytes; bytes are converted to bf16 fragments in registers (exact), accumulation stays fp32.
template <int MT, int NT, int NW, int EPI, bool OUT_F32, bool NORM, bool PACKED, int WQ = 0>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p, float norm_eps) {
    // LDS (dynamic: 64 rows x 2 weight-row blocks x 8 waves do not fit the 64 KiB static limit): every wave's partial fragments
    // red[NW][NT * MT][64][4] fp32, then the per-wave row sums of squares ssq[NW][MT][16]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef float RedT[NT * MT][64][4];
    typedef float SsqT[MT][16];
    RedT* red = reinterpret_cast<RedT*>(smem_raw);
    SsqT* ssq = reinterpret_cast<SsqT*>(smem_raw + sizeof(RedT) * NW);
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int nks = (p.K + 31) / 32;
    constexpr int U = (MT <= 2) ? 8 : ((MT * NT >= 8 || MT > 4) ? 2 : 4);   // K-steps in flight per wave (VGPR budget)

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) ss[j] = 0.f;

    const x16_t* wrow[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int n = n0 + i * 16 + frow;
        n = n < p.N ? n : p.N - 1;
        wrow[i] = p.W + (long)n * p.ldw;
    }
    // x fragment of K-step ks = 16 bytes at xrow[j] + ks * xstep.  Row-major activations: 16 rows x 64 B per wave
    // instruction (address unit at a quarter rate — the limiter once 32 rows are decoded together); packed activations
    // (a_pack): the same fragment is 1 KiB contiguous in lane order.
    const x16_t* xrow[MT];
    bool xok[MT];
    const int xstep = p.a_pack ? 512 : 32;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = j * 16 + frow;
        xok[j] = p.a_pack ? (j * 16 < p.M) : (m < p.M);          // packed: a row block past the last valid row is not in the buffer (M = 96 → 6 of MT = 8)
        xrow[j] = p.a_pack ? p.A + (long)j * 16 * p.lda + lane * 8 : p.A + (long)(xok[j] ? m : 0) * p.lda + fq * 8;
    }

    // wave w owns K-step PAIRS w, w+NW, ... (KG = 2 consecutive K-steps = 2 KiB of a packed weight row block) and keeps U K-steps =
    // U/2 of its pairs in flight, consumed in K order.  The pair → wave map does NOT depend on the row count (MT) — only U does — so a
    // row's fp32 summation order, and with it every output bit, is the same whether 8, 16, 32 or 64 rows share the launch (in-flight
    // batching of decode groups must not change a sample's tokens; tools/check_rows_invariance.py).
    // Round 3 measured three restructurings of this loop at 64 rows and kept none (profiles/r03_decode_experiments.md): a two-stage register
    // ping-pong (gate/up 26.0 → 27.6 us: 204 VGPRs, 2 waves per SIMD), and two "shared activation" kernels whose waves split the weight rows
    // instead of K so that a block reads the activations once (through L1: 24.5 us; through an LDS double buffer: 26.2 us; 12 waves: spills).
    // split-K: gridDim.y blocks share an n-block, block y takes pairs y*NW + wave, stepping by NW*gridDim.y
    constexpr int KG = 2, GPI = U / KG;
    static_assert(U % KG == 0, "U is a whole number of K-step pairs");
    const int g_stride = NW * gridDim.y;
    for (int gi = blockIdx.y * NW + wave; gi * KG < nks; gi += g_stride * GPI) {
        x16x8 wf[U][NT], xf[U][MT];
        x16x8 xl[WQ == 2 ? U : 1][WQ == 2 ? MT : 1];             // WQ = 2: the lo halves of the same rows
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = (gi + (u / KG) * g_stride) * KG + (u % KG);
            const int k = ks * 32 + fq * 8;
            const bool kok = (ks < nks) && (k < p.K);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (WQ == 1) {
                    if ((u & 1) == 0) {                           // U is even and groups start at even K-steps: one load = K-steps (ks, ks + 1)
                        const unsigned char* wq = reinterpret_cast<const unsigned char*>(p.W) +
                                                  (((long)(n0 / 16 + i) * (p.ldw / 64) + (ks >> 1)) * 64 + lane) * 16;
                        u32x4 q = u32x4{0u, 0u, 0u, 0u};
                        if (ks < nks) q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wq));
                        wf[u][i] = fp8x8_to_x16x8(q[0], q[1]);
                        wf[u + (U > 1 ? 1 : 0)][i] = fp8x8_to_x16x8(q[2], q[3]);
                    }
                    continue;
                }
                // PACKED: tile (n16, k32) of the fragment-packed image is 1 KiB in lane order → one contiguous wave load
                const x16_t* wp = PACKED ? p.W + ((long)(n0 / 16 + i) * (p.ldw / 32) + ks) * 512 + lane * 8 : wrow[i] + k;
                wf[u][i] = kok ? __builtin_nontemporal_load(reinterpret_cast<const x16x8*>(wp)) : zero_frag();
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[u][j] = (kok && xok[j]) ? ld_frag(xrow[j] + (long)ks * xstep) : zero_frag();
            if constexpr (WQ == 2) {
#pragma unroll
                for (int j = 0; j < MT; ++j) xl[u][j] = (kok && xok[j]) ? ld_frag(xrow[j] + p.a_lo_off + (long)ks * xstep) : zero_frag();
            }
            if (NORM) {
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    float xv[8];
                    unpack8(__builtin_bit_cast(u32x4, xf[u][j]), xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss[j] += xv[e] * xv[e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    acc[i][j] = mfma16(wf[u][i], xf[u][j], acc[i][j]);
                    if constexpr (WQ == 2) acc[i][j] = mfma16(wf[u][i], xl[u][j], acc[i][j]);
                }
    }

    if (NORM) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float t = ss[j];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (fq == 0) ssq[wave][j][frow] = t;
        }
    }
    // Cross-wave reduction + epilogue, spread over the waves: wave w sums the NW partials of row blocks w, w + NW, ... (< MT) and
    // runs their epilogues — with 64+ decode rows the tail is as long as the K loop, one wave doing all of it left the other
    // waves of the block idle.  Partials are summed in wave order whatever wave does the summing.
    constexpr int JW = (MT + NW - 1) / NW;                       // row blocks per finishing wave
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * MT + j][lane][0]) = acc[i][j];
    __syncthreads();
    f32x4 sum[JW][NT];
#pragma unroll
    for (int q = 0; q < JW; ++q) {
        const int jw = wave + q * NW;
        if (jw < MT) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                sum[q][i] = *reinterpret_cast<f32x4*>(&red[0][i * MT + jw][lane][0]);
#pragma unroll
                for (int w = 1; w < NW; ++w) sum[q][i] += *reinterpret_cast<f32x4*>(&red[w][i * MT + jw][lane][0]);
            }
        }
    }
    if (!NORM && gridDim.y > 1) {
        // split-K: the last of the n-block's split blocks to arrive sums the partials and runs the epilogue
        const int S = gridDim.y;
#pragma unroll
        for (int q = 0; q < JW; ++q) {
            const int jw = wave + q * NW;
            if (jw < MT) {
                float* mine = p.ws + (((long)(blockIdx.x * S + blockIdx.y) * (NT * MT)) * 64 + lane) * 4;
#pragma unroll
                for (int i = 0; i < NT; ++i) st_agent4(mine + (i * MT + jw) * 256, sum[q][i]);    // one 16-byte write-through store per fragment
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's partial stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&p.ticket[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == S - 1);
            if (last) __hip_atomic_store(&p.ticket[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag = last;
        }
        __syncthreads();
        if (!flag) return;
        // partials are added in split order 0, 1, ..., S - 1 whichever block finishes (its own comes from registers): with S > 2 an
        // "own first" order would make the fp32 sum depend on the arrival order
#pragma unroll
        for (int q = 0; q < JW; ++q) {
            const int jw = wave + q * NW;
            if (jw < MT) {
                f32x4 tot[NT];
                for (int y = 0; y < S; ++y) {
                    const float* part = p.ws + (((long)(blockIdx.x * S + y) * (NT * MT)) * 64 + lane) * 4;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        f32x4 v = sum[q][i];
                        if (y != (int)blockIdx.y) v = ld_agent4(part + (i * MT + jw) * 256);
                        tot[i] = (y == 0) ? v : tot[i] + v;
                    }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) sum[q][i] = tot[i];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < JW; ++q) {
        const int jw = wave + q * NW;
        if (jw >= MT) continue;
        const int m = jw * 16 + frow;
        if (NORM) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += ssq[w][jw][frow];
            const float rstd = rsqrtf(t / (float)p.K + norm_eps);
#pragma unroll
            for (int i = 0; i < NT; ++i) sum[q][i] *= rstd;
        }
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int i = 0; i < NT; i += 2) store_swiglu(p, m, n0 + i * 16 + fq * 4, sum[q][i], sum[q][i + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) store_frag<EPI, OUT_F32>(p, m, n0 + i * 16 + fq * 4, sum[q][i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// gemm256.hip: phase-pipelined 256x256 kernel for large-N shapes; returns 0 if it took the launch
extern "C" int PADT_TWIN(padt_gemm256_try)(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                                const float* row_scale, const RopeEpi* rope, long* rows_done, int resid_f32, long lo_off,
                                void* C2, long ldc2, unsigned long long* prof);

static unsigned long long* next_prof_slot() {
    unsigned long long* slots = padt_g_gemm_prof.slots.load();
    if (!slots) return nullptr;
    const long i = padt_g_gemm_prof.n.fetch_add(1);
    if (i >= padt_g_gemm_prof.cap.load()) { padt_g_gemm_prof.n.store(padt_g_gemm_prof.cap.load()); return nullptr; }
    return slots + 2 * i;
}

template <int EPI, bool F32, int BK>
static void launch_tile_bk(const GemmArgs& a, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<EPI, F32, BK>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<BK>::LDS); });
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_tile_kernel<EPI, F32, BK>), dim3(ntm * ntn), dim3(256), TileCfg<BK>::LDS, s, a);
}

template <int EPI, bool F32>
static void launch_tile(const GemmArgs& a, hipStream_t s) { launch_tile_bk<EPI, F32, 64>(a, s); }     // BK = 32 measured −15 % (round 1) and is gone

template <int MT, int NT, int NW, int EPI, bool F32, bool NORM, bool PACKED, int WQ>
static void launch_skinny_nt(const GemmArgs& a, float eps, hipStream_t s) {
    const int nb = (a.N + 16 * NT - 1) / (16 * NT);
    const int split = (a.ws && !NORM) ? a.split : 1;
    constexpr int lds = NW * NT * MT * 64 * 16 + NW * MT * 16 * 4;      // red + ssq (gemm_skinny_kernel)
    if constexpr (lds > 64 * 1024) {
        static PerDeviceOnce once;
        once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MT, NT, NW, EPI, F32, NORM, PACKED, WQ>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, NW, EPI, F32, NORM, PACKED, WQ>), dim3(nb, split), dim3(NW * 64), lds, s, a, eps);
}

// Wide row blocks at 64 decode rows (round 3).  A wave loads MT = 4 activation fragments (L2 hits) per weight fragment there, and that
// L2 → register traffic — 176 MB next to 90 MB of gate/up weights — not the weight stream, sets the launch time
// (profiles/r03_decode_experiments.md §3).  Twice the weight fragments per wave and K-step (NT = 4: two gate/up pairs) halve the activation
// loads per weight byte IN REGISTERS, the only sharing that removes load instructions: gate/up 26.4 → 24.0 us.  NT changes neither the
// K-step → wave map nor the cross-wave summation order, so every output bit is the narrow launch's (NW comes from the narrow block count).
// Taken only where the halved grid still covers the chip (>= 256 blocks); measured and NOT taken: 128 rows (NT x MT = 32 accumulator
// fragments: one wave per SIMD, 42.5 → 47.7 us), qkv / o (80 / 64 blocks left: 8.9 → 10.5, 7.6 → 10.9 us), 4 K-step pairs in flight (24.9 us);
// down with twice the split: 20.3 → 19.7 us, not worth a different split at 8 rows.  PADT_SKINNY_WIDE=0 is the A/B switch.
template <int MT, int NW, int EPI, bool F32, bool NORM, bool PACKED = false, int WQ = 0>
static void launch_skinny_nw(const GemmArgs& a, float eps, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    if constexpr (PACKED && MT == 4 && EPI == EPI_SWIGLU && NW == 4) {
        static const int wide = getenv("PADT_SKINNY_WIDE") ? atoi(getenv("PADT_SKINNY_WIDE")) : 1;
        if (wide && a.N % (32 * NT) == 0 && a.N / (32 * NT) >= 256) {
            launch_skinny_nt<MT, 2 * NT, NW, EPI, F32, NORM, PACKED, WQ>(a, eps, s);
            return;
        }
    }
    launch_skinny_nt<MT, NT, NW, EPI, F32, NORM, PACKED, WQ>(a, eps, s);
}

// waves per block: enough waves chip-wide (>= ~2048) to keep HBM busy even when N/16 < #CUs
template <int MT, int EPI, bool F32, bool NORM, bool PACKED = false, int WQ = 0>
static void launch_skinny(const GemmArgs& a, float eps, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int nb = (a.N + 16 * NT - 1) / (16 * NT);
    const int ksteps = (a.K + 31) / 32 / (a.ws ? a.split : 1);   // per block; each wave keeps U = 8 K-steps in flight
    // (the wave count must not depend on MT either: the cross-wave sum runs in wave order.  A 16-wave variant for <= 16 rows was 5 %
    //  faster on the 8-row down-projection and is gone for that reason.)
    if (nb <= 512 && ksteps >= 64) launch_skinny_nw<MT, 8, EPI, F32, NORM, PACKED, WQ>(a, eps, s);
    else launch_skinny_nw<MT, 4, EPI, F32, NORM, PACKED, WQ>(a, eps, s);
}

template <int EPI, bool F32>
static void dispatch_m(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 16) launch_skinny<1, EPI, F32, false>(a, 0.f, s);
    else if (a.M <= 32) launch_skinny<2, EPI, F32, false>(a, 0.f, s);
    else if (a.M <= 64) launch_skinny<4, EPI, F32, false>(a, 0.f, s);
    else launch_tile<EPI, F32>(a, s);
}

template <int EPI>
static void dispatch_norm(const GemmArgs& a, float eps, hipStream_t s) {
    if (a.M <= 16) launch_skinny<1, EPI, false, true>(a, eps, s);
    else if (a.M <= 32) launch_skinny<2, EPI, false, true>(a, eps, s);
    else launch_skinny<4, EPI, false, true>(a, eps, s);
}

static int gemm_bf16_impl(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                          long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                          const void* row_scale, const RopeEpi& rope, int resid_f32 = 0, long lo_off = 0, void* C2 = nullptr, long ldc2 = 0) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldw & 7) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) {
        padt_set_error("padt_gemm_bf16: K, lda, ldw must be multiples of 8 and A, W 16-byte aligned");
        return -1;
    }
    const long out_n = (epilogue == EPI_SWIGLU) ? N / 2 : N;
    if ((ldc & 3) || ((uintptr_t)C & 15) || (epilogue == EPI_SWIGLU && ((N & 31) || out_f32)) ||
        (epilogue == EPI_RESID && (R == nullptr || (ldr & 3) || ((uintptr_t)R & 7))) || ((uintptr_t)bias & 7) || ldc < out_n) {
        padt_set_error("padt_gemm_bf16: bad C/ldc/epilogue arguments (ldc % 4, C 16-byte aligned, SwiGLU needs N % 32 == 0 and bf16 out)");
        return -1;
    }
    if (epilogue < 0 || epilogue > 3) { padt_set_error("padt_gemm_bf16: unknown epilogue"); return -1; }
    long done = 0;
    const float* rs = (const float*)row_scale;
    RopeEpi rp = rope;
    if (resid_f32 && (epilogue != EPI_RESID || !out_f32 || ((uintptr_t)R & 15))) {
        padt_set_error("padt_gemm_bf16_ex: an fp32 residual needs epilogue 2, fp32 output and a 16-byte aligned R");
        return -1;
    }
    if (lo_off && (out_f32 || (lo_off & 3) || lo_off < out_n || ldc < lo_off + out_n)) {
        padt_set_error("padt_gemm_bf16_ex: a split (hi|lo) output needs bf16 output, lo_off % 4 == 0, lo_off >= the output width (N; N / 2 with SwiGLU) and ldc >= lo_off + that width");
        return -1;
    }
    const int tag256 = (epilogue == EPI_SWIGLU && lo_off) ? 1 : out_f32;      // gemm_tile256_kernel<EPI_SWIGLU, true> = the split SwiGLU (16-bit pairs)
    if (C2 && (!out_f32 || (ldc2 & 7) || ((uintptr_t)C2 & 15) || ldc2 < N)) {
        padt_set_error("padt_gemm_resid32: the bf16 mirror needs fp32 output, ldxb % 8 == 0, ldxb >= N and a 16-byte aligned pointer");
        return -1;
    }
    unsigned long long* prof = (M > 64) ? next_prof_slot() : nullptr;
    if (M > 64 && PADT_TWIN(padt_gemm256_try)(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, tag256, rs, &rp, &done, resid_f32, lo_off, C2, ldc2, prof) == 0) {
        if (done >= M) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
            return 0;
        }
        // a peeled ragged tail (<= 64 rows): the rest of this function streams it through the skinny kernel
        A = (const x16_t*)A + done * lda;
        C = out_f32 ? (void*)((float*)C + done * ldc) : (void*)((x16_t*)C + done * ldc);
        if (R) R = resid_f32 ? (const void*)((const float*)R + done * ldr) : (const void*)((const x16_t*)R + done * ldr);
        if (rs) rs += done;
        if (C2) C2 = (x16_t*)C2 + done * ldc2;
        if (rp.cos) { rp.cos += done * rp.ld; rp.sin += done * rp.ld; }
        M -= done;
    }
    GemmArgs a{(const x16_t*)A, lda, (const x16_t*)W, ldw, (const x16_t*)bias, C, ldc, (const x16_t*)R, ldr,
               (int)M, (int)N, (int)K};
    a.rs = rs;
    a.rope = rp;
    a.r_f32 = resid_f32;
    a.lo_off = lo_off;
    a.C2 = (x16_t*)C2;
    a.ldc2 = ldc2;
    a.prof = (M > 64) ? prof : nullptr;                           // a peeled <= 64-row tail runs on the skinny kernel: not part of the tile-GEMM time
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue * 2 + (out_f32 ? 1 : 0)) {
        case 0: dispatch_m<EPI_NONE, false>(a, s); break;
        case 1: dispatch_m<EPI_NONE, true>(a, s); break;
        case 2: dispatch_m<EPI_GELU, false>(a, s); break;
        case 3: dispatch_m<EPI_GELU, true>(a, s); break;
        case 4: dispatch_m<EPI_RESID, false>(a, s); break;
        case 5: dispatch_m<EPI_RESID, true>(a, s); break;
        case 6: dispatch_m<EPI_SWIGLU, false>(a, s); break;
        default: padt_set_error("padt_gemm_bf16: unsupported epilogue/out combination"); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

extern "C" int PADT_SYM(padt_gemm_, )(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                              long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                              const void* row_scale) {
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, out_f32, row_scale,
                          RopeEpi{nullptr, nullptr, 0, 0, 0});
}

// Extended epilogues for the split-precision PaDT decoder (decoder_hp.hip): resid_f32 — R (and C: out_f32 must be set) is the
// fp32 residual stream; lo_off != 0 — bf16 output stored as a (hi, lo) pair, hi at C[m][n], lo = bf16(x - hi) at C[m][lo_off + n],
// i.e. directly the [hi | lo] A operand (K' = 2N) of the next GEMM whose weight image is [W | W].
#if !PADT_OP16_F16
extern "C" int padt_gemm_bf16_ex(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                 long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                                 const void* row_scale, int resid_f32, long lo_off) {
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, out_f32, row_scale,
                          RopeEpi{nullptr, nullptr, 0, 0, 0}, resid_f32, lo_off);
}
#endif

#if !PADT_OP16_F16
// Split-precision projection for FEW rows (the decode steps of precision="reference"; padt_decoder / HF Linear at one token per row):
// C_f32[m][n] = epi(sum_k (hi[m][k] + lo[m][k]) * W[n][k] + bias[n] (+ R_f32[m][n])), A rows = [hi(K) | lo(K)] bf16 pairs (lo at a_lo_off),
// W [N][ldw] bf16 with the first K columns used — pass the doubled image [W | W] of padt_gemm_bf16_ex with ldw = 2K and it is read ONCE.
// M <= 64; epilogue 0 or 2 (fp32 residual, in place allowed) → fp32 C; epilogue 3 (SwiGLU over [gate16 | up16]-interleaved weight rows, N % 32 == 0)
// → C = bf16 split rows: silu(gate) * up as (hi, lo) pairs, hi at C[m][n], lo at C[m][c_lo_off + n], n < N / 2.
// layout (round 6): bit 0 — A is in the 16-row fragment-packed activation layout (padt_pack_rows of the split rows: lda = their row length 2K, a_lo_off =
// 16 K: the lo fragment of K-step ks is the fragment of K-step ks + K / 32); bit 1 — W is the fragment-packed image of padt_gemm_packed_bf16
// ([N/16][ldw/32][64 lanes][8], ldw = K padded to 32).  Row-major rows cost 16 x 64-byte pieces per wave load on both operands — the address unit, not
// HBM, bounded these launches (gate/up 70 us at 64 rows against 24 us for the default path's packed launch).
template <bool PACKED>
static void split_rows_launch(const GemmArgs& a, long M, int epilogue, hipStream_t s) {
    if (epilogue == EPI_SWIGLU) {
        if (M <= 16) launch_skinny<1, EPI_SWIGLU, false, false, PACKED, 2>(a, 0.f, s);
        else if (M <= 32) launch_skinny<2, EPI_SWIGLU, false, false, PACKED, 2>(a, 0.f, s);
        else launch_skinny<4, EPI_SWIGLU, false, false, PACKED, 2>(a, 0.f, s);
    } else if (epilogue == EPI_RESID) {
        if (M <= 16) launch_skinny<1, EPI_RESID, true, false, PACKED, 2>(a, 0.f, s);
        else if (M <= 32) launch_skinny<2, EPI_RESID, true, false, PACKED, 2>(a, 0.f, s);
        else launch_skinny<4, EPI_RESID, true, false, PACKED, 2>(a, 0.f, s);
    } else {
        if (M <= 16) launch_skinny<1, EPI_NONE, true, false, PACKED, 2>(a, 0.f, s);
        else if (M <= 32) launch_skinny<2, EPI_NONE, true, false, PACKED, 2>(a, 0.f, s);
        else launch_skinny<4, EPI_NONE, true, false, PACKED, 2>(a, 0.f, s);
    }
}

extern "C" int padt_gemm_split_rows(void* stream, const void* A_split, long lda, long a_lo_off, const void* W, long ldw, const void* bias, void* C,
                                    long ldc, long c_lo_off, const void* R_f32, long ldr, long M, long N, long K, int epilogue, int layout) {
    if (M <= 0 || N <= 0) return 0;
    const bool glu = epilogue == EPI_SWIGLU;
    const bool a_packed = layout & 1, w_packed = layout & 2;
    if (M > 64 || K <= 0 || (K & 7) || (lda & 7) || (ldw & 7) || (a_lo_off & 7) || a_lo_off < K || ((uintptr_t)A_split & 15) || ((uintptr_t)W & 15) ||
        (ldc & 3) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 7) || (epilogue != EPI_NONE && epilogue != EPI_RESID && !glu) ||
        (epilogue == EPI_RESID && (R_f32 == nullptr || (ldr & 3) || ((uintptr_t)R_f32 & 15))) ||
        (glu && ((N & 31) || (c_lo_off & 3) || c_lo_off < N / 2 || ldc < c_lo_off + N / 2)) || (!glu && c_lo_off != 0) ||
        (w_packed && ((ldw & 31) || ldw < K || (N & 15))) || (a_packed && ((K & 31) || a_lo_off != 16 * K)) || (layout & ~3)) {
        padt_set_error("padt_gemm_split_rows: M <= 64, K / lda / ldw / a_lo_off multiples of 8, a_lo_off >= K, 16-byte aligned A / W / C / R, epilogue 0, 2 "
                       "(fp32 C, c_lo_off 0) or 3 (N % 32 == 0, bf16 pairs: c_lo_off % 4 == 0, c_lo_off >= N / 2, ldc >= c_lo_off + N / 2); packed W: ldw % 32 == 0, "
                       "N % 16 == 0; packed A: K % 32 == 0 and a_lo_off == 16 K");
        return -1;
    }
    GemmArgs a{(const x16_t*)A_split, lda, (const x16_t*)W, ldw, (const x16_t*)bias, C, ldc, (const x16_t*)R_f32, ldr, (int)M, (int)N, (int)K};
    a.a_lo_off = a_lo_off;
    a.lo_off = c_lo_off;
    a.r_f32 = epilogue == EPI_RESID ? 1 : 0;
    a.a_pack = a_packed ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (w_packed) split_rows_launch<true>(a, M, epilogue, s);
    else split_rows_launch<false>(a, M, epilogue, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
#endif

// fp32 residual stream: X32[m][n] += (A · W^T)[m][n] + bias[n] in place, and Xb = bf16(X32) (row-major mirror, may be null) — the residual
// adds of the ViT block (HF:318-320) and the LLM layer (HF:741,757) with the stream kept in fp32 between kernels and the next projection's
// bf16 A operand produced by the same epilogue.
extern "C" int PADT_TWIN(padt_gemm_resid32)(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* X32, long ldx,
                                 void* Xb, long ldxb, long M, long N, long K) {
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, X32, ldx, X32, ldx, M, N, K, EPI_RESID, 1, nullptr,
                          RopeEpi{nullptr, nullptr, 0, 0, 0}, 1, 0, Xb, ldxb);
}

// C = rope(row_scale[m] * (A · W^T) + bias): the rotate-half RoPE of the leading `rope_cols` output columns fused into the
// epilogue (ViT qkv projection: q and k columns, pair-interleaved per head by a load-time permutation of W's rows).
extern "C" int PADT_SYM(padt_gemm_rope_, )(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                   long ldc, long M, long N, long K, const void* row_scale, const void* rope_cos,
                                   const void* rope_sin, long ld_cs, long rope_cols, int head_dim) {
    if (rope_cos == nullptr || rope_sin == nullptr || head_dim <= 0 || (head_dim & 3) || (rope_cols & 3) || rope_cols > N ||
        rope_cols % head_dim || (N & 3) || (ld_cs & 1) || ((uintptr_t)rope_cos & 7) || ((uintptr_t)rope_sin & 7)) {
        padt_set_error("padt_gemm_rope_bf16: need cos/sin tables, head_dim % 4 == 0, rope_cols a multiple of head_dim and <= N, N % 4 == 0");
        return -1;
    }
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, C, ldc, nullptr, 0, M, N, K, EPI_NONE, 0, row_scale,
                          RopeEpi{(const float*)rope_cos, (const float*)rope_sin, ld_cs, (int)rope_cols, head_dim});
}

// Fused RMSNorm + projection for decode-sized batches (M <= 64):  C = epi(rstd(A)[m] * (A · W^T)[m] + bias), where
// rstd = rsqrt(mean(A[m]^2) + eps) and W already carries the norm weight (W·diag(g), folded at load time).
// epilogue 0 (none) or 3 (SwiGLU).  Replaces {input_layernorm → q/k/v_proj} and {post_attention_layernorm → gate/up_proj}
// (HF:727-757) for the single-token decode step.
extern "C" int PADT_SYM(padt_gemm_rmsnorm_, )(void* stream, const void* A, long lda, float eps, const void* W, long ldw,
                                      const void* bias, void* C, long ldc, long M, long N, long K, int epilogue) {
    if (M <= 0 || N <= 0) return 0;
    if (M > 64 || K <= 0 || (K & 7) || (lda & 7) || (ldw & 7) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) {
        padt_set_error("padt_gemm_rmsnorm_bf16: M <= 64, K/lda/ldw multiples of 8, 16-byte aligned A/W required");
        return -1;
    }
    if ((ldc & 3) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 7) || (epilogue != EPI_NONE && epilogue != EPI_SWIGLU) ||
        (epilogue == EPI_SWIGLU && (N & 31))) {
        padt_set_error("padt_gemm_rmsnorm_bf16: bad C/ldc/epilogue (0 or 3; SwiGLU needs N % 32 == 0)");
        return -1;
    }
    GemmArgs a{(const x16_t*)A, lda, (const x16_t*)W, ldw, (const x16_t*)bias, C, ldc, nullptr, 0, (int)M, (int)N, (int)K};
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) dispatch_norm<EPI_SWIGLU>(a, eps, s);
    else dispatch_norm<EPI_NONE>(a, eps, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// Same kernel over the fragment-packed weight image (see include/padt_hip.h): the decode step's projections.
template <int EPI, bool NORM, int WQ = 0, bool F32 = false>
static void dispatch_packed(const GemmArgs& a, float eps, hipStream_t s) {
    if (a.M <= 16) launch_skinny<1, EPI, F32, NORM, true, WQ>(a, eps, s);
    else if (a.M <= 32) launch_skinny<2, EPI, F32, NORM, true, WQ>(a, eps, s);
    else if (a.M <= 64) launch_skinny<4, EPI, F32, NORM, true, WQ>(a, eps, s);
    else launch_skinny<8, EPI, F32, NORM, true, WQ>(a, eps, s);
}

static long splitk_ticket_bytes(long N) { return (((N + 15) / 16 * 4 + 255) / 256) * 256; }

static int gemm_packed_impl(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* bias, void* C,
                            long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                            int split_k, void* workspace, int act_packed, const float* wscale, void* Xb = nullptr, long ldxb = 0) {
    if (M <= 0 || N <= 0) return 0;
    const bool resid32 = Xb != nullptr;          // padt_gemm_packed_resid32: C = R = fp32 row-major stream, Xb = its packed bf16 mirror
    if (resid32 && (epilogue != EPI_RESID || (act_packed & 2) || (ldxb & 7) || ((uintptr_t)Xb & 15) || ((uintptr_t)R & 15) || R != C)) {
        padt_set_error("padt_gemm_packed_resid32: in-place fp32 stream (16-byte aligned) and a 16-byte aligned packed mirror with ldxb % 8 == 0 required");
        return -1;
    }
    if (wscale && ((Kp & 63) || (N & 15) || ((uintptr_t)wscale & 15))) {
        padt_set_error("padt_gemm_packed_fp8: Kp % 64 == 0, N % 16 == 0 and 16-byte aligned scales required");
        return -1;
    }
    if ((act_packed & ~3) || ((act_packed & 1) && (lda & 7)) || ((act_packed & 2) && ((ldc & 7) || (R != nullptr && ldr != ldc)))) {
        padt_set_error("padt_gemm_packed_bf16: act_packed bit 0 = A packed (lda % 8), bit 1 = C and R packed (ldc % 8, ldr == ldc)");
        return -1;
    }
    if (split_k > 1 && (workspace == nullptr || split_k > 8 || norm_eps >= 0.f || epilogue == EPI_SWIGLU)) {
        padt_set_error("padt_gemm_packed_bf16: split_k in [2, 8] needs a workspace, no fused norm and epilogue 0 or 2");
        return -1;
    }
    if (M > 128 || K <= 0 || (K & 7) || K > Kp || (Kp & 31) || (lda & 7) || ((uintptr_t)A & 15) || ((uintptr_t)Wp & 15)) {
        padt_set_error("padt_gemm_packed_bf16: M <= 128, K % 8 == 0, K <= Kp, Kp % 32 == 0, 16-byte aligned A/Wp required");
        return -1;
    }
    const bool norm = norm_eps >= 0.f;
    if ((ldc & 3) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 7) || (epilogue != EPI_NONE && epilogue != EPI_RESID && epilogue != EPI_SWIGLU) ||
        (epilogue == EPI_SWIGLU && (N & 31)) || (epilogue == EPI_RESID && (R == nullptr || (ldr & 3) || ((uintptr_t)R & 7) || norm))) {
        padt_set_error("padt_gemm_packed_bf16: bad C/ldc/bias/epilogue (0, 2 without norm, or 3 with N % 32 == 0)");
        return -1;
    }
    GemmArgs a{(const x16_t*)A, lda, (const x16_t*)Wp, Kp, (const x16_t*)bias, C, ldc, (const x16_t*)R, ldr, (int)M, (int)N, (int)K};
    a.a_pack = act_packed & 1;
    a.c_pack = (act_packed >> 1) & 1;
    if (resid32) { a.r_f32 = 1; a.C2 = (x16_t*)Xb; a.ldc2 = ldxb; a.c_pack = 1; }   // c_pack addresses the MIRROR here (R / C are fp32 row-major)
    if (split_k > 1) {
        a.ticket = reinterpret_cast<int*>(workspace);
        a.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + splitk_ticket_bytes(N));
        a.split = split_k;
    }
    hipStream_t s = (hipStream_t)stream;
    if (resid32) {
        a.cs = wscale;
        if (wscale) dispatch_packed<EPI_RESID, false, 1, true>(a, 0.f, s);
        else dispatch_packed<EPI_RESID, false, 0, true>(a, 0.f, s);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { padt_set_error(hipGetErrorString(e2)); return -2; }
        return 0;
    }
    if (wscale) {
        a.cs = wscale;
        if (epilogue == EPI_SWIGLU) { if (norm) dispatch_packed<EPI_SWIGLU, true, 1>(a, norm_eps, s); else dispatch_packed<EPI_SWIGLU, false, 1>(a, 0.f, s); }
        else if (epilogue == EPI_RESID) dispatch_packed<EPI_RESID, false, 1>(a, 0.f, s);
        else { if (norm) dispatch_packed<EPI_NONE, true, 1>(a, norm_eps, s); else dispatch_packed<EPI_NONE, false, 1>(a, 0.f, s); }
    } else if (epilogue == EPI_SWIGLU) { if (norm) dispatch_packed<EPI_SWIGLU, true>(a, norm_eps, s); else dispatch_packed<EPI_SWIGLU, false>(a, 0.f, s); }
    else if (epilogue == EPI_RESID) dispatch_packed<EPI_RESID, false>(a, 0.f, s);
    else { if (norm) dispatch_packed<EPI_NONE, true>(a, norm_eps, s); else dispatch_packed<EPI_NONE, false>(a, 0.f, s); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

extern "C" int PADT_SYM(padt_gemm_packed_, )(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* bias, void* C,
                                     long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                                     int split_k, void* workspace, int act_packed) {
    return gemm_packed_impl(stream, A, lda, Wp, Kp, bias, C, ldc, R, ldr, M, N, K, epilogue, norm_eps, split_k, workspace, act_packed, nullptr);
}

// Same decode-step projection over fp8 weights (BASELINE configs[4], the 7B "fp8 MFMA weight path"): Wq = OCP e4m3 bytes in the fp8
// fragment-packed image [N/16][Kp/64][64 lanes][16 B] (ops.pack_weight_fp8), scales fp32 [N] (one per weight row; powers of two in
// weights.py, so the bf16 prefill copy of the same matrix is bit-consistent).  C = epi(rstd?(A) * scale[n] * (A · Wq^T) + bias).
extern "C" int PADT_TWIN(padt_gemm_packed_fp8)(void* stream, const void* A, long lda, const void* Wq, long Kp, const void* scales, const void* bias,
                                    void* C, long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                                    int split_k, void* workspace, int act_packed) {
    if (scales == nullptr) { padt_set_error("padt_gemm_packed_fp8: scales are required"); return -1; }
    return gemm_packed_impl(stream, A, lda, Wq, Kp, bias, C, ldc, R, ldr, M, N, K, epilogue, norm_eps, split_k, workspace, act_packed,
                            (const float*)scales);
}

// Decode-step residual projection (o_proj, down_proj: HF:741,757 at one token per row) over the fp32 residual stream:
// X32[m][n] += scale?[n] * (A · W^T)[m][n]  in place (fp32 row-major), Xb = bf16(X32) in the fragment-packed activation layout — the A operand
// of the next projection.  Wp: bf16 fragment-packed weights, or (scales != null) the fp8 image of padt_gemm_packed_fp8.
extern "C" int PADT_TWIN(padt_gemm_packed_resid32)(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* scales, void* X32, long ldx,
                                        void* Xb, long ldxb, long M, long N, long K, int split_k, void* workspace, int a_packed) {
    if (Xb == nullptr) { padt_set_error("padt_gemm_packed_resid32: the packed mirror is required"); return -1; }
    return gemm_packed_impl(stream, A, lda, Wp, Kp, nullptr, X32, ldx, X32, ldx, M, N, K, EPI_RESID, -1.0f, split_k, workspace, a_packed ? 1 : 0,
                            (const float*)scales, Xb, ldxb);
}

// fp8 x fp8 MFMA GEMM at prompt length (gemm256.hip, FP8 instantiations of the phase-pipelined kernel); takes a profile slot like every tile GEMM.
extern "C" int PADT_TWIN(padt_gemm_fp8_impl)(void* stream, const void* A8, long lda, const void* W8, long ldw, const void* row_scale, const void* col_scale,
                                  const void* bias, void* C, long ldc, void* X32, long ldx, void* Xb, long ldxb, long M, long N, long K, int epilogue,
                                  unsigned long long* prof);
extern "C" int PADT_TWIN(padt_gemm_fp8)(void* stream, const void* A8, long lda, const void* W8, long ldw, const void* row_scale, const void* col_scale,
                             const void* bias, void* C, long ldc, void* X32, long ldx, void* Xb, long ldxb, long M, long N, long K, int epilogue) {
    return PADT_TWIN(padt_gemm_fp8_impl)(stream, A8, lda, W8, ldw, row_scale, col_scale, bias, C, ldc, X32, ldx, Xb, ldxb, M, N, K, epilogue,
                              (M > 64) ? next_prof_slot() : nullptr);
}

}  // namespace PADT_NS
