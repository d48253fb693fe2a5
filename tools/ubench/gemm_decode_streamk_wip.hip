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

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_SWIGLU = 3 };

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;          // [N] or null
    void* C; long ldc;           // bf16 or f32
    const bf16_t* R; long ldr;   // residual (EPI_RESID)
    int M, N, K;
    float* ws = nullptr;         // split-K (skinny kernel, gridDim.y > 1): partial accumulators [nb][split][frags][64][4]
    int* ticket = nullptr;       //          and one completion ticket per n-block (zero between launches)
    int split = 1;
    const float* rs = nullptr;   // optional per-row scale applied to the accumulator before bias (fused RMSNorm: rstd[m])
    RopeEpi rope = {nullptr, nullptr, 0, 0, 0};   // optional fused RoPE of the leading output columns (EPI_NONE only)
    int a_pack = 0;              // skinny kernel: A / (C and R) stored in the 16-row fragment-packed activation layout
    int c_pack = 0;              //   element (m, k) at (m/16)*16*ld + ((k/8)*16 + m%16)*8 + k%8   (see padt_hip.h)
    int r_f32 = 0;               // EPI_RESID: R is fp32 [M][ldr] (fp32 residual stream; with OUT_F32)
    long lo_off = 0;             // bf16 output: also store lo = bf16(x - hi) at C + lo_off (split-precision pair, padt_gemm_bf16_ex)
    int dec_tiles = 0;           // gemm_decode_kernel: number of 64-output tiles and partial-slab slots per tile
    int dec_slots = 0;
};

// bf16 split pair of 4 fp32 values: hi = bf16(x), lo = bf16(x - hi)  (hi + lo carries 16 mantissa bits)
PADT_DEV void split4(const float* o, u32x2& hi, u32x2& lo) {
    hi = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    float h[4];
    h[0] = __builtin_bit_cast(float, hi[0] << 16);
    h[1] = __builtin_bit_cast(float, hi[0] & 0xffff0000u);
    h[2] = __builtin_bit_cast(float, hi[1] << 16);
    h[3] = __builtin_bit_cast(float, hi[1] & 0xffff0000u);
    lo = u32x2{pack2bf(o[0] - h[0], o[1] - h[1]), pack2bf(o[2] - h[2], o[3] - h[3])};
}

// offset of element (m, n) of a row-major or fragment-packed [rows][ld] activation matrix (n % 4 == 0 keeps 4 elements together)
PADT_DEV long act_index(int m, int n, long ld, int packed) {
    return packed ? (long)(m >> 4) * 16 * ld + ((long)(n >> 3) * 16 + (m & 15)) * 8 + (n & 7) : (long)m * ld + n;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];   // 256 B of zeros (K-tail source)

// ---------------------------------------------------------------------------------------------------------------------
// epilogue for one 16x16 fragment held "swapped": lane has row m, columns n..n+3 in v[0..3]
PADT_DEV void unpack4(u32x2 v, float* f) {
    f[0] = __builtin_bit_cast(float, v[0] << 16);
    f[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, v[1] << 16);
    f[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
}

template <int EPI, bool OUT_F32>
PADT_DEV void store_frag(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M || n >= p.N) return;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (p.rs) {
        const float sc = p.rs[m];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] *= sc;
    }
    if (n + 3 < p.N) {                                   // full fragment: 8-byte bias / residual loads, one vector store
        const bf16_t* bp = p.bias ? p.bias + n : reinterpret_cast<const bf16_t*>(g_zero_page);
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
        if (OUT_F32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
        else if (p.lo_off) {
            u32x2 hi, lo;
            split4(o, hi, lo);
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
            *reinterpret_cast<u32x2*>(c) = hi;
            *reinterpret_cast<u32x2*>(c + p.lo_off) = lo;
        }
        else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + act_index(m, n, p.ldc, p.c_pack)) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        return;
    }
    for (int r = 0; r < 4 && n + r < p.N; ++r) {         // ragged N tail: scalar
        float x = o[r];
        if (p.bias) x += bf2f(p.bias[n + r]);
        if (EPI == EPI_GELU) x = gelu_erf(x);
        if (EPI == EPI_RESID) x += p.r_f32 ? reinterpret_cast<const float*>(p.R)[(long)m * p.ldr + n + r] : bf2f(p.R[(long)m * p.ldr + n + r]);
        if (OUT_F32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + n + r] = x;
        else {
            const bf16_t h = f2bf(x);
            reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + n + r] = h;
            if (p.lo_off) reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + n + r + p.lo_off] = f2bf(x - bf2f(h));
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
    const bf16_t* bp = p.bias ? p.bias + n_gate : reinterpret_cast<const bf16_t*>(g_zero_page);
    unpack4(*reinterpret_cast<const u32x2*>(bp), gb);
    unpack4(*reinterpret_cast<const u32x2*>(bp + 16), ub);
    float o[4];
    const float sc = p.rs ? p.rs[m] : 1.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = silu(g[r] * sc + gb[r]) * (u[r] * sc + ub[r]);
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + act_index(m, no, p.ldc, p.c_pack);
    *reinterpret_cast<u32x2*>(c) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
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
PADT_DEV void stage_tile(const bf16_t* __restrict__ base, long ld, int row0, int nrows, int k0, int K,
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
            bf16x8 af[4], wf[4];
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
    if (interior && EPI != EPI_SWIGLU && !p.r_f32 && !p.lo_off) {
        // block-uniform fast path: no per-fragment bounds checks; every bias / residual load is issued up front
        const int mb = m0 + wm * 64 + frow, nb = n0 + wn * 64 + fq * 4;
        u32x2 braw[4], rraw[4][4];
        const bf16_t* bp = p.bias ? p.bias + nb : reinterpret_cast<const bf16_t*>(g_zero_page);
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
                else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + off) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
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
template <int MT, int NT, int NW, int EPI, bool OUT_F32, bool NORM, bool PACKED>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p, float norm_eps) {
    __shared__ __attribute__((aligned(16))) float red[NW - 1][NT * MT][64][4];
    __shared__ __attribute__((aligned(16))) float red0[NT * (MT > 1 ? MT - 1 : 1)][64][4];
    __shared__ float ssq[NW][MT][16];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int nks = (p.K + 31) / 32;
    constexpr int U = (MT <= 2) ? 8 : (MT * NT >= 8 ? 2 : 4);   // K-steps in flight per wave (VGPR budget; deeper at MT = 4 measured slower: occupancy)

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ss[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) ss[j] = 0.f;

    const bf16_t* wrow[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int n = n0 + i * 16 + frow;
        n = n < p.N ? n : p.N - 1;
        wrow[i] = p.W + (long)n * p.ldw;
    }
    // x fragment of K-step ks = 16 bytes at xrow[j] + ks * xstep.  Row-major activations: 16 rows x 64 B per wave
    // instruction (address unit at a quarter rate — the limiter once 32 rows are decoded together); packed activations
    // (a_pack): the same fragment is 1 KiB contiguous in lane order.
    const bf16_t* xrow[MT];
    bool xok[MT];
    const int xstep = p.a_pack ? 512 : 32;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = j * 16 + frow;
        xok[j] = p.a_pack ? true : (m < p.M);
        xrow[j] = p.a_pack ? p.A + (long)j * 16 * p.lda + lane * 8 : p.A + (long)(xok[j] ? m : 0) * p.lda + fq * 8;
    }

    // wave w owns groups w, w+NW, ... of U CONSECUTIVE K-steps: one round = U*64 B contiguous per weight row
    // split-K: gridDim.y blocks share an n-block, block y takes groups y*NW + wave, stepping by NW*gridDim.y
    for (int grp = blockIdx.y * NW + wave; grp * U < nks; grp += NW * gridDim.y) {
        bf16x8 wf[U][NT], xf[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = grp * U + u;
            const int k = ks * 32 + fq * 8;
            const bool kok = (ks < nks) && (k < p.K);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                // PACKED: tile (n16, k32) of the fragment-packed image is 1 KiB in lane order → one contiguous wave load
                const bf16_t* wp = PACKED ? p.W + ((long)(n0 / 16 + i) * (p.ldw / 32) + ks) * 512 + lane * 8 : wrow[i] + k;
                wf[u][i] = kok ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp)) : zero_frag();
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[u][j] = (kok && xok[j]) ? ld_frag(xrow[j] + (long)ks * xstep) : zero_frag();
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
                for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[u][i], xf[u][j], acc[i][j]);
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
    // Cross-wave reduction + epilogue, spread over the waves: wave j (< MT) sums the NW partials of batch row block j and
    // runs its epilogue — with 64 decode rows the tail is as long as the K loop, one wave doing all of it left the other
    // waves of the block idle.  Partials are summed in wave order (the order the single-wave version used).
    constexpr int NWK = NW - 1;
    const bool worker = wave < MT;
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(&red[wave - 1][i * MT + j][lane][0]) = acc[i][j];
    }
    if (MT > 1 && wave == 0) {                                    // wave 0's partials of the row blocks other waves finish
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 1; j < MT; ++j) *reinterpret_cast<f32x4*>(&red0[i * (MT - 1) + (j - 1)][lane][0]) = acc[i][j];
    }
    __syncthreads();
    f32x4 sum[NT];
    const int jw = worker ? wave : 0;                             // row block this wave finishes
    if (worker) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            sum[i] = (wave == 0) ? acc[i][0] : *reinterpret_cast<f32x4*>(&red0[i * (MT - 1) + (jw - 1)][lane][0]);
#pragma unroll
            for (int w = 0; w < NWK; ++w) sum[i] += *reinterpret_cast<f32x4*>(&red[w][i * MT + jw][lane][0]);
        }
    }
    if (!NORM && gridDim.y > 1) {
        // split-K: the last of the n-block's split blocks to arrive sums the partials and runs the epilogue
        const int S = gridDim.y;
        if (worker) {
            float* mine = p.ws + (((long)(blockIdx.x * S + blockIdx.y) * (NT * MT)) * 64 + lane) * 4;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) st_agent(mine + (i * MT + jw) * 256 + r, sum[i][r]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partial stores are acknowledged
        }
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(&p.ticket[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (old == S - 1);
            if (last) __hip_atomic_store(&p.ticket[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag = last;
        }
        __syncthreads();
        if (!flag) return;
        if (worker) {
            for (int y = 0; y < S; ++y) {
                if (y == (int)blockIdx.y) continue;
                const float* other = p.ws + (((long)(blockIdx.x * S + y) * (NT * MT)) * 64 + lane) * 4;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[i][r] += ld_agent(other + (i * MT + jw) * 256 + r);
            }
        }
    }
    if (!worker) return;
    const int m = jw * 16 + frow;
    if (NORM) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += ssq[w][jw][frow];
        const float rstd = rsqrtf(t / (float)p.K + norm_eps);
#pragma unroll
        for (int i = 0; i < NT; ++i) sum[i] *= rstd;
    }
    if (EPI == EPI_SWIGLU) {
        store_swiglu(p, m, n0 + fq * 4, sum[0], sum[NT - 1]);
    } else {
#pragma unroll
        for (int i = 0; i < NT; ++i) store_frag<EPI, OUT_F32>(p, m, n0 + i * 16 + fq * 4, sum[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode kernel for 17..64 rows (in-flight batching: the decode steps of several batches share one weight pass).
//
// Measured on MI355X (profiles/r02_decode_gemm_experiments.md): ONE CU pulls at most ≈24.5 GB/s from HBM (256 x 24.5 = 6.3 TB/s,
// the chip's achievable rate), so a weight-streaming kernel is at speed only when all 256 CUs stream equal shares with ≥64 KiB
// in flight each.  gemm_skinny_kernel gives every 16 weight rows their own block whose waves split K and each pull the whole
// [rows x K] activation slab through the vector memory pipe: at 64 rows that is 2-4x the weight bytes per CU and (register
// pressure) only 2 K-steps in flight per wave — 3.5 TB/s on gate/up, 1.1 TB/s on the small projections.  This kernel instead:
//   * stream-K: the (tile, K-step) space — tile = 64 outputs: 4 N-waves x NT 16-row weight blocks — is cut into gridDim.x = 256
//     EQUAL contiguous ranges, one block per CU; a range covers the tail of one tile and/or the head of the next ("segments");
//   * a block is KG K-groups x 4 N-waves: the groups split a segment's K-steps, the 4 N-waves of a group SHARE the activation
//     fragments of a chunk of CH K-steps through LDS (LDS-DMA, 1 KiB contiguous pieces of the packed layout; x traffic = 1x the
//     weight bytes instead of 4x); weights go HBM → VGPR (non-temporal), the next chunk's loads are in flight while this chunk's
//     MFMAs run; groups are summed through LDS in group order;
//   * tiles that span several blocks exchange fp32 partial slabs through write-through (sc1) 16-byte stores + one ticket per
//     tile; the last arriver sums the slabs IN K ORDER (its own from registers at its place): bit-reproducible for a given shape
//     whatever the arrival order.  The fused RMSNorm's Σx² partials travel with the slabs.
template <int MT, int NT, int CH, int KG, int EPI, bool NORM>
__global__ __launch_bounds__(256 * KG) void gemm_decode_kernel(GemmArgs p, float norm_eps) {
    // LDS: KG x [2][CH][MT] KiB x rings (re-used for the cross-group reduction) | float ssq[KG][64] | int flag
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CHUNK_BYTES = CH * MT * 1024;
    constexpr int PPW = CH * MT / 4;                                       // DMA pieces per wave per chunk
    constexpr int SLAB = 4 * NT * MT * 256 + 64;                           // floats: 4 waves x NT*MT fragments, then 64 Σx² partials
    static_assert((CH * MT) % 4 == 0, "pieces must divide over the 4 waves of a K-group");
    static_assert((KG - 1) * 4 * NT * MT * 1024 <= KG * 2 * CHUNK_BYTES || KG == 1, "cross-group reduction buffer aliases the rings");
    float* ssq = reinterpret_cast<float*>(smem + KG * 2 * CHUNK_BYTES);
    int* flag = reinterpret_cast<int*>(ssq + KG * 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform (scalar branches / addresses)
    const int kg = wave >> 2, nw = wave & 3;                               // K-group, N-wave inside the group
    const int frow = lane & 15, fq = lane >> 4;
    const int nks = p.K >> 5;                                              // K-steps per tile
    const int kpt = (int)(p.ldw >> 5);                                     // K-steps per weight row block in the packed image
    const long G = gridDim.x, Ltot = (long)p.dec_tiles * nks;
    const long r0 = (long)blockIdx.x * Ltot / G, r1 = (long)(blockIdx.x + 1) * Ltot / G;    // this block's range of (tile, K-step)
    char* rings = smem + kg * 2 * CHUNK_BYTES;

    for (long pos = r0; pos < r1;) {
        const int tile = (int)(pos / nks);
        const int b0 = (int)(pos - (long)tile * nks);                      // segment = K-steps [b0, b1) of `tile`
        const int b1 = (int)((r1 - (long)tile * nks) < nks ? (r1 - (long)tile * nks) : nks);
        pos += b1 - b0;
        // contributors of this tile: blocks first..last; slot = this block's place in K order
        const int first = (int)((((long)tile * nks + 1) * G - 1) / Ltot), last = (int)((((long)tile * nks + nks) * G - 1) / Ltot);
        const int S = last - first + 1, sl = (int)blockIdx.x - first;
        const int ks0 = b0 + (int)((long)(b1 - b0) * kg / KG), ks1 = b0 + (int)((long)(b1 - b0) * (kg + 1) / KG);   // this group's K-steps
        const int nit = (((b1 - b0) + KG - 1) / KG + CH - 1) / CH;         // chunk iterations, the same for every group
        const int nb16 = (tile * 4 + nw) * NT;                             // first 16-row weight block of this wave
        const bool wave_ok = nb16 * 16 < p.N;

        f32x4 acc[NT][MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float ss = 0.f;

        auto dma_chunk = [&](int kbase, int buf) {                         // x fragments of K-steps [kbase, kbase + CH) → ring[buf]
            if (kbase >= ks1) return;
            const bool full = kbase + CH <= ks1;                           // group-uniform: no per-piece guards on the common path
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int piece = nw * PPW + i;
                const int u = piece / MT, j = piece % MT;
                const int ks = kbase + u;
                if (full || ks < ks1) {
                    const bf16_t* src = p.A + (long)j * 16 * p.lda + (long)ks * 512 + lane * 8;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(rings + buf * CHUNK_BYTES + piece * 1024), 16, 0, 0);
                }
            }
        };
        auto load_w = [&](int kbase, bf16x8 (&wf)[CH][NT]) {
            const bf16_t* wbase = p.W + ((long)nb16 * kpt + kbase) * 512 + lane * 8;
            if (kbase + CH <= ks1 && wave_ok) {                            // common path: CH*NT unconditional 1-KiB wave loads
#pragma unroll
                for (int u = 0; u < CH; ++u)
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        wf[u][i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wbase + ((long)i * kpt + u) * 512));
                return;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    wf[u][i] = (kbase + u < ks1 && wave_ok) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wbase + ((long)i * kpt + u) * 512))
                                                            : zero_frag();
        };
        auto compute = [&](int kbase, int buf, bf16x8 (&wf)[CH][NT]) {
            const char* ring = rings + buf * CHUNK_BYTES + lane * 16;
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (kbase + u < ks1) {
                    bf16x8 xf[MT];
#pragma unroll
                    for (int j = 0; j < MT; ++j) xf[j] = ld_frag(ring + (u * MT + j) * 1024);
                    if (NORM && nw < MT) {                                 // N-wave j owns the Σx² of row block j (its own LDS read: a
                        float xv[8];                                       // runtime index into xf[] would cost a select chain per register)
                        unpack8(__builtin_bit_cast(u32x4, ld_frag(ring + (u * MT + nw) * 1024)), xv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss += xv[e] * xv[e];
                    }
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[u][i], xf[j], acc[i][j]);
                }
            }
        };

        // software pipeline over chunks: weights and x of chunk c + 1 are requested before chunk c is multiplied
        bf16x8 wa[CH][NT], wb[CH][NT];
        dma_chunk(ks0, 0);
        load_w(ks0, wa);
        int kb = ks0;
        for (int it = 0; it < nit; it += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                              // everyone's pieces of this chunk landed; ring[1] is free
            dma_chunk(kb + CH, 1);
            load_w(kb + CH, wb);
            compute(kb, 0, wa);
            kb += CH;
            if (it + 1 >= nit) break;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            dma_chunk(kb + CH, 0);
            load_w(kb + CH, wa);
            compute(kb, 1, wb);
            kb += CH;
        }

        if (NORM) {
            float t = ss;
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            ss = t;                                                        // lanes of one frow hold the row's partial sum (N-waves < MT)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                  // all ring reads done: the rings become the exchange buffer
        if (KG > 1) {
            // ---- cross-group reduction through LDS (groups 1.. hand their partials to group 0, summed in group order)
            float* red = reinterpret_cast<float*>(smem);
            if (kg > 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        *reinterpret_cast<f32x4*>(red + ((((kg - 1) * 4 + nw) * NT + i) * MT + j) * 256 + lane * 4) = acc[i][j];
                if (NORM && nw < MT && fq == 0) ssq[kg * 64 + nw * 16 + frow] = ss;
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int g = 1; g < KG; ++g) {
#pragma unroll
                    for (int i = 0; i < NT; ++i)
#pragma unroll
                        for (int j = 0; j < MT; ++j)
                            acc[i][j] += *reinterpret_cast<const f32x4*>(red + ((((g - 1) * 4 + nw) * NT + i) * MT + j) * 256 + lane * 4);
                    if (NORM && nw < MT) ss += ssq[g * 64 + nw * 16 + frow];
                }
            }
        }
        // from here on group 0 works; the other groups only keep the barriers company
        float tot[MT];                                                     // Σx² of row j*16 + frow over the whole K (NORM)
#pragma unroll
        for (int j = 0; j < MT; ++j) tot[j] = 0.f;
        bool finish = true;                                               // this block writes the tile's output
        if (S > 1) {
            // ---- split-K hand-off: [tile][slot] slabs
            float* slab0 = p.ws + (long)tile * p.dec_slots * SLAB;
            auto rs = __builtin_amdgcn_make_buffer_rsrc(slab0, 0, (int)((long)S * SLAB * 4), 0x00020000);
            const int mine = sl * SLAB;
            if (kg == 0) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, (mine + ((nw * NT + i) * MT + j) * 256 + lane * 4) * 4, 0, 16);   // aux 16 = sc1 (write-through)
                if (NORM && nw < MT && fq == 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ss), rs, (mine + 4 * NT * MT * 256 + nw * 16 + frow) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its write-through stores
            __syncthreads();
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(&p.ticket[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int lastone = (old == S - 1);
                if (lastone) __hip_atomic_store(&p.ticket[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = lastone;
            }
            __syncthreads();
            finish = *flag != 0;
            if (finish && kg == 0) {
                f32x4 sum[NT][MT];
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) sum[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                constexpr int RB = (KG == 4) ? (NT == 2 ? 1 : 2) : (NT == 2 ? 2 : 4);   // slots whose loads are in flight together (VGPR budget)
                for (int y0 = 0; y0 < S; y0 += RB) {                       // K order, own partial from registers at its place
                    f32x4 v[RB][NT][MT];
#pragma unroll
                    for (int dy = 0; dy < RB; ++dy) {
                        const int y = y0 + dy;
#pragma unroll
                        for (int i = 0; i < NT; ++i)
#pragma unroll
                            for (int j = 0; j < MT; ++j) {
                                v[dy][i][j] = (y == sl) ? acc[i][j] : f32x4{0.f, 0.f, 0.f, 0.f};
                                if (y < S && y != sl)
                                    v[dy][i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (y * SLAB + ((nw * NT + i) * MT + j) * 256 + lane * 4) * 4, 0, 16));
                            }
                    }
                    float tv[RB][MT];
                    if (NORM) {
#pragma unroll
                        for (int dy = 0; dy < RB; ++dy)
#pragma unroll
                            for (int j = 0; j < MT; ++j)
                                tv[dy][j] = (y0 + dy < S) ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ((y0 + dy) * SLAB + 4 * NT * MT * 256 + j * 16 + frow) * 4, 0, 16)) : 0.f;
                    }
#pragma unroll
                    for (int dy = 0; dy < RB; ++dy) {
                        if (y0 + dy < S) {
#pragma unroll
                            for (int i = 0; i < NT; ++i)
#pragma unroll
                                for (int j = 0; j < MT; ++j) sum[i][j] += v[dy][i][j];
                            if (NORM) {
#pragma unroll
                                for (int j = 0; j < MT; ++j) tot[j] += tv[dy][j];
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[i][j] = sum[i][j];
            }
        } else if (NORM) {
            if (kg == 0 && nw < MT && fq == 0) ssq[nw * 16 + frow] = ss;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT; ++j) tot[j] = ssq[j * 16 + frow];
        }
        if (finish && kg == 0 && wave_ok) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int m = j * 16 + frow;
                const float rstd = NORM ? rsqrtf(tot[j] / (float)p.K + norm_eps) : 1.f;
                if (EPI == EPI_SWIGLU) {
                    store_swiglu(p, m, nb16 * 16 + fq * 4, acc[0][j] * rstd, acc[NT - 1][j] * rstd);
                } else {
#pragma unroll
                    for (int i = 0; i < NT; ++i) store_frag<EPI, false>(p, m, (nb16 + i) * 16 + fq * 4, acc[i][j] * rstd);
                }
            }
        }
        __syncthreads();                                                  // LDS (rings / exchange buffer / ssq / flag) is free for the next segment
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
extern "C" void padt_set_error(const char* msg);
// gemm256.hip: phase-pipelined 256x256 kernel for large-N shapes; returns 0 if it took the launch
extern "C" int padt_gemm256_try(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                                const float* row_scale, const RopeEpi* rope, long* rows_done, int resid_f32, long lo_off);

template <int EPI, bool F32, int BK>
static void launch_tile_bk(const GemmArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<EPI, F32, BK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<BK>::LDS);
        attr_done = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_tile_kernel<EPI, F32, BK>), dim3(ntm * ntn), dim3(256), TileCfg<BK>::LDS, s, a);
}

template <int EPI, bool F32>
static void launch_tile(const GemmArgs& a, hipStream_t s) {
    static const int bk = getenv("PADT_TILE_BK") ? atoi(getenv("PADT_TILE_BK")) : 64;        // tuning knob
    if (bk == 32) launch_tile_bk<EPI, F32, 32>(a, s);
    else launch_tile_bk<EPI, F32, 64>(a, s);
}

template <int MT, int NW, int EPI, bool F32, bool NORM, bool PACKED = false>
static void launch_skinny_nw(const GemmArgs& a, float eps, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int nb = (a.N + 16 * NT - 1) / (16 * NT);
    const int split = (a.ws && !NORM) ? a.split : 1;
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, NW, EPI, F32, NORM, PACKED>), dim3(nb, split), dim3(NW * 64), 0, s, a, eps);
}

// waves per block: enough waves chip-wide (>= ~2048) to keep HBM busy even when N/16 < #CUs
template <int MT, int EPI, bool F32, bool NORM, bool PACKED = false>
static void launch_skinny(const GemmArgs& a, float eps, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int nb = (a.N + 16 * NT - 1) / (16 * NT);
    const int ksteps = (a.K + 31) / 32 / (a.ws ? a.split : 1);   // per block; each wave keeps U = 8 K-steps in flight
    static const int force_nw = getenv("PADT_SKINNY_NW") ? atoi(getenv("PADT_SKINNY_NW")) : 0;   // tuning knob
    if constexpr (MT == 1 && NT == 1) {
        if (force_nw == 16 || (!force_nw && nb <= 256 && ksteps >= 128)) { launch_skinny_nw<MT, 16, EPI, F32, NORM, PACKED>(a, eps, s); return; }
    }
    if (force_nw == 8 || (!force_nw && nb <= 512 && ksteps >= 64)) launch_skinny_nw<MT, 8, EPI, F32, NORM, PACKED>(a, eps, s);
    else launch_skinny_nw<MT, 4, EPI, F32, NORM, PACKED>(a, eps, s);
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
                          const void* row_scale, const RopeEpi& rope, int resid_f32 = 0, long lo_off = 0) {
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
    if (lo_off && (out_f32 || epilogue == EPI_SWIGLU || (lo_off & 3) || lo_off < N || ldc < lo_off + N)) {
        padt_set_error("padt_gemm_bf16_ex: a split (hi|lo) output needs bf16 output, lo_off % 4 == 0, lo_off >= N and ldc >= lo_off + N");
        return -1;
    }
    if (M > 64 && padt_gemm256_try(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, out_f32, rs, &rp, &done, resid_f32, lo_off) == 0) {
        if (done >= M) {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
            return 0;
        }
        // a peeled ragged tail (<= 64 rows): the rest of this function streams it through the skinny kernel
        A = (const bf16_t*)A + done * lda;
        C = out_f32 ? (void*)((float*)C + done * ldc) : (void*)((bf16_t*)C + done * ldc);
        if (R) R = resid_f32 ? (const void*)((const float*)R + done * ldr) : (const void*)((const bf16_t*)R + done * ldr);
        if (rs) rs += done;
        if (rp.cos) { rp.cos += done * rp.ld; rp.sin += done * rp.ld; }
        M -= done;
    }
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, C, ldc, (const bf16_t*)R, ldr,
               (int)M, (int)N, (int)K};
    a.rs = rs;
    a.rope = rp;
    a.r_f32 = resid_f32;
    a.lo_off = lo_off;
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

extern "C" int padt_gemm_bf16(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                              long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                              const void* row_scale) {
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, out_f32, row_scale,
                          RopeEpi{nullptr, nullptr, 0, 0, 0});
}

// Extended epilogues for the split-precision PaDT decoder (decoder_hp.hip): resid_f32 — R (and C: out_f32 must be set) is the
// fp32 residual stream; lo_off != 0 — bf16 output stored as a (hi, lo) pair, hi at C[m][n], lo = bf16(x - hi) at C[m][lo_off + n],
// i.e. directly the [hi | lo] A operand (K' = 2N) of the next GEMM whose weight image is [W | W].
extern "C" int padt_gemm_bf16_ex(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                 long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                                 const void* row_scale, int resid_f32, long lo_off) {
    return gemm_bf16_impl(stream, A, lda, W, ldw, bias, C, ldc, R, ldr, M, N, K, epilogue, out_f32, row_scale,
                          RopeEpi{nullptr, nullptr, 0, 0, 0}, resid_f32, lo_off);
}

// C = rope(row_scale[m] * (A · W^T) + bias): the rotate-half RoPE of the leading `rope_cols` output columns fused into the
// epilogue (ViT qkv projection: q and k columns, pair-interleaved per head by a load-time permutation of W's rows).
extern "C" int padt_gemm_rope_bf16(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
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
extern "C" int padt_gemm_rmsnorm_bf16(void* stream, const void* A, long lda, float eps, const void* W, long ldw,
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
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, C, ldc, nullptr, 0, (int)M, (int)N, (int)K};
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) dispatch_norm<EPI_SWIGLU>(a, eps, s);
    else dispatch_norm<EPI_NONE>(a, eps, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// Same kernel over the fragment-packed weight image (see include/padt_hip.h): the decode step's projections.
template <int EPI, bool NORM>
static void dispatch_packed(const GemmArgs& a, float eps, hipStream_t s) {
    if (a.M <= 16) launch_skinny<1, EPI, false, NORM, true>(a, eps, s);
    else if (a.M <= 32) launch_skinny<2, EPI, false, NORM, true>(a, eps, s);
    else launch_skinny<4, EPI, false, NORM, true>(a, eps, s);
}

static long splitk_ticket_bytes(long N) { return (((N + 15) / 16 * 4 + 255) / 256) * 256; }

extern "C" long padt_gemm_splitk_workspace(long N, int split_k) {
    return splitk_ticket_bytes(N) + (N + 15) / 16 * (long)split_k * 4 * 64 * 16;   // up to 4 row blocks of fp32 fragments
}

extern "C" int padt_gemm_packed_bf16(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* bias, void* C,
                                     long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                                     int split_k, void* workspace, int act_packed) {
    if (M <= 0 || N <= 0) return 0;
    if ((act_packed & ~3) || ((act_packed & 1) && (lda & 7)) || ((act_packed & 2) && ((ldc & 7) || (R != nullptr && ldr != ldc)))) {
        padt_set_error("padt_gemm_packed_bf16: act_packed bit 0 = A packed (lda % 8), bit 1 = C and R packed (ldc % 8, ldr == ldc)");
        return -1;
    }
    if (split_k > 1 && (workspace == nullptr || split_k > 8 || norm_eps >= 0.f || epilogue == EPI_SWIGLU)) {
        padt_set_error("padt_gemm_packed_bf16: split_k in [2, 8] needs a workspace, no fused norm and epilogue 0 or 2");
        return -1;
    }
    if (M > 64 || K <= 0 || (K & 7) || K > Kp || (Kp & 31) || (lda & 7) || ((uintptr_t)A & 15) || ((uintptr_t)Wp & 15)) {
        padt_set_error("padt_gemm_packed_bf16: M <= 64, K % 8 == 0, K <= Kp, Kp % 32 == 0, 16-byte aligned A/Wp required");
        return -1;
    }
    const bool norm = norm_eps >= 0.f;
    if ((ldc & 3) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 7) || (epilogue != EPI_NONE && epilogue != EPI_RESID && epilogue != EPI_SWIGLU) ||
        (epilogue == EPI_SWIGLU && (N & 31)) || (epilogue == EPI_RESID && (R == nullptr || (ldr & 3) || ((uintptr_t)R & 7) || norm))) {
        padt_set_error("padt_gemm_packed_bf16: bad C/ldc/bias/epilogue (0, 2 without norm, or 3 with N % 32 == 0)");
        return -1;
    }
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)Wp, Kp, (const bf16_t*)bias, C, ldc, (const bf16_t*)R, ldr, (int)M, (int)N, (int)K};
    a.a_pack = act_packed & 1;
    a.c_pack = (act_packed >> 1) & 1;
    if (split_k > 1) {
        a.ticket = reinterpret_cast<int*>(workspace);
        a.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + splitk_ticket_bytes(N));
        a.split = split_k;
    }
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) { if (norm) dispatch_packed<EPI_SWIGLU, true>(a, norm_eps, s); else dispatch_packed<EPI_SWIGLU, false>(a, 0.f, s); }
    else if (epilogue == EPI_RESID) dispatch_packed<EPI_RESID, false>(a, 0.f, s);
    else { if (norm) dispatch_packed<EPI_NONE, true>(a, norm_eps, s); else dispatch_packed<EPI_NONE, false>(a, 0.f, s); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- decode kernel dispatch (17..64 rows, packed weights + packed activations) ------------------------------------------
static long decode_slab_floats(int nt, int mt) { return 4L * nt * mt * 256 + 64; }
static long decode_ticket_bytes(long N) { return (((N + 63) / 64 * 4 + 255) / 256) * 256; }
static int decode_slots(long tiles, long nks, long G) {                  // most blocks whose ranges can touch one tile
    const long lmin = (tiles * nks) / G;                                  // shortest range (G <= tiles * nks)
    return (int)((nks + lmin - 1) / lmin + 1);
}

extern "C" long padt_gemm_decode_workspace(long N, long K, int n_blocks) {
    // worst case over the instantiations: tiles of 64 weight rows (NT = 1) or 128 (SwiGLU, NT = 2), MT = 4
    const long nks = K / 32 > 0 ? K / 32 : 1;
    long best = 0;
    for (int nt = 1; nt <= 2; ++nt) {
        const long tiles = (N + 64 * nt - 1) / (64 * nt);
        long G = n_blocks > 0 ? n_blocks : 256;
        if (G > tiles * nks) G = tiles * nks;
        const long bytes = tiles * decode_slots(tiles, nks, G) * decode_slab_floats(nt, 4) * 4;
        if (bytes > best) best = bytes;
    }
    return decode_ticket_bytes(N) + best;
}

template <int MT, int CH, int KG, int EPI, bool NORM>
static void launch_decode_cfg(const GemmArgs& a, float eps, int G, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int LDS = KG * 2 * CH * MT * 1024 + (KG * 64 + 64) * 4 + 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_decode_kernel<MT, NT, CH, KG, EPI, NORM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_decode_kernel<MT, NT, CH, KG, EPI, NORM>), dim3(G), dim3(256 * KG), LDS, s, a, eps);
}

// K-groups per block: 16 waves (4 groups, chunks of 4 K-steps) where a wave owns one 16-row weight block, 8 waves (2 groups,
// chunks of 8) for the gate/up pairs — 128 KiB of LDS and ≥128 KiB of weight loads in flight per CU either way
template <int MT, int EPI, bool NORM>
static void launch_decode(const GemmArgs& a, float eps, int G, int kg, hipStream_t s) {
    if (kg >= 4) launch_decode_cfg<MT, 4, 4, EPI, NORM>(a, eps, G, s);
    else if (kg == 2) launch_decode_cfg<MT, 8, 2, EPI, NORM>(a, eps, G, s);
    else launch_decode_cfg<MT, 8, 1, EPI, NORM>(a, eps, G, s);
}

template <int EPI, bool NORM>
static void dispatch_decode(const GemmArgs& a, float eps, int G, int kg, hipStream_t s) {
    if (a.M <= 32) launch_decode<2, EPI, NORM>(a, eps, G, kg, s);
    else if (a.M <= 48) launch_decode<3, EPI, NORM>(a, eps, G, kg, s);
    else launch_decode<4, EPI, NORM>(a, eps, G, kg, s);
}

// Decode-step projection for 17..64 rows: C = epi(rstd?(A) * (A · W^T) + bias) over pack_weight() images with A (and optionally
// C / R) in the fragment-packed activation layout.  n_blocks: grid size of the stream-K partition (0 = one block per CU, 256);
// workspace from padt_gemm_decode_workspace(N, K, n_blocks), zero before the first call (the kernel leaves its tickets zero).
extern "C" int padt_gemm_decode_bf16(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* bias, void* C,
                                     long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                                     int n_blocks, void* workspace, int c_packed) {
    if (M <= 0 || N <= 0) return 0;
    if (M <= 16 || M > 64 || K <= 0 || (K & 31) || K > Kp || (Kp & 31) || (N & 15) || (lda & 7) || ((uintptr_t)A & 15) || ((uintptr_t)Wp & 15)) {
        padt_set_error("padt_gemm_decode_bf16: 16 < M <= 64, K % 32 == 0, K <= Kp, Kp % 32 == 0, N % 16 == 0, 16-byte aligned packed A / Wp required");
        return -1;
    }
    const bool norm = norm_eps >= 0.f;
    if ((ldc & 3) || ((uintptr_t)C & 15) || ((uintptr_t)bias & 7) || (epilogue != EPI_NONE && epilogue != EPI_RESID && epilogue != EPI_SWIGLU) ||
        (epilogue == EPI_SWIGLU && (N & 31)) || (epilogue == EPI_RESID && (R == nullptr || (ldr & 3) || ((uintptr_t)R & 7) || norm)) ||
        (c_packed && ((ldc & 7) || (R != nullptr && ldr != ldc))) || n_blocks < 0 || n_blocks > 4096 || workspace == nullptr) {
        padt_set_error("padt_gemm_decode_bf16: bad C/ldc/bias/epilogue (0, 2 without norm, 3 with N % 32 == 0), n_blocks in [0, 4096], workspace required");
        return -1;
    }
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)Wp, Kp, (const bf16_t*)bias, C, ldc, (const bf16_t*)R, ldr, (int)M, (int)N, (int)K};
    a.a_pack = 1;
    a.c_pack = c_packed ? 1 : 0;
    const int nt = (epilogue == EPI_SWIGLU) ? 2 : 1;
    const long tiles = (N + 64 * nt - 1) / (64 * nt);
    const long nks = K >> 5;
    static const int env_kg = getenv("PADT_DEC_KG") ? atoi(getenv("PADT_DEC_KG")) : 0;             // tuning knobs
    static const int env_g = getenv("PADT_DEC_G") ? atoi(getenv("PADT_DEC_G")) : 256;              // blocks chip-wide (one per CU)
    const int kg = env_kg ? env_kg : (nt == 2 ? 2 : 4);
    long G = n_blocks > 0 ? n_blocks : env_g;
    if (G > tiles * nks) G = tiles * nks;
    a.dec_tiles = (int)tiles;
    a.dec_slots = decode_slots(tiles, nks, G);
    a.ticket = reinterpret_cast<int*>(workspace);
    a.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + decode_ticket_bytes(N));
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SWIGLU) { if (norm) dispatch_decode<EPI_SWIGLU, true>(a, norm_eps, (int)G, kg, s); else dispatch_decode<EPI_SWIGLU, false>(a, 0.f, (int)G, kg, s); }
    else if (epilogue == EPI_RESID) dispatch_decode<EPI_RESID, false>(a, 0.f, (int)G, kg, s);
    else { if (norm) dispatch_decode<EPI_NONE, true>(a, norm_eps, (int)G, kg, s); else dispatch_decode<EPI_NONE, false>(a, 0.f, (int)G, kg, s); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
