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

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_SWIGLU = 3 };

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    const bf16_t* bias;          // [N] or null
    void* C; long ldc;           // bf16 or f32
    const bf16_t* R; long ldr;   // residual (EPI_RESID)
    int M, N, K;
};

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[64];   // 256 B of zeros (K-tail source)

// ---------------------------------------------------------------------------------------------------------------------
// epilogue for one 16x16 fragment held "swapped": lane has row m, columns n..n+3 in v[0..3]
template <int EPI, bool OUT_F32>
PADT_DEV void store_frag(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M || n >= p.N) return;
    const bool full = (n + 3 < p.N);
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float x = v[r];
        if (p.bias && (n + r < p.N)) x += bf2f(p.bias[n + r]);
        if (EPI == EPI_GELU) x = gelu_erf(x);
        if (EPI == EPI_RESID && (n + r < p.N)) x += bf2f(p.R[(long)m * p.ldr + n + r]);
        o[r] = x;
    }
    if (OUT_F32) {
        float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
        if (full) *reinterpret_cast<f32x4*>(c) = f32x4{o[0], o[1], o[2], o[3]};
        else for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = o[r];
    } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n;
        if (full) *reinterpret_cast<u32x2*>(c) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        else for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = f2bf(o[r]);
    }
}

// SwiGLU pair: g/u fragments of weight rows [32q,32q+16) / [32q+16,32q+32) → output columns 16q + ...
PADT_DEV void store_swiglu(const GemmArgs& p, int m, int n_gate, f32x4 g, f32x4 u) {
    // n_gate = interleaved row index of the first gate element held by this lane (multiple of 4, inside a gate block)
    if (m >= p.M || n_gate >= p.N) return;
    const int blk = n_gate >> 5, in = n_gate & 15;
    const int no = blk * 16 + in;                       // output column
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float gv = g[r], uv = u[r];
        if (p.bias) { gv += bf2f(p.bias[n_gate + r]); uv += bf2f(p.bias[n_gate + 16 + r]); }
        o[r] = silu(gv) * uv;
    }
    bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + no;
    *reinterpret_cast<u32x2*>(c) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile kernel
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;            // 16 KiB per operand tile
constexpr int GEMM_LDS = 4 * TILE_BYTES;           // 2 buffers x (A, W)

PADT_DEV void stage_tile(const bf16_t* __restrict__ base, long ld, int row0, int nrows, int k0, int K,
                         char* lds_tile, int wave, int lane) {
    // 16 chunks of 1 KiB (8 rows x 128 B); wave w issues chunks 4w..4w+3.  LDS slot (r, s) holds global chunk s^(r&7).
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;
        const int r = c * 8 + (lane >> 3);
        const int s = lane & 7;
        const int j = s ^ (r & 7);
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

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmArgs p) {
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
    stage_tile(p.A, p.lda, m0, p.M, 0, p.K, smem, wave, lane);
    stage_tile(p.W, p.ldw, n0, p.N, 0, p.K, smem + TILE_BYTES, wave, lane);

    const int frow = lane & 15, fq = lane >> 4;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // tile t landed everywhere; everyone is done reading buf[(t+1)&1]
        const int cur = t & 1;
        if (t + 1 < nk) {
            char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
            stage_tile(p.A, p.lda, m0, p.M, (t + 1) * BK, p.K, nxt, wave, lane);
            stage_tile(p.W, p.ldw, n0, p.N, (t + 1) * BK, p.K, nxt + TILE_BYTES, wave, lane);
        }
        const char* a_t = smem + cur * 2 * TILE_BYTES;
        const char* w_t = a_t + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int j = kk * 4 + fq;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ra = wm * 64 + i * 16 + frow;
                af[i] = ld_frag(a_t + ra * 128 + ((j ^ (ra & 7)) << 4));
                const int rw = wn * 64 + i * 16 + frow;
                wf[i] = ld_frag(w_t + rw * 128 + ((j ^ (rw & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(wf[ni], af[mi], acc[mi][ni]);
        }
    }

    // epilogue: acc[mi][ni][r] = C[m0 + wm*64 + mi*16 + (lane&15)][n0 + wn*64 + ni*16 + (lane>>4)*4 + r]
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
// Skinny kernel: M <= 16*MT.  Block = 4 waves, owns NT*16 weight rows; wave w handles K-steps w, w+4, ...
template <int MT, int NT, int EPI, bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[3][NT * MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int nks = (p.K + 31) / 32;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bf16_t* wrow[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        int n = n0 + i * 16 + frow;
        n = n < p.N ? n : p.N - 1;
        wrow[i] = p.W + (long)n * p.ldw;
    }
    const bf16_t* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = j * 16 + frow;
        xok[j] = m < p.M;
        xrow[j] = p.A + (long)(xok[j] ? m : 0) * p.lda;
    }

    constexpr int U = 4;                                   // K-steps in flight per wave
    for (int ks0 = wave; ks0 < nks; ks0 += 4 * U) {
        bf16x8 wf[U][NT], xf[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = ks0 + u * 4;
            const int k = ks * 32 + fq * 8;
            const bool kok = (ks < nks) && (k < p.K);
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[u][i] = kok ? ld_frag(wrow[i] + k) : zero_frag();
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[u][j] = (kok && xok[j]) ? ld_frag(xrow[j] + k) : zero_frag();
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[u][i], xf[u][j], acc[i][j]);
    }

    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(&red[wave - 1][i * MT + j][lane][0]) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int w = 0; w < 3; ++w) acc[i][j] += *reinterpret_cast<f32x4*>(&red[w][i * MT + j][lane][0]);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = j * 16 + frow;
            if (EPI == EPI_SWIGLU) {
                store_swiglu(p, m, n0 + fq * 4, acc[0][j], acc[NT - 1][j]);
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i) store_frag<EPI, OUT_F32>(p, m, n0 + i * 16 + fq * 4, acc[i][j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
extern "C" void padt_set_error(const char* msg);

template <int EPI, bool F32>
static void launch_tile(const GemmArgs& a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<EPI, F32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        attr_done = true;
    }
    const int ntm = (a.M + BM - 1) / BM, ntn = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_tile_kernel<EPI, F32>), dim3(ntm * ntn), dim3(256), GEMM_LDS, s, a);
}

template <int MT, int EPI, bool F32>
static void launch_skinny(const GemmArgs& a, hipStream_t s) {
    constexpr int NT = (EPI == EPI_SWIGLU) ? 2 : 1;
    const int nb = (a.N + 16 * NT - 1) / (16 * NT);
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, EPI, F32>), dim3(nb), dim3(256), 0, s, a);
}

template <int EPI, bool F32>
static void dispatch_m(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 16) launch_skinny<1, EPI, F32>(a, s);
    else if (a.M <= 32) launch_skinny<2, EPI, F32>(a, s);
    else if (a.M <= 64) launch_skinny<4, EPI, F32>(a, s);
    else launch_tile<EPI, F32>(a, s);
}

extern "C" int padt_gemm_bf16(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                              long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 7) || (lda & 7) || (ldw & 7) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15)) {
        padt_set_error("padt_gemm_bf16: K, lda, ldw must be multiples of 8 and A, W 16-byte aligned");
        return -1;
    }
    const long out_n = (epilogue == EPI_SWIGLU) ? N / 2 : N;
    if ((ldc & 3) || ((uintptr_t)C & 15) || (epilogue == EPI_SWIGLU && ((N & 31) || out_f32)) ||
        (epilogue == EPI_RESID && R == nullptr) || ldc < out_n) {
        padt_set_error("padt_gemm_bf16: bad C/ldc/epilogue arguments (ldc % 4, C 16-byte aligned, SwiGLU needs N % 32 == 0 and bf16 out)");
        return -1;
    }
    if (epilogue < 0 || epilogue > 3) { padt_set_error("padt_gemm_bf16: unknown epilogue"); return -1; }
    GemmArgs a{(const bf16_t*)A, lda, (const bf16_t*)W, ldw, (const bf16_t*)bias, C, ldc, (const bf16_t*)R, ldr,
               (int)M, (int)N, (int)K};
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
