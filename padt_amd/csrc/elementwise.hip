// HBM-bound row kernels for gfx950: norms, rotary embeddings, gathers, KV-cache append, PaDT decoder glue.
// All bf16 traffic is 16-byte vectorised (8 elements per lane per access); reductions are wave64 xor-shuffles.
#include "common.h"
#include <cstdint>

extern "C" void padt_set_error(const char* msg);

namespace PADT_NS {

#define PADT_CHECK_LAUNCH(name)                                          \
    do {                                                                 \
        hipError_t e_ = hipGetLastError();                               \
        if (e_ != hipSuccess) { padt_set_error(hipGetErrorString(e_)); return -2; } \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// RMSNorm (HF Qwen2_5_VLRMSNorm, modeling_qwen2_5_vl.py:74-79; padt_decoder.py:71-74,143,156,170): one wave per row,
// fp32 accumulate, single rounding on the way out (act = 1 applies exact-erf GELU after the weight: the
// Linear→RMSNorm→GELU of mask_output_upscaling1, padt_decoder.py:168-172).  Optional fused "add" input: y = norm(x + a[row / a_div]) which
// implements padt_decoder.py:220  high = RMSNorm(repeat4(low) + high)  with a_div = 4.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const x16_t* __restrict__ x, long ldx, const x16_t* __restrict__ a,
                                                      long lda, int a_div, const x16_t* __restrict__ w,
                                                      x16_t* __restrict__ y, long ldy, int rows, int D, float eps,
                                                      int act) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const x16_t* xr = x + (long)row * ldx;
    const x16_t* ar = a ? a + (long)(row / a_div) * lda : nullptr;
    float ss = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
        if (ar) {
            float g[8];
            unpack8(*reinterpret_cast<const u32x4*>(ar + c), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += g[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    x16_t* yr = y + (long)row * ldy;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8], g[8], wv[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
        if (ar) {
            unpack8(*reinterpret_cast<const u32x4*>(ar + c), g);
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += g[i];
        }
        unpack8(*reinterpret_cast<const u32x4*>(w + c), wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f[i] = f[i] * rstd * wv[i];
            if (act == 1) f[i] = gelu_erf(f[i]);
        }
        *reinterpret_cast<u32x4*>(yr + c) = pack8(f);
    }
}

extern "C" int PADT_TWIN(padt_rmsnorm)(void* stream, const void* x, long ldx, const void* add, long ld_add, int add_div,
                            const void* w, void* y, long ldy, long rows, long D, float eps, int act) {
    if (rows <= 0) return 0;
    if ((D & 7) || (ldx & 7) || (ldy & 7) || (add && (ld_add & 7)) || add_div <= 0) {
        padt_set_error("padt_rmsnorm: D and strides must be multiples of 8");
        return -1;
    }
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const x16_t*)x, ldx,
                       (const x16_t*)add, ld_add, add_div, (const x16_t*)w, (x16_t*)y, ldy, (int)rows, (int)D, eps, act);
    PADT_CHECK_LAUNCH("rmsnorm");
    return 0;
}

// LayerNorm with mean centring (vis_norm, padt.py:121,188; eps 1e-5, weight + bias).
__global__ __launch_bounds__(256) void layernorm_kernel(const x16_t* __restrict__ x, long ldx, const x16_t* __restrict__ w,
                                                        const x16_t* __restrict__ b, x16_t* __restrict__ y, long ldy,
                                                        int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const x16_t* xr = x + (long)row * ldx;
    float s = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[i];
    }
    const float mean = wave_sum(s) / (float)D;
    float v = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = f[i] - mean; v += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
    x16_t* yr = y + (long)row * ldy;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8], wv[8], bv[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
        unpack8(*reinterpret_cast<const u32x4*>(w + c), wv);
        unpack8(*reinterpret_cast<const u32x4*>(b + c), bv);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * wv[i] + bv[i];
        *reinterpret_cast<u32x4*>(yr + c) = pack8(f);
    }
}

extern "C" int PADT_TWIN(padt_layernorm)(void* stream, const void* x, long ldx, const void* w, const void* b, void* y, long ldy,
                              long rows, long D, float eps) {
    if (rows <= 0) return 0;
    if ((D & 7) || (ldx & 7) || (ldy & 7)) { padt_set_error("padt_layernorm: D and strides must be multiples of 8"); return -1; }
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const x16_t*)x, ldx,
                       (const x16_t*)w, (const x16_t*)b, (x16_t*)y, ldy, (int)rows, (int)D, eps);
    PADT_CHECK_LAUNCH("layernorm");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Rotate-half rotary in place over `nh` consecutive heads of width D per token, fp32 math, cos/sin tables fp32 [T][ld_cs]
// (first D/2 columns used).  ViT q,k (HF apply_rotary_pos_emb_vision :160-171; q and k are adjacent in the fused qkv
// row so one launch covers both) and the PaDT decoder's image-side q or k (padt_decoder.py:38-51).
__global__ __launch_bounds__(256) void rope_half_kernel(x16_t* __restrict__ x, long ldx, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, long ld_cs, long T, int nh, int D) {
    const int half = D >> 1;
    const long total = T * nh * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % half);
        const long th = i / half;
        const int h = (int)(th % nh);
        const long t = th / nh;
        x16_t* p = x + t * ldx + (long)h * D;
        const float c = cs[t * ld_cs + d], s = sn[t * ld_cs + d];
        const float x1 = x2f(p[d]), x2 = x2f(p[d + half]);
        p[d] = f2x(rounded32(rope_lo(x1, x2, c, s)));
        p[d + half] = f2x(rounded32(rope_hi(x1, x2, c, s)));
    }
}

// 16-byte path (D/2 and every stride a multiple of 8): a lane rotates 8 (d, d + D/2) pairs of one (token, head).
__global__ __launch_bounds__(256) void rope_half_vec_kernel(x16_t* __restrict__ x, long ldx, const float* __restrict__ cs,
                                                            const float* __restrict__ sn, long ld_cs, long T, int nh, int D) {
    const int half = D >> 1, cph = half >> 3;                     // 8-wide chunks per half head
    const long total = T * nh * cph;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cph);
        const long th = i / cph;
        const int h = (int)(th % nh);
        const long t = th / nh;
        x16_t* p = x + t * ldx + (long)h * D + c * 8;
        const u32x4 r1 = *reinterpret_cast<const u32x4*>(p), r2 = *reinterpret_cast<const u32x4*>(p + half);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs + t * ld_cs + c * 8), c1 = *reinterpret_cast<const f32x4*>(cs + t * ld_cs + c * 8 + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sn + t * ld_cs + c * 8), s1 = *reinterpret_cast<const f32x4*>(sn + t * ld_cs + c * 8 + 4);
        float x1[8], x2[8], o1[8], o2[8];
        unpack8(r1, x1);
        unpack8(r2, x2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cc = e < 4 ? c0[e & 3] : c1[e & 3], ss = e < 4 ? s0[e & 3] : s1[e & 3];
            o1[e] = rope_lo(x1[e], x2[e], cc, ss);
            o2[e] = rope_hi(x1[e], x2[e], cc, ss);
        }
        *reinterpret_cast<u32x4*>(p) = pack8(o1);
        *reinterpret_cast<u32x4*>(p + half) = pack8(o2);
    }
}

extern "C" int PADT_TWIN(padt_rope_half)(void* stream, void* x, long ldx, const void* cos_t, const void* sin_t, long ld_cs, long T,
                              int n_heads, int head_dim) {
    if (T <= 0) return 0;
    if (head_dim & 1) { padt_set_error("padt_rope_half: head_dim must be even"); return -1; }
    if ((head_dim % 16) == 0 && (ldx % 8) == 0 && (ld_cs % 4) == 0 && ((uintptr_t)x % 16) == 0 &&
        ((uintptr_t)cos_t % 16) == 0 && ((uintptr_t)sin_t % 16) == 0) {
        const long tv = T * n_heads * (head_dim / 16);
        long bv = (tv + 255) / 256;
        if (bv > 32768) bv = 32768;
        hipLaunchKernelGGL(rope_half_vec_kernel, dim3((unsigned)bv), dim3(256), 0, (hipStream_t)stream, (x16_t*)x, ldx,
                           (const float*)cos_t, (const float*)sin_t, ld_cs, T, n_heads, head_dim);
        PADT_CHECK_LAUNCH("rope_half");
        return 0;
    }
    const long total = T * n_heads * (head_dim / 2);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(rope_half_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (x16_t*)x, ldx,
                       (const float*)cos_t, (const float*)sin_t, ld_cs, T, n_heads, head_dim);
    PADT_CHECK_LAUNCH("rope_half");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// rstd[row] = rsqrt(mean(x[row]^2) + eps): the statistics half of an RMSNorm whose weight is folded into the following
// projection (y = rstd[m] * (x @ (W·diag(g))^T)[m] + b) — the GEMM then reads x itself and scales its accumulator
// (padt_gemm_bf16 row_scale), so the normalised copy of x is never written or re-read.  One wave per row.
__global__ __launch_bounds__(256) void row_rstd_kernel(const x16_t* __restrict__ x, long ldx, float* __restrict__ out, int rows,
                                                       int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const x16_t* xr = x + (long)row * ldx;
    float ss = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
    }
    ss = wave_sum(ss);
    if (lane == 0) out[row] = rsqrtf(ss / (float)D + eps);
}

extern "C" int PADT_TWIN(padt_row_rstd)(void* stream, const void* x, long ldx, void* out_f32, long rows, long D, float eps) {
    if (rows <= 0) return 0;
    if ((D & 7) || (ldx & 7) || ((uintptr_t)x & 15)) { padt_set_error("padt_row_rstd: D, ldx multiples of 8, x 16-byte aligned"); return -1; }
    hipLaunchKernelGGL(row_rstd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const x16_t*)x, ldx,
                       (float*)out_f32, (int)rows, (int)D, eps);
    PADT_CHECK_LAUNCH("row_rstd");
    return 0;
}

#if !PADT_OP16_F16   // type-independent (or bf16-decoder-only): compiled once
// ---------------------------------------------------------------------------------------------------------------------
// Row-major [M][K] <-> 16-row fragment-packed activation layout (padt_hip.h), 16-byte chunks.  Once per decode step on each
// side of the layer loop (embedding output in, final hidden state out); inside the loop the projections read and write the
// packed form directly.
__global__ __launch_bounds__(256) void pack_rows_kernel(const x16_t* __restrict__ src, long ld_src, x16_t* __restrict__ dst,
                                                        long ld_dst, int M, int chunks, int to_packed) {
    const long total = (long)M * chunks;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), c = (int)(i % chunks);
        const long rm = (long)m * (to_packed ? ld_src : ld_dst) + c * 8;
        const long pk = (long)(m >> 4) * 16 * (to_packed ? ld_dst : ld_src) + ((long)c * 16 + (m & 15)) * 8;
        if (to_packed) *reinterpret_cast<u32x4*>(dst + pk) = *reinterpret_cast<const u32x4*>(src + rm);
        else *reinterpret_cast<u32x4*>(dst + rm) = *reinterpret_cast<const u32x4*>(src + pk);
    }
}

extern "C" int padt_pack_rows(void* stream, const void* src, long ld_src, void* dst, long ld_dst, long M, long K, int to_packed) {
    if (M <= 0 || K <= 0) return 0;
    if ((K & 7) || (ld_src & 7) || (ld_dst & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) {
        padt_set_error("padt_pack_rows: K and strides must be multiples of 8, pointers 16-byte aligned");
        return -1;
    }
    const long total = M * (K / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const x16_t*)src, ld_src,
                       (x16_t*)dst, ld_dst, (int)M, (int)(K / 8), to_packed);
    PADT_CHECK_LAUNCH("pack_rows");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// dst[i] = src[idx[i]]  (16-byte vectors).  ViT window permutation (padt.py:70-75), un-permutation of the merger output
// (padt.py:103-104), per-object replication of image memory (padt.py:365-373).
__global__ __launch_bounds__(256) void gather_rows_kernel(const x16_t* __restrict__ src, long ld_src,
                                                          const int* __restrict__ idx, x16_t* __restrict__ dst,
                                                          long ld_dst, long n, int vec_per_row) {
    const long total = n * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int c = (int)(i % vec_per_row) * 8;
        *reinterpret_cast<u32x4*>(dst + r * ld_dst + c) = *reinterpret_cast<const u32x4*>(src + (long)idx[r] * ld_src + c);
    }
}

extern "C" int padt_gather_rows(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst,
                                long n, long D) {
    if (n <= 0) return 0;
    if ((D & 7) || (ld_src & 7) || (ld_dst & 7)) { padt_set_error("padt_gather_rows: D and strides must be multiples of 8"); return -1; }
    const long total = n * (D / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const x16_t*)src,
                       ld_src, idx, (x16_t*)dst, ld_dst, n, (int)(D / 8));
    PADT_CHECK_LAUNCH("gather_rows");
    return 0;
}

// same for fp32 rows (cos/sin tables replicated per object in vl_decode)
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ src, long ld_src,
                                                              const int* __restrict__ idx, float* __restrict__ dst,
                                                              long ld_dst, long n, int D) {
    const long total = n * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D;
        const int c = (int)(i % D);
        dst[r * ld_dst + c] = src[(long)idx[r] * ld_src + c];
    }
}

extern "C" int padt_gather_rows_f32(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst,
                                    long n, long D) {
    if (n <= 0) return 0;
    long blocks = (n * D + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, ld_src, idx, (float*)dst, ld_dst, n, (int)D);
    PADT_CHECK_LAUNCH("gather_rows_f32");
    return 0;
}

// dst[idx[i]] = src[i] for fp32 rows: the decode step's new K | V row of every sample into its slot of the fp32 KV cache (reference-precision
// LLM, padt_amd/reference.py; HF:641-689 cache update).  Indices must be distinct.
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float* __restrict__ src, long ld_src, const int* __restrict__ idx,
                                                               float* __restrict__ dst, long ld_dst, long n, int vec_per_row) {
    const long total = n * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int c = (int)(i % vec_per_row) * 4;
        *reinterpret_cast<f32x4*>(dst + (long)idx[r] * ld_dst + c) = *reinterpret_cast<const f32x4*>(src + r * ld_src + c);
    }
}

extern "C" int padt_scatter_rows_f32(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst, long n, long D) {
    if (n <= 0) return 0;
    if ((D & 3) || (ld_src & 3) || (ld_dst & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15) || idx == nullptr) {
        padt_set_error("padt_scatter_rows_f32: D and strides multiples of 4, 16-byte aligned rows");
        return -1;
    }
    long blocks = (n * (D / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, idx, (float*)dst,
                       ld_dst, n, (int)(D / 4));
    PADT_CHECK_LAUNCH("scatter_rows_f32");
    return 0;
}

#endif

// y = a + b[row % b_rows]   (decoder: key/query + positional query, padt_decoder.py:30-31; b_rows == rows → plain add)
__global__ __launch_bounds__(256) void add_rows_kernel(const x16_t* __restrict__ a, long lda, const x16_t* __restrict__ b,
                                                       long ldb, long b_rows, x16_t* __restrict__ y, long ldy, long n,
                                                       int vec_per_row) {
    const long total = n * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int c = (int)(i % vec_per_row) * 8;
        float f[8], g[8];
        unpack8(*reinterpret_cast<const u32x4*>(a + r * lda + c), f);
        unpack8(*reinterpret_cast<const u32x4*>(b + (r % b_rows) * ldb + c), g);
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] += g[k];
        *reinterpret_cast<u32x4*>(y + r * ldy + c) = pack8(f);
    }
}

extern "C" int PADT_TWIN(padt_add_rows)(void* stream, const void* a, long lda, const void* b, long ldb, long b_rows, void* y,
                             long ldy, long n, long D) {
    if (n <= 0) return 0;
    if ((D & 7) || (lda & 7) || (ldb & 7) || (ldy & 7) || b_rows <= 0) { padt_set_error("padt_add_rows: bad arguments"); return -1; }
    const long total = n * (D / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const x16_t*)a, lda,
                       (const x16_t*)b, ldb, b_rows, (x16_t*)y, ldy, n, (int)(D / 8));
    PADT_CHECK_LAUNCH("add_rows");
    return 0;
}

// fp32 → bf16 cast (pixel_values.type(self.visual.dtype), padt.py:184) with optional zero padding of the row tail.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, long ldx, x16_t* __restrict__ y,
                                                            long ldy, long rows, int D, int D_pad, float scale) {
    const long total = rows * D_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / D_pad;
        const int c = (int)(i % D_pad);
        y[r * ldy + c] = c < D ? f2x(x[r * ldx + c] * scale) : (x16_t)0;
    }
}

extern "C" int PADT_SYM(padt_cast_f32_, )(void* stream, const void* x, long ldx, void* y, long ldy, long rows, long D, long D_pad, float scale) {
    if (rows <= 0) return 0;
    long blocks = (rows * D_pad + 255) / 256;
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       ldx, (x16_t*)y, ldy, rows, (int)D, (int)D_pad, scale);
    PADT_CHECK_LAUNCH("cast_f32_bf16");
    return 0;
}

// bf16 → fp32 (16-byte loads): the token embeddings entering the fp32 residual stream of the LLM (padt_gemm_resid32).
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const x16_t* __restrict__ x, long ldx, float* __restrict__ y, long ldy, long rows, int vec) {
    const long total = rows * vec;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec;
        const int c = (int)(i % vec) * 8;
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(x + r * ldx + c), f);
        float* yp = y + r * ldy + c;
        *reinterpret_cast<f32x4*>(yp) = f32x4{f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(yp + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
}

extern "C" int PADT_SYM(padt_cast_, _f32)(void* stream, const void* x, long ldx, void* y, long ldy, long rows, long D) {
    if (rows <= 0) return 0;
    if ((D & 7) || (ldx & 7) || (ldy & 3) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) {
        padt_set_error("padt_cast_bf16_f32: D, ldx multiples of 8, ldy of 4, 16-byte aligned pointers");
        return -1;
    }
    long blocks = (rows * (D / 8) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const x16_t*)x, ldx, (float*)y, ldy,
                       rows, (int)(D / 8));
    PADT_CHECK_LAUNCH("cast_bf16_f32");
    return 0;
}

// RMSNorm of fp32 rows → bf16 (the norms that read the fp32 residual stream: ViT merger ln_q HF:141-148, LLM final norm HF:867): one wave per
// row, statistics and scaling in fp32, ONE rounding on the way out.
__global__ __launch_bounds__(256) void rmsnorm_f32_kernel(const float* __restrict__ x, long ldx, const x16_t* __restrict__ w, x16_t* __restrict__ y,
                                                          long ldy, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * ldx;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    x16_t* yr = y + (long)row * ldy;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        float wv[4];
        unpack4x(*reinterpret_cast<const u32x2*>(w + c), wv);
        // HF:74-79 rounds the normalised value to the input dtype before the weight multiply; with an fp32 stream that rounding is fp32
        *reinterpret_cast<u32x2*>(yr + c) = u32x2{pack2x(v[0] * rstd * wv[0], v[1] * rstd * wv[1]), pack2x(v[2] * rstd * wv[2], v[3] * rstd * wv[3])};
    }
}

extern "C" int PADT_TWIN(padt_rmsnorm_f32)(void* stream, const void* x_f32, long ldx, const void* w, void* y, long ldy, long rows, long D, float eps) {
    if (rows <= 0) return 0;
    if ((D & 3) || (ldx & 3) || (ldy & 3) || ((uintptr_t)x_f32 & 15) || ((uintptr_t)y & 7) || ((uintptr_t)w & 7)) {
        padt_set_error("padt_rmsnorm_f32: D and strides multiples of 4, x 16-byte / y, w 8-byte aligned");
        return -1;
    }
    hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)x_f32, ldx,
                       (const x16_t*)w, (x16_t*)y, ldy, (int)rows, (int)D, eps);
    PADT_CHECK_LAUNCH("rmsnorm_f32");
    return 0;
}

#if !PADT_OP16_F16   // type-independent (or bf16-decoder-only): compiled once
// in-place fp32 sigmoid (bbox head's nn.Sigmoid, padt_decoder.py:164)
__global__ void sigmoid_f32_kernel(float* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        x[i] = 1.0f / (1.0f + expf(-x[i]));
}

extern "C" int padt_sigmoid_f32(void* stream, void* x, long n) {
    if (n <= 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sigmoid_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float*)x, n);
    PADT_CHECK_LAUNCH("sigmoid_f32");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Finite check of a result tensor (round 5: the fp16-operand safety net).  flags[row / rows_per_flag] |= 1 when any element of the row
// is +-inf or NaN.  kind 0: fp32, 1: bf16, 2: fp16.  An fp16-operand overflow anywhere upstream (SwiGLU hidden, q / k / v, attention
// output, merger hidden, a stream beyond 2^20) reaches the rows checked here as inf / NaN (common.h rope_fin for the one case that
// could otherwise vanish), so the product path never returns a silently wrong number: modeling.py raises or re-runs the batch on the
// bf16 instantiation.  Reads the tensor once (16-byte vectors); the atomic only fires on a bad element.
__global__ __launch_bounds__(256) void check_finite_kernel(const unsigned* __restrict__ x, long ld_bytes, long rows, int vec_per_row, int kind,
                                                           int* __restrict__ flags, long rows_per_flag) {
    const long total = rows * vec_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vec_per_row;
        const int c = (int)(i % vec_per_row);
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(x) + r * ld_bytes + (long)c * 16);
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w = v[j];
            if (kind == 0) bad |= (w & 0x7F800000u) == 0x7F800000u;
            else if (kind == 1) bad |= ((w & 0x7F80u) == 0x7F80u) | ((w & 0x7F800000u) == 0x7F800000u);
            else bad |= ((w & 0x7C00u) == 0x7C00u) | ((w & 0x7C000000u) == 0x7C000000u);
        }
        if (bad) atomicOr(flags + r / rows_per_flag, 1);
    }
}

extern "C" int padt_check_finite(void* stream, const void* x, long ldx, long rows, long cols, int kind, int* flags, long rows_per_flag) {
    if (rows <= 0 || cols <= 0) return 0;
    const int es = kind == 0 ? 4 : 2;
    if (kind < 0 || kind > 2 || flags == nullptr || rows_per_flag <= 0 || ((cols * es) & 15) || ((ldx * es) & 15) || ((uintptr_t)x & 15)) {
        padt_set_error("padt_check_finite: kind 0 (fp32) / 1 (bf16) / 2 (fp16), 16-byte aligned rows of whole 16-byte vectors, flags and rows_per_flag >= 1 required");
        return -1;
    }
    const int vpr = (int)(cols * es / 16);
    long blocks = (rows * vpr + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)x, ldx * es, rows, vpr, kind,
                       flags, rows_per_flag);
    PADT_CHECK_LAUNCH("check_finite");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// VRT embedding: inputs_embeds[t] = image_embeds[img_index[t]] if img_index[t] >= 0 else [E ‖ proto][ids[t]]
// (padt.py:193-219 prefill, 226-229 decode) — the table is never concatenated: two base pointers.
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long* __restrict__ ids, const int* __restrict__ img_index,
                                                           const x16_t* __restrict__ E, const x16_t* __restrict__ proto,
                                                           const x16_t* __restrict__ image_embeds, x16_t* __restrict__ out,
                                                           long T, int V, int n_proto, int vec_per_row, int* __restrict__ err) {
    const long total = T * vec_per_row;
    const int D = vec_per_row * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long t = i / vec_per_row;
        const int c = (int)(i % vec_per_row) * 8;
        const long id = ids[t];
        const int ii = img_index ? img_index[t] : -1;
        const x16_t* src;
        if (ii >= 0) src = image_embeds + (long)ii * D;
        else if (id >= 0 && id < V) src = E + id * D;
        else if (id >= V && id < (long)V + n_proto) src = proto + (id - V) * D;
        else { if (err) atomicExch(err, 1); src = E; }            // assert input_ids.max() < table rows (padt.py:203)
        *reinterpret_cast<u32x4*>(out + t * D + c) = *reinterpret_cast<const u32x4*>(src + c);
    }
}

extern "C" int padt_embed_tokens(void* stream, const long* ids, const int* img_index, const void* embed_table,
                                 const void* proto, const void* image_embeds, void* out, long T, long vocab, long n_proto,
                                 long D, int* err_flag) {
    if (T <= 0) return 0;
    if (D & 7) { padt_set_error("padt_embed_tokens: D must be a multiple of 8"); return -1; }
    long blocks = (T * (D / 8) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ids, img_index,
                       (const x16_t*)embed_table, (const x16_t*)proto, (const x16_t*)image_embeds, (x16_t*)out, T,
                       (int)vocab, (int)n_proto, (int)(D / 8), err_flag);
    PADT_CHECK_LAUNCH("embed_tokens");
    return 0;
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// LLM q/k/v post-processing for T tokens: mRoPE on q and k (HF apply_multimodal_rotary_pos_emb :557-599, sections
// [s0,s1,s2] over head_dim/2, rotate-half pairs), write q to q_out, roped k to the row-major K cache and (prefill) to a
// packed k buffer, v to the TRANSPOSED V cache.  qkv row = [Hq*D | Hkv*D | Hkv*D] (bias already added by the GEMM).
struct QkvPostArgs {
    const x16_t* qkv; long ld;
    const int* pos;            // [3][T]  (t, h, w) rope positions
    const int* sample;         // [T] sample index (cache row); null → token index
    const int* slot;           // [T] cache slot; null → read from lens[sample] (decode)
    const int* lens;           // [B] current lengths (decode: slot = lens[b])
    const float* inv_freq;     // [D/2]
    x16_t* q_out; long ld_q;
    x16_t* k_pack; long ld_kp; // may be null
    x16_t* kc; x16_t* vtc;   // caches
    int T, Hq, Hkv, D, S_max, sec0, sec1;
    int cache_packed;          // 1: fragment-packed cache images (csrc/attention.hip decode_attn_rope_packed_kernel), 0: row-major K / V^T
};

// element offsets inside one (sample, kv head) cache image; D % 32 == 0, S_max % 32 == 0 for the packed forms
PADT_DEV long kc_offset(int slot, int d, int D, int packed) {           // K: row-major [S][D] or [S/16][D/32][64 lanes][8]
    return packed ? ((((long)(slot >> 4) * (D >> 5) + (d >> 5)) * 64) + ((d >> 3) & 3) * 16 + (slot & 15)) * 8 + (d & 7) : (long)slot * D + d;
}
PADT_DEV long vtc_offset(int slot, int d, int S_max, int packed) {      // V^T: row-major [D][S] or [D/16][S/32][64 lanes][8]
    return packed ? (((long)(d >> 4) * (S_max >> 5) + (slot >> 5)) * 64 + ((slot & 15) >> 2) * 16 + (d & 15)) * 8 + ((slot >> 4) & 1) * 4 + (slot & 3)
                  : (long)d * S_max + slot;
}

__global__ __launch_bounds__(256) void llm_qkv_post_kernel(QkvPostArgs p) {
    __shared__ float cs[2][128];                                  // cos / sin of this token's D/2 angles (D <= 256)
    const int t = blockIdx.x;
    const int half = p.D >> 1;
    const int b = p.sample ? p.sample[t] : t;
    const int slot = p.slot ? p.slot[t] : p.lens[b];
    const x16_t* row = p.qkv + (long)t * p.ld;
    // the angle depends on (token, d) only: computed once per token, not once per head
    for (int d = threadIdx.x; d < half; d += blockDim.x) {
        const int axis = d < p.sec0 ? 0 : (d < p.sec0 + p.sec1 ? 1 : 2);
        const float ang = (float)p.pos[(long)axis * p.T + t] * p.inv_freq[d];
        cs[0][d] = cosf(ang);
        cs[1][d] = sinf(ang);
    }
    __syncthreads();
    if ((half & 7) == 0 && (p.ld & 7) == 0 && (p.ld_q & 7) == 0 && (p.ld_kp & 7) == 0) {
        // 16-byte path: item = (head, 8-wide chunk of the first half); its partner chunk sits D/2 further
        const int cph = half >> 3;
        const int items = (p.Hq + p.Hkv) * cph;
        for (int i = threadIdx.x; i < items; i += blockDim.x) {
            const int h = i / cph, d = (i % cph) * 8;
            const x16_t* x = row + (long)h * p.D + d;
            float x1[8], x2[8], o1[8], o2[8];
            unpack8(*reinterpret_cast<const u32x4*>(x), x1);
            unpack8(*reinterpret_cast<const u32x4*>(x + half), x2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = cs[0][d + e], sn = cs[1][d + e];
                o1[e] = rope_lo(x1[e], x2[e], c, sn);
                o2[e] = rope_hi(x1[e], x2[e], c, sn);
            }
            const u32x4 r1 = pack8(o1), r2 = pack8(o2);
            if (h < p.Hq) {
                x16_t* q = p.q_out + (long)t * p.ld_q + (long)h * p.D + d;
                *reinterpret_cast<u32x4*>(q) = r1;
                *reinterpret_cast<u32x4*>(q + half) = r2;
            } else {
                const int g = h - p.Hq;
                x16_t* kc = p.kc + ((long)b * p.Hkv + g) * p.S_max * p.D;         // an 8-wide chunk (d % 8 == 0) is contiguous in both images
                *reinterpret_cast<u32x4*>(kc + kc_offset(slot, d, p.D, p.cache_packed)) = r1;
                *reinterpret_cast<u32x4*>(kc + kc_offset(slot, d + half, p.D, p.cache_packed)) = r2;
                if (p.k_pack) {
                    x16_t* kp = p.k_pack + (long)t * p.ld_kp + (long)g * p.D + d;
                    *reinterpret_cast<u32x4*>(kp) = r1;
                    *reinterpret_cast<u32x4*>(kp + half) = r2;
                }
            }
        }
    } else {
        const int nqk = (p.Hq + p.Hkv) * half;
        for (int i = threadIdx.x; i < nqk; i += blockDim.x) {
            const int h = i / half, d = i % half;
            const float c = cs[0][d], s = cs[1][d];
            const x16_t* x = row + (long)h * p.D;
            const float x1 = x2f(x[d]), x2 = x2f(x[d + half]);
            const x16_t o1 = f2x(rounded32(rope_lo(x1, x2, c, s))), o2 = f2x(rounded32(rope_hi(x1, x2, c, s)));
            if (h < p.Hq) {
                x16_t* q = p.q_out + (long)t * p.ld_q + (long)h * p.D;
                q[d] = o1; q[d + half] = o2;
            } else {
                const int g = h - p.Hq;
                x16_t* kc = p.kc + ((long)b * p.Hkv + g) * p.S_max * p.D;
                kc[kc_offset(slot, d, p.D, p.cache_packed)] = o1; kc[kc_offset(slot, d + half, p.D, p.cache_packed)] = o2;
                if (p.k_pack) {
                    x16_t* kp = p.k_pack + (long)t * p.ld_kp + (long)g * p.D;
                    kp[d] = o1; kp[d + half] = o2;
                }
            }
        }
    }
    const x16_t* v = row + (long)(p.Hq + p.Hkv) * p.D;
    for (int i = threadIdx.x; i < p.Hkv * p.D; i += blockDim.x) {
        const int g = i / p.D, d = i % p.D;
        p.vtc[((long)b * p.Hkv + g) * p.D * p.S_max + vtc_offset(slot, d, p.S_max, p.cache_packed)] = v[i];
    }
}

extern "C" int PADT_TWIN(padt_llm_qkv_post)(void* stream, const void* qkv, long ld_qkv, const int* pos3, const int* sample,
                                 const int* slot, const int* lens, const void* inv_freq, void* q_out, long ld_q,
                                 void* k_pack, long ld_kp, void* k_cache, void* vt_cache, long T, int n_heads,
                                 int n_kv_heads, int head_dim, int s_max, int sec0, int sec1, int cache_packed) {
    if (T <= 0) return 0;
    if (!slot && !lens) { padt_set_error("padt_llm_qkv_post: need slot[] or lens[]"); return -1; }
    if (head_dim > 256 || (head_dim & 1)) { padt_set_error("padt_llm_qkv_post: head_dim must be even and <= 256"); return -1; }
    if (cache_packed && ((head_dim & 31) || (s_max & 31))) { padt_set_error("padt_llm_qkv_post: packed caches need head_dim % 32 == 0 and s_max % 32 == 0"); return -1; }
    QkvPostArgs a{(const x16_t*)qkv, ld_qkv, pos3, sample, slot, lens, (const float*)inv_freq, (x16_t*)q_out, ld_q,
                  (x16_t*)k_pack, ld_kp, (x16_t*)k_cache, (x16_t*)vt_cache, (int)T, n_heads, n_kv_heads, head_dim,
                  s_max, sec0, sec1, cache_packed ? 1 : 0};
    hipLaunchKernelGGL(llm_qkv_post_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, a);
    PADT_CHECK_LAUNCH("llm_qkv_post");
    return 0;
}

#if !PADT_OP16_F16   // type-independent (or bf16-decoder-only): compiled once
// ---------------------------------------------------------------------------------------------------------------------
// PaDT mask head tail (padt_decoder.py:241-274): e2[(n,a,b)][(c,d,:)] · mask_tok[obj(n)] → masks[obj][4*row+2a+c][4*col+2b+d]
// e2: [4*Np][4*dm] bf16 (row = patch*4 + a*2 + b, col = (c*2+d)*dm + k), tok: [n_obj][dm] bf16.
__global__ __launch_bounds__(256) void mask_scatter_kernel(const x16_t* __restrict__ e2, long ld_e2,
                                                           const x16_t* __restrict__ tok, long ld_tok,
                                                           const int* __restrict__ cu_patch, const int* __restrict__ obj_w,
                                                           float* __restrict__ masks, int n_obj, int Hm4, int Wm4, int dm) {
    // one thread per output logit: index = ((patch*4 + ab)*4 + cd)
    const long total = (long)cu_patch[n_obj] * 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cd = (int)(i & 3), ab = (int)((i >> 2) & 3);
        const long patch = i >> 4;
        int lo = 0, hi = n_obj;                                    // obj = last o with cu_patch[o] <= patch
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cu_patch[mid] <= patch) lo = mid; else hi = mid; }
        const int obj = lo;
        const int pin = (int)(patch - cu_patch[obj]);
        const int W = obj_w[obj];
        const int prow = pin / W, pcol = pin % W;
        const x16_t* e = e2 + (patch * 4 + ab) * ld_e2 + (long)cd * dm;
        const x16_t* tk = tok + (long)obj * ld_tok;
        float acc = 0.f;
        for (int k = 0; k < dm; ++k) acc += x2f(e[k]) * x2f(tk[k]);
        const int a = ab >> 1, bb = ab & 1, c = cd >> 1, d = cd & 1;
        masks[((long)obj * Hm4 + prow * 4 + a * 2 + c) * Wm4 + pcol * 4 + bb * 2 + d] = acc;
    }
}

extern "C" int padt_mask_scatter(void* stream, const void* e2, long ld_e2, const void* mask_tok, long ld_tok,
                                 const int* cu_patch, const int* obj_w, void* masks_f32, int n_obj, long total_patches,
                                 int Hm4, int Wm4, int dm) {
    if (n_obj <= 0 || total_patches <= 0) return 0;
    long blocks = (total_patches * 16 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(mask_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const x16_t*)e2,
                       ld_e2, (const x16_t*)mask_tok, ld_tok, cu_patch, obj_w, (float*)masks_f32, n_obj, Hm4, Wm4, dm);
    PADT_CHECK_LAUNCH("mask_scatter");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Caller-side mask post-processing (eval/evaluation_scripts/utils.py:262, eval/test_demo.py:153):
//     F.interpolate(mask[None, None, :4H, :4W], size=(h, w), mode='bilinear')[0, 0].sigmoid() > 0.5
// fused into one pass: fp32 bilinear resize (align_corners=False; source index = max(scale*(dst+0.5)-0.5, 0), scale =
// in/out as float, the same expression order as torch's upsample_bilinear2d) of each object's valid logit region to its
// image size, fp32 sigmoid, threshold, one byte per pixel.  Optionally also stores the up-sampled logits (tests).
struct MaskPostArgs {
    const float* masks; long ld_obj, ld_row;
    const int* src_h; const int* src_w;
    const int* dst_h; const int* dst_w;
    unsigned char* out; long out_ld_obj, out_ld_row;
    float* up; long up_ld_obj, up_ld_row;
};

__global__ __launch_bounds__(256) void mask_upsample_binarize_kernel(MaskPostArgs p) {
    const int o = blockIdx.z;
    const int H = p.dst_h[o], W = p.dst_w[o], hs = p.src_h[o], ws = p.src_w[o];
    const int y = blockIdx.y;
    if (y >= H) return;
    const float sy = (float)hs / (float)H, sx = (float)ws / (float)W;
    float fy = sy * ((float)y + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < hs - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    const float* r0 = p.masks + (long)o * p.ld_obj + (long)y0 * p.ld_row;
    const float* r1 = p.masks + (long)o * p.ld_obj + (long)y1 * p.ld_row;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < W; x += gridDim.x * blockDim.x) {
        float fx = sx * ((float)x + 0.5f) - 0.5f;
        fx = fx < 0.f ? 0.f : fx;
        const int x0 = (int)fx;
        const int x1 = x0 + (x0 < ws - 1 ? 1 : 0);
        const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
        const float v = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
        const float sg = 1.f / (1.f + expf(-v));
        p.out[(long)o * p.out_ld_obj + (long)y * p.out_ld_row + x] = sg > 0.5f ? 1 : 0;
        if (p.up) p.up[(long)o * p.up_ld_obj + (long)y * p.up_ld_row + x] = v;
    }
}

extern "C" int padt_mask_upsample_binarize(void* stream, const void* masks_f32, long ld_obj, long ld_row, const int* src_h,
                                           const int* src_w, const int* dst_h, const int* dst_w, void* out_u8,
                                           long out_ld_obj, long out_ld_row, void* up_f32, long up_ld_obj, long up_ld_row,
                                           int n_obj, int max_dst_h, int max_dst_w) {
    if (n_obj <= 0 || max_dst_h <= 0 || max_dst_w <= 0) return 0;
    if (n_obj > 65535 || max_dst_h > 65535) { padt_set_error("padt_mask_upsample_binarize: n_obj and max_dst_h must be <= 65535"); return -1; }
    MaskPostArgs a{(const float*)masks_f32, ld_obj, ld_row, src_h, src_w, dst_h, dst_w, (unsigned char*)out_u8, out_ld_obj,
                   out_ld_row, (float*)up_f32, up_ld_obj, up_ld_row};
    const int bx = (max_dst_w + 255) / 256;
    hipLaunchKernelGGL(mask_upsample_binarize_kernel, dim3(bx, max_dst_h, n_obj), dim3(256), 0, (hipStream_t)stream, a);
    PADT_CHECK_LAUNCH("mask_upsample_binarize");
    return 0;
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// Image front-end tail (HF Qwen2-VL image processor: rescale → normalize → patchify,
// image_processing_pil_qwen2_vl.py _preprocess / patchify): uint8 (H, W, 3) → pixel_values rows [C=3][T=2][14][14] in
// (h/2, w/2, 2, 2) block-major patch order.  A byte has 256 values per channel, so rescale+normalize is a 3 x 256 fp32
// table built on the host with the processor's exact arithmetic — the kernel is a gather + layout change (bit-exact fp32,
// or its round-to-nearest-even bf16).
__global__ __launch_bounds__(256) void patchify_normalize_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                                 const float* __restrict__ lut, void* __restrict__ out,
                                                                 long ld_out, int out_bf16, int patch, int merge, int temporal) {
    const int gw = W / patch;
    const int row_len = 3 * temporal * patch * patch;
    const int p = blockIdx.x;                                     // output row = patch in block-major order
    const int bw_n = gw / merge;
    const int mw = p % merge, mh = (p / merge) % merge, bw = (p / (merge * merge)) % bw_n, bh = p / (merge * merge * bw_n);
    const int y0 = (bh * merge + mh) * patch, x0 = (bw * merge + mw) * patch;
    for (int e = threadIdx.x; e < row_len; e += blockDim.x) {
        const int px = e % patch, py = (e / patch) % patch, c = e / (patch * patch * temporal);
        const unsigned char u = img[((long)(y0 + py) * W + (x0 + px)) * 3 + c];
        const float v = lut[c * 256 + u];
        if (out_bf16) reinterpret_cast<x16_t*>(out)[(long)p * ld_out + e] = f2x(v);
        else reinterpret_cast<float*>(out)[(long)p * ld_out + e] = v;
    }
}

extern "C" int PADT_TWIN(padt_patchify_normalize)(void* stream, const void* img_u8, int H, int W, const void* lut_f32, void* out,
                                       long ld_out, int out_bf16, int patch, int merge, int temporal) {
    if (H <= 0 || W <= 0) return 0;
    if (patch <= 0 || merge <= 0 || temporal <= 0 || H % (patch * merge) || W % (patch * merge)) {
        padt_set_error("padt_patchify_normalize: H and W must be multiples of patch*merge");
        return -1;
    }
    const int n_patch = (H / patch) * (W / patch);
    hipLaunchKernelGGL(patchify_normalize_kernel, dim3(n_patch), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)img_u8,
                       H, W, (const float*)lut_f32, out, ld_out, out_bf16, patch, merge, temporal);
    PADT_CHECK_LAUNCH("patchify_normalize");
    return 0;
}

#if !PADT_OP16_F16   // type-independent (or bf16-decoder-only): compiled once
// ---------------------------------------------------------------------------------------------------------------------
// Result record of the data-parallel exchange (pipeline.pack_results, SURVEY.md §8e: the decoded boxes / masks of one batch as ONE
// fixed-capacity record, so that one all-gather moves it).  32-bit words:
//   [n, cap, mask_hw, has_mask | sample_idx (cap) | valid_hw (2 cap) | boxes f32 (4 cap) | scores f32 (cap) | mask logits f32 (cap * mask_hw^2)]
// floats travel as their bit patterns; everything outside the n valid objects / the H x W mask window is zero.  One thread per word, no host
// round trip (the torch statement of this pack was a dozen small ops + three pageable H2D copies per batch).
struct PackArgs {
    int* out; long words;
    int n, cap, mask_hw, has_mask;
    const int* sample_idx; const long* valid_h; const long* valid_w;
    const float* boxes; long ld_box; const float* scores; long ld_score;
    const float* masks; long ld_obj, ld_row; int H, W;
};

__global__ __launch_bounds__(256) void pack_results_kernel(PackArgs p) {
    const long o_idx = 4, o_hw = o_idx + p.cap, o_box = o_hw + 2L * p.cap, o_sc = o_box + 4L * p.cap, o_mask = o_sc + p.cap;
    const long plane = (long)p.mask_hw * p.mask_hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.words; i += (long)gridDim.x * blockDim.x) {
        int v = 0;
        if (i < o_idx) v = i == 0 ? p.n : (i == 1 ? p.cap : (i == 2 ? p.mask_hw : p.has_mask));
        else if (i < o_hw) { const long k = i - o_idx; if (k < p.n) v = p.sample_idx[k]; }
        else if (i < o_box) { const long k = (i - o_hw) >> 1; if (k < p.n && p.has_mask) v = (int)(((i - o_hw) & 1) ? p.valid_w[k] : p.valid_h[k]); }
        else if (i < o_sc) { const long k = (i - o_box) >> 2; if (k < p.n) v = __builtin_bit_cast(int, p.boxes[k * p.ld_box + ((i - o_box) & 3)]); }
        else if (i < o_mask) { const long k = i - o_sc; if (k < p.n) v = __builtin_bit_cast(int, p.scores[k * p.ld_score]); }
        else if (p.has_mask) {
            const long k = (i - o_mask) / plane, r = (i - o_mask) % plane;
            const int y = (int)(r / p.mask_hw), x = (int)(r % p.mask_hw);
            if (k < p.n && y < p.H && x < p.W) v = __builtin_bit_cast(int, p.masks[k * p.ld_obj + (long)y * p.ld_row + x]);
        }
        p.out[i] = v;
    }
}

extern "C" int padt_pack_results(void* stream, void* out_i32, long words, int n, int cap, int mask_hw, const int* sample_idx, const long* valid_h,
                                 const long* valid_w, const void* boxes_f32, long ld_box, const void* scores_f32, long ld_score,
                                 const void* masks_f32, long ld_obj, long ld_row, int H, int W) {
    const int has_mask = (masks_f32 != nullptr && n > 0) ? 1 : 0;
    if (n < 0 || n > cap || words != 4 + 8L * cap + (long)cap * mask_hw * mask_hw || (has_mask && (H > mask_hw || W > mask_hw)) ||
        (n > 0 && (sample_idx == nullptr || boxes_f32 == nullptr || scores_f32 == nullptr)) || (has_mask && (valid_h == nullptr || valid_w == nullptr))) {
        padt_set_error("padt_pack_results: n <= cap, words = 4 + 8 cap + cap mask_hw^2, mask H, W <= mask_hw and non-null fields required");
        return -1;
    }
    PackArgs a{(int*)out_i32, words, n, cap, mask_hw, has_mask, sample_idx, valid_h, valid_w, (const float*)boxes_f32, ld_box,
               (const float*)scores_f32, ld_score, (const float*)masks_f32, ld_obj, ld_row, H, W};
    long blocks = (words + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    PADT_CHECK_LAUNCH("pack_results");
    return 0;
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// Per-row fp8 quantisation of activations for the fp8 x fp8 MFMA GEMM (padt_gemm_fp8): x8[row] = e4m3(x[row] / s), s = 2^ceil(log2(amax / 448))
// (a power of two: the division is exact, the only rounding is the e4m3 one), rs[row] = s, or s * rsqrt(mean(x^2) + eps) when the row also
// feeds a folded RMSNorm (norm_eps >= 0) — the GEMM scales its accumulator with rs[m] * weight_scale[n].  One wave per row, two passes (the
// second one re-reads the row from L2).
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const x16_t* __restrict__ x, long ldx, unsigned char* __restrict__ y, long ldy,
                                                             float* __restrict__ rs, int rows, int K, float norm_eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const x16_t* xr = x + (long)row * ldx;
    float amax = 0.f, ss = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { amax = fmaxf(amax, fabsf(f[i])); ss += f[i] * f[i]; }
    }
    amax = wave_max(amax);
    ss = wave_sum(ss);
    // a non-finite row (inf or NaN anywhere: fmaxf drops NaNs, the sum of squares does not) must not turn into a garbage scale and finite-looking
    // codes: its row scale becomes NaN, so the GEMM's accumulator row is NaN like the bf16 path's would be
    const bool finite = ss < INFINITY;
    if (!finite) amax = 0.f;
    int e = 0;
    (void)frexpf(amax * (1.0f / 448.0f), &e);                     // amax / 448 = m * 2^e, m in [0.5, 1)  →  2^e >= amax / 448
    float scale = (amax > 0.f) ? ldexpf(1.0f, e) : 1.0f;
    if (amax > 0.f && ldexpf(1.0f, e - 1) * 448.0f >= amax) scale = ldexpf(1.0f, e - 1);   // m == 0.5 exactly: the smaller power still covers amax
    const float inv = 1.0f / scale;
    unsigned char* yr = y + (long)row * ldy;
    for (int c = lane * 8; c < K; c += 512) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4*>(xr + c), f);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0] * inv, f[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2] * inv, f[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4] * inv, f[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6] * inv, f[7] * inv, hi, true);
        *reinterpret_cast<u32x2*>(yr + c) = u32x2{(unsigned)lo, (unsigned)hi};
    }
    if (lane == 0) rs[row] = !finite ? __builtin_nanf("") : ((norm_eps >= 0.f) ? scale * rsqrtf(ss / (float)K + norm_eps) : scale);
}

extern "C" int PADT_TWIN(padt_quant_rows_fp8)(void* stream, const void* x, long ldx, void* x8, long ld8, void* row_scale_f32, long rows, long K, float norm_eps) {
    if (rows <= 0) return 0;
    if ((K & 7) || (ldx & 7) || (ld8 & 7) || ((uintptr_t)x & 15) || ((uintptr_t)x8 & 7) || ld8 < K || rows > 0x7fffffffL || K > 0x7fffffffL) {
        padt_set_error("padt_quant_rows_fp8: K, ldx, ld8 multiples of 8, x 16-byte / x8 8-byte aligned, ld8 >= K, rows and K below 2^31");
        return -1;
    }
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const x16_t*)x, ldx,
                       (unsigned char*)x8, ld8, (float*)row_scale_f32, (int)rows, (int)K, norm_eps);
    PADT_CHECK_LAUNCH("quant_rows_fp8");
    return 0;
}

}  // namespace PADT_NS
