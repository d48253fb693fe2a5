// Device-side helpers shared by every kernel file.  gfx950 (MI355X, CDNA4) only: wave = 64 lanes,
// MFMA 16x16x32 fragments of a 16-bit operand type (bf16 or fp16, below), LDS-DMA (global_load_lds_dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- the 16-bit operand type X of this translation unit ---------------------------------------------------------------------
// Every kernel that touches 16-bit activations / weights is written against x16_t (raw bits in memory), x16x8 (one MFMA A/B fragment),
// x2f / f2x / pack2x / unpack8 / pack8 (conversions, round-to-nearest-even) and mfma16 — and build.py compiles those translation units
// TWICE into the one library: X = bf16 (8 mantissa bits, fp32 range; the split-precision PaDT decoder's (hi, lo) pairs, and the A/B
// fallback for ViT / LLM) and X = IEEE fp16 (11 mantissa bits; `v_mfma_f32_16x16x32_f16` runs at the bf16 rate on gfx950 and the LDS
// image, DMA pieces and `ds_read_b64_tr_b16` are type-agnostic) — the default ViT / LLM operand type since round 4: the same kernels
// land 8x closer to the fp32 reference (tests/studies/operand_attribution.py).  An instantiation lives in its own namespace and exports
// its C entry points under its own names: padt_gemm_bf16 / padt_gemm_f16, padt_row_rstd / padt_row_rstd_f16 (include/padt_hip_f16.h).
#ifndef PADT_OP16_F16
#define PADT_OP16_F16 0
#endif
#define PADT_CAT_(a, b) a##b
#define PADT_CAT(a, b) PADT_CAT_(a, b)
#define PADT_CAT3(a, b, c) PADT_CAT(PADT_CAT(a, b), c)
#if PADT_OP16_F16
#define PADT_T16 f16
#define PADT_NS padt_x_f16
#define PADT_TWIN(name) name##_f16                       // padt_row_rstd → padt_row_rstd_f16
#else
#define PADT_T16 bf16
#define PADT_NS padt_x_bf16
#define PADT_TWIN(name) name                             // padt_row_rstd
#endif
#define PADT_SYM(pre, post) PADT_CAT3(pre, PADT_T16, post)   // PADT_SYM(padt_gemm_, ) → padt_gemm_bf16 / padt_gemm_f16

typedef unsigned short x16_t;                                    // raw 16-bit float bits in memory (bf16 or fp16, see above)
#if PADT_OP16_F16
typedef _Float16 x16n_t;                                         // the native scalar type
#else
typedef __bf16 x16n_t;
#endif
typedef __attribute__((ext_vector_type(8))) x16n_t x16x8;        // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) x16n_t x16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;         // one 16x16 MFMA C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16 bytes
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;  // 8 bytes

// fp32 residual stream → its 16-bit MIRROR (the A operand of the next projection, padt_gemm_resid32 and friends): the mirror holds
// X(STREAM_SCALE * x32).  A residual stream is un-normalised — real checkpoints carry "massive activations" of 1e3-1e4 in a few channels —
// and every consumer of a mirror is scale-invariant (row_rstd / the fused RMSNorm statistics divide the factor out again, given
// eps * STREAM_SCALE^2), so the fp16 instantiation stores it 2^-4 down: 16x head room above fp16's 65504 at no cost in precision for the
// elements that matter (|x| < 1e-3 falls into fp16 subnormals: absolute error 5e-7).  bf16 has fp32's range: factor 1.
#if PADT_OP16_F16
#define PADT_STREAM_SCALE 0.0625f
#else
#define PADT_STREAM_SCALE 1.0f
#endif

#define PADT_DEV __device__ __forceinline__

// Host side: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE — a process that drives several GPUs must raise the limit
// on each of them.  One bit per device ordinal; the attribute is set BEFORE the bit (two threads may both set it, none launches early).
#include <atomic>
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    template <class F> void run(F&& set_attribute) {
        int d = 0;
        (void)hipGetDevice(&d);
        const unsigned long long bit = 1ull << (d & 63);
        if (mask.load(std::memory_order_acquire) & bit) return;
        set_attribute();
        mask.fetch_or(bit, std::memory_order_release);
    }
};

// In-kernel launch timing (bench.py's in-situ roofline): thread 0 of every block folds its start / end wall-clock tick (100 MHz constant
// clock, s_memrealtime) into {min start, max end} of the launch's slot.  No barrier, no extra packet in the queue: unlike an event pair
// around the launch it neither serialises the stream nor counts the dispatch gap.  prof == nullptr: two predicated-off instructions.
struct ProfScope {
    unsigned long long* slot;
    __device__ __forceinline__ ProfScope(unsigned long long* s, int tid) : slot(tid == 0 ? s : nullptr) {
        if (slot) atomicMin(slot, (unsigned long long)wall_clock64());
    }
    __device__ __forceinline__ ~ProfScope() {
        if (slot) atomicMax(slot + 1, (unsigned long long)wall_clock64());
    }
};

// round-to-nearest-even via the hardware conversions (v_cvt_pk_bf16_f32 / v_cvt_f16_f32); NaN stays NaN, +-inf stays +-inf; fp16
// overflows to +-inf above 65504 (a NaN downstream, never a silently wrong number)
#if PADT_OP16_F16
PADT_DEV float x2f(x16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
PADT_DEV void unpack2x(unsigned v, float& lo, float& hi) {
    const x16x2 h = __builtin_bit_cast(x16x2, v);
    lo = (float)h[0];
    hi = (float)h[1];
}
#else
PADT_DEV float x2f(x16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
PADT_DEV void unpack2x(unsigned v, float& lo, float& hi) {
    lo = __builtin_bit_cast(float, v << 16);
    hi = __builtin_bit_cast(float, v & 0xffff0000u);
}
#endif
PADT_DEV x16_t f2x(float f) { return __builtin_bit_cast(x16_t, (x16n_t)f); }
PADT_DEV unsigned pack2x(float lo, float hi) {
    x16x2 v = {(x16n_t)lo, (x16n_t)hi};
    return __builtin_bit_cast(unsigned, v);
}
// 4 values of one 8-byte vector
PADT_DEV void unpack4x(u32x2 v, float* f) {
    unpack2x(v[0], f[0], f[1]);
    unpack2x(v[1], f[2], f[3]);
}

PADT_DEV x16x8 ld_frag(const void* p) { return *reinterpret_cast<const x16x8*>(p); }

PADT_DEV x16x8 zero_frag() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(x16x8, z);
}

// 8 OCP e4m3 bytes (two dwords, element j = byte j) → one MFMA fragment; exact (e4m3 ⊂ bf16, ⊂ fp16): v_cvt_pk_f32_fp8 + the pack above
typedef __attribute__((ext_vector_type(2))) float f32x2;
PADT_DEV x16x8 fp8x8_to_x16x8(unsigned lo, unsigned hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4 r = {pack2x(a[0], a[1]), pack2x(b[0], b[1]), pack2x(c[0], c[1]), pack2x(d[0], d[1])};
    return __builtin_bit_cast(x16x8, r);
}

#if PADT_OP16_F16
PADT_DEV f32x4 mfma16(x16x8 a, x16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#else
PADT_DEV f32x4 mfma16(x16x8 a, x16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif

// Cross-block hand-off inside one kernel (split-K / split-KV "last block reduces"): the 8 XCDs have separate L2s, so
// partials are written and read with agent-scope relaxed atomics (sc1: served at the device coherence point) instead
// of agent-scope fences, which write back / invalidate whole caches.  Protocol: st_agent(...)*, handoff_arrive(),
// and in the block that drew the last ticket: ld_agent(...)*.
PADT_DEV void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PADT_DEV float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The same hand-off 16 bytes at a time (round 6): a scalar sc1 store is one fabric write each — a dword costs ≈6x a dwordx4 per byte
// (MI355X_MICROARCH.md, "stores of each flavour") — and the split-K partials are whole f32x4 fragments.  sc1 = write-through / L1-bypassing at
// agent scope on both sides (the guide's valid form "{sc1 stores and loads both sides}" + the drained ticket of handoff_arrive / the
// s_waitcnt vmcnt(0) in front of the ticket); the load waits for its own data before returning.
PADT_DEV void st_agent4(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
PADT_DEV f32x4 ld_agent4(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// one-wave blocks (or wave 0 only): returns true in the block that arrives last; the ticket is left at zero
PADT_DEV bool handoff_arrive(int* ticket, int total, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's partial stores are acknowledged
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != total - 1) return false;
    if (lane == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Rotate-half RoPE fused into a GEMM epilogue.  The projection's output columns are stored PAIR-INTERLEAVED per head
// ((d, d + D/2) adjacent: weights.py permutes the q / k weight rows at load; q·k dot products do not care about a common
// permutation of d), so the 4 consecutive columns a lane holds are two complete rotation pairs.
struct RopeEpi {
    const float* cos; const float* sin; long ld;   // fp32 [row][>= D/2] tables; cos == nullptr → off
    int cols;                                      // columns [0, cols) are rotated (q and k), the rest (v) pass through
    int D;                                         // head width (D % 4 == 0)
};

// One rotation pair — the SAME two roundings at every rope site of the library.  Left to itself hipcc contracts x1*c - x2*s into
// fma(x1, c, -(x2*s)) in one kernel and fma(-x2, s, x1*c) in another, so the prompt and the decode path of one token could differ in the
// last fp32 bit (a rare bf16 flip, caught by a test that draws fresh positions on every run).
// In the fp16 instantiation there is a third way to differ: where the rotated value goes straight into a 16-bit store hipcc folds the fma and the
// fp32 → fp16 conversion into ONE v_fma_mixlo_f16 (a single rounding of the exact fma), elsewhere it emits v_fma_f32 + v_cvt_(pk_)f16_f32 (two
// roundings).  The two agree except when the fp32 result lands on an fp16 tie — about one element in 2^13 (round 4: one q element of 6144
// made the fused decode attention differ from its unfused test reference in 23 outputs; tools/diag/decode_attn_paths.py).  rounded32() makes
// the fp32 value opaque to the combiner; the scalar rotation sites (the ones that got the mixed instruction) wrap their result in it, so every
// site rounds twice — "rotate in fp32, then cast", the reference's rule (HF:557-599).
#if PADT_OP16_F16
PADT_DEV float rounded32(float r) { asm("" : "+v"(r)); return r; }
#else
PADT_DEV float rounded32(float r) { return r; }                  // no mixed-precision fma writes bf16: nothing to pin
#endif
// RANGE (round 5).  q and k are un-normalised projections: with fp16 operands a value beyond 65504 would become +-inf in the 16-bit store, and
// an infinite k (or q) is the ONE overflow that can vanish silently — q·k = -inf is a key the softmax drops, the output stays finite and wrong.
// Every rotated value therefore leaves through rope_fin(): in the fp16 instantiation a value fp16 cannot hold becomes NaN, which no kernel
// downstream can lose (scores, soft-max, P·V, the residual stream and the final norm all propagate it) and which the finite checks of the
// product path (padt_check_finite on the hidden rows of every prefill / decode step and on the ViT outputs) report.  Everywhere else an
// overflow is +-inf and reaches the stream as inf / NaN by itself (h = silu(g)·u → down_proj, v → P·V → o_proj).  rope_fin also pins
// "rotate in fp32, THEN cast" (rounded32) at every site, vector paths included.  bf16 has fp32's range: identity.
#if PADT_OP16_F16
PADT_DEV float rope_fin(float r) {
    r = rounded32(r);
    return __builtin_fabsf(r) <= 65504.0f ? r : __builtin_nanf("");
}
#else
PADT_DEV float rope_fin(float r) { return r; }
#endif
PADT_DEV float rope_lo(float x1, float x2, float c, float s) { return rope_fin(__builtin_fmaf(-x2, s, x1 * c)); }   // x1 cos - x2 sin
PADT_DEV float rope_hi(float x1, float x2, float c, float s) { return rope_fin(__builtin_fmaf(x1, s, x2 * c)); }    // x2 cos + x1 sin

PADT_DEV void rope_pairs(float* o, int m, int n, const RopeEpi& r) {
    if (r.cos == nullptr || n >= r.cols) return;
    const int i = (n % r.D) >> 1;                  // pair index of columns (n, n+1); (n+2, n+3) is pair i + 1
    // i is even (n % 4 == 0, D % 4 == 0) and ld is even → one 8-byte load per table
    const float2 cc = *reinterpret_cast<const float2*>(r.cos + (long)m * r.ld + i);
    const float2 ss = *reinterpret_cast<const float2*>(r.sin + (long)m * r.ld + i);
    const float c0 = cc.x, c1 = cc.y, s0 = ss.x, s1 = ss.y;
    const float a0 = o[0], b0 = o[1], a1 = o[2], b1 = o[3];
    o[0] = rope_lo(a0, b0, c0, s0);
    o[1] = rope_hi(a0, b0, c0, s0);
    o[2] = rope_lo(a1, b1, c1, s1);
    o[3] = rope_hi(a1, b1, c1, s1);
}

PADT_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of an IEEE division: the SwiGLU epilogue of the gate/up GEMMs evaluates 64 of these per
// lane per tile and the division's ~10-instruction sequence was a third of that epilogue (profiles/r02_gemm256_experiments.md)
PADT_DEV float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// unpack 8 X values (one 16-byte vector) to floats
PADT_DEV void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2x(v[i], f[2 * i], f[2 * i + 1]);
}
PADT_DEV u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2x(f[2 * i], f[2 * i + 1]);
    return v;
}

PADT_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
PADT_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware, bijective remap of a 1-D block id (guide T1): each of the 8 XCDs (private L2) gets a contiguous chunk of
// the tile space, so neighbouring tiles that share an operand panel hit the same L2.
PADT_DEV int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / nx;
}
