// Device-side helpers shared by every kernel file.  gfx950 (MI355X, CDNA4) only: wave = 64 lanes,
// MFMA 16x16x32 bf16 fragments, LDS-DMA (global_load_lds_dwordx4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;       // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;         // one 16x16 MFMA C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // 16 bytes
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;  // 8 bytes

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

PADT_DEV float bf2f(bf16_t v) { return __builtin_bit_cast(float, (unsigned)v << 16); }

// round-to-nearest-even via the gfx950 hardware conversion (v_cvt_pk_bf16_f32); NaN stays NaN, +-inf stays +-inf
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
PADT_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
PADT_DEV unsigned pack2bf(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

PADT_DEV bf16x8 ld_frag(const void* p) { return *reinterpret_cast<const bf16x8*>(p); }

PADT_DEV bf16x8 zero_frag() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}

// 8 OCP e4m3 bytes (two dwords, element j = byte j) → one bf16 MFMA fragment; exact (e4m3 ⊂ bf16): v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float f32x2;
PADT_DEV bf16x8 fp8x8_to_bf16x8(unsigned lo, unsigned hi) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
    const u32x4 r = {pack2bf(a[0], a[1]), pack2bf(b[0], b[1]), pack2bf(c[0], c[1]), pack2bf(d[0], d[1])};
    return __builtin_bit_cast(bf16x8, r);
}

PADT_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// Cross-block hand-off inside one kernel (split-K / split-KV "last block reduces"): the 8 XCDs have separate L2s, so
// partials are written and read with agent-scope relaxed atomics (sc1: served at the device coherence point) instead
// of agent-scope fences, which write back / invalidate whole caches.  Protocol: st_agent(...)*, handoff_arrive(),
// and in the block that drew the last ticket: ld_agent(...)*.
PADT_DEV void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
PADT_DEV float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
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
PADT_DEV float rope_lo(float x1, float x2, float c, float s) { return __builtin_fmaf(-x2, s, x1 * c); }   // x1 cos - x2 sin
PADT_DEV float rope_hi(float x1, float x2, float c, float s) { return __builtin_fmaf(x1, s, x2 * c); }    // x2 cos + x1 sin

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

// unpack 8 bf16 (one 16-byte vector) to floats
PADT_DEV void unpack8(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
    }
}
PADT_DEV u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
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
