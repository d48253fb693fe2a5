// PROTOTYPE (not part of libpadt_hip.so): decode-step attention with mRoPE + KV append, second form — DESIGN.md §6.5 item 1.
//
// Same decomposition as csrc/attention.hip's decode_attn_rope_kernel (grid = (64-key split, kv head, sample), one wave per block, fp32
// partials (m, l, O) per split, merged by a second launch) and the same partial layout, but the single wave's serial chain is shorter:
//   * NO LDS and NO barrier.  The q rows of the group and the fresh k are rotated in registers straight into MFMA fragments: lane
//     (frow, fq) loads chunks fq, fq+4, fq+8, fq+12 (16 B each) of head `frow` — chunk c and chunk c+8 are a rotate-half pair — so the
//     four rotated chunks ARE its four K-step fragments.  The fresh k is rotated by every lane for its own fq (16x redundant, 16 pairs).
//   * scores are computed TRANSPOSED (S^T = K Q^T: keys on the MFMA rows, heads on the columns), so a lane holds 16 keys of ONE head:
//     soft-max statistics are in-lane + 2 shuffles (instead of 4 rows x 4 shuffles twice), and the probabilities feed the second MFMA
//     (O^T = V^T P^T) from registers as its B operand — no 16-bit round trip through LDS.  The 8 keys of a B fragment are two runs of 4
//     (keys fq*4.. of two neighbouring 16-key blocks); the V^T fragment is loaded as the same two 8-byte runs.
//   * dead query rows (group < 16) are never zero-filled, computed on, or stored: MFMA columns are independent, a lane with
//     frow >= group carries zeros and skips its stores (half of the 10.5 MB of partials per layer at group = 8).
// HF:557-599 (mRoPE), :665-666 (cache update), :641-689 (attention with Lq == 1).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPADT_OP16_F16=1 -shared -fPIC tools/ubench/decode_attn_v2.hip -o tools/ubench/libdecode_attn_v2.so
//   python tools/bench_decode_attn_v2.py            (GPU box: parity against the library's kernel + timing)
#include "../../padt_amd/csrc/common.h"
#include <math.h>

namespace {

struct Args {
    const x16_t* qkv; long ld_qkv;   // [B][(Hq + 2 Hkv) * D], bias already added
    const float* rope_cs;             // [B][D/2][2] cos, sin of this step's position
    const int* slot;                  // [B] append index; valid keys afterwards = slot + 1
    x16_t* kc; x16_t* vtc;           // K cache [B][Hkv][S_max][D], V^T cache [B][Hkv][D][S_max]
    float* part_o; float* part_ml;    // [B][Hkv][nsplit][16][D], [B][Hkv][nsplit][16][2]
    x16_t* out;                      // [B][Hq * D] (merge)
    int B, Hq, Hkv, S_max, nsplit;
    float scale_log2;
    int out_packed = 0;               // v3: out in the 16-row fragment-packed activation layout (row length Hq * D)
};

// rotate one chunk pair (8 rotate-half pairs): x1 = chunk c (d = 8c ..), x2 = chunk c + D/16 (d + D/2 ..), cs = (cos, sin) of d = 8c .. 8c+7
PADT_DEV void rope_chunk_pair(const u32x4& r1, const u32x4& r2, const float2* cs, u32x4& lo, u32x4& hi) {
    float x1[8], x2[8], o1[8], o2[8];
    unpack8(r1, x1);
    unpack8(r2, x2);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o1[e] = rope_lo(x1[e], x2[e], cs[e].x, cs[e].y);
        o2[e] = rope_hi(x1[e], x2[e], cs[e].x, cs[e].y);
    }
    lo = pack8(o1);
    hi = pack8(o2);
}

// PACKED: the caches are stored fragment by fragment, in lane order — K as [S_max/16][D/32][64 lanes][8], V^T as [D/16][S_max/32][64 lanes][8]
// (element e of lane (frow, fq) = key 32 ks + (e < 4 ? 4 fq + e : 16 + 4 fq + e - 4) of row 16 i + frow) — so one wave-wide load
// instruction reads 1 KiB of contiguous memory instead of 16 rows x 64 B (K) or 32 pieces of 8 B (V^T): what csrc/gemm.hip's packed weight
// image does for the decode projections (row-major fragments run the address unit at a quarter of its rate).
template <int D, bool PACKED>
__global__ __launch_bounds__(64) void decode_attn_rope_v2_kernel(Args p) {
#pragma clang fp contract(off)
    static_assert(D == 128, "fragment map below is written for 16 chunks per head");
    constexpr int KQ = D / 32, NB = D / 16, HALF = D / 2;
    const int lane = threadIdx.x, frow = lane & 15, fq = lane >> 4;
    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int group = p.Hq / p.Hkv;
    const int slot = p.slot[b];
    const int len = slot + 1;
    const int k0 = split * 64;
    const long pbase = (((long)b * p.Hkv + g) * p.nsplit + split) * 16;
    if (k0 >= len) {                                              // empty split: neutral partial
        if (lane < 16) { p.part_ml[(pbase + lane) * 2] = -INFINITY; p.part_ml[(pbase + lane) * 2 + 1] = 0.f; }
        return;
    }
    const x16_t* row = p.qkv + (long)b * p.ld_qkv;
    x16_t* kbase = p.kc + ((long)b * p.Hkv + g) * p.S_max * D;
    x16_t* vbase = p.vtc + ((long)b * p.Hkv + g) * D * (long)p.S_max;
    const bool owner = (slot >= k0) && (slot < k0 + 64);
    const bool liveq = frow < group;

    // ---- every global load of the block, issued back to back -------------------------------------------------------------
    const x16_t* qrow = row + (long)(g * group + (liveq ? frow : 0)) * D;
    const x16_t* krow = row + (long)(p.Hq + g) * D;
    const x16_t* vrow = row + (long)(p.Hq + p.Hkv + g) * D;
    u32x4 qraw[4], knraw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                  // chunks fq, fq+4 (first half), fq+8, fq+12 (their partners)
        qraw[j] = *reinterpret_cast<const u32x4*>(qrow + (fq + 4 * j) * 8);
        knraw[j] = *reinterpret_cast<const u32x4*>(krow + (fq + 4 * j) * 8);
    }
    float2 cs[2][8];                                              // (cos, sin) of d = 8 fq .. and d = 8 (fq + 4) ..
    {
        const float* csb = p.rope_cs + (long)b * HALF * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float4 v = *reinterpret_cast<const float4*>(csb + 2 * ((fq + 4 * h) * 8 + e));
                cs[h][e] = float2{v.x, v.y};
                cs[h][e + 1] = float2{v.z, v.w};
            }
    }
    u32x4 kraw[4][KQ];
    u32x4 vfr[2][NB];                                             // word 0, 1 = run 0 (keys 32 ks + 4 fq ..), word 2, 3 = run 1 (+ 16)
    if constexpr (PACKED) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk)
                kraw[kb][kk] = *reinterpret_cast<const u32x4*>(kbase + ((long)((k0 >> 4) + kb) * KQ + kk) * 512 + lane * 8);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NB; ++i)
                vfr[ks][i] = *reinterpret_cast<const u32x4*>(vbase + ((long)i * (p.S_max >> 5) + (k0 >> 5) + ks) * 512 + lane * 8);
    } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int key = k0 + kb * 16 + frow;
            const int kcl = key < p.S_max ? key : p.S_max - 1;
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) kraw[kb][kk] = *reinterpret_cast<const u32x4*>(kbase + (long)kcl * D + kk * 32 + fq * 8);
        }
        // V^T fragments: for K-step ks the lane's 8 keys are runs [32 ks + 4 fq, +4) and [32 ks + 16 + 4 fq, +4) of the split
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const x16_t* vr = vbase + (long)(i * 16 + frow) * p.S_max + k0 + ks * 32 + fq * 4;
                const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                vfr[ks][i] = u32x4{v0[0], v0[1], v1[0], v1[1]};
            }
    }
    unsigned vnew[NB];                                            // fresh v at d = 16 i + frow (the rows of this lane's V^T fragments)
    unsigned vapp[2];                                             // fresh v at d = lane, lane + 64 (the cache append)
    if (owner) {
#pragma unroll
        for (int i = 0; i < NB; ++i) vnew[i] = vrow[i * 16 + frow];
        vapp[0] = vrow[lane];
        vapp[1] = vrow[lane + 64];
    }

    // ---- rotate q (this lane's head) and the fresh k into fragments ---------------------------------------------------------
    u32x4 qf[KQ], kn[KQ];
    rope_chunk_pair(qraw[0], qraw[2], cs[0], qf[0], qf[2]);        // chunks (fq, fq + 8)   → K-steps 0 and 2
    rope_chunk_pair(qraw[1], qraw[3], cs[1], qf[1], qf[3]);        // chunks (fq+4, fq+12)  → K-steps 1 and 3
    rope_chunk_pair(knraw[0], knraw[2], cs[0], kn[0], kn[2]);
    rope_chunk_pair(knraw[1], knraw[3], cs[1], kn[1], kn[3]);
    if (!liveq) {
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) qf[kk] = u32x4{0u, 0u, 0u, 0u};
    }
    if (owner) {
        if constexpr (PACKED) {
            if (frow == 0) {
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk)
                    *reinterpret_cast<u32x4*>(kbase + ((((long)(slot >> 4) * KQ + kk) * 64) + fq * 16 + (slot & 15)) * 8) = kn[kk];
            }
            const int r32 = slot & 31;
            const long vcol = ((long)(slot >> 5) * 64 + ((r32 & 15) >> 2) * 16) * 8 + (r32 >> 4) * 4 + (r32 & 3);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d = lane + 64 * h;
                vbase[(long)(d >> 4) * (p.S_max >> 5) * 512 + vcol + (d & 15) * 8] = (x16_t)vapp[h];
            }
        } else {
            if (frow == 0) {
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) *reinterpret_cast<u32x4*>(kbase + (long)slot * D + kk * 32 + fq * 8) = kn[kk];
            }
            vbase[(long)lane * p.S_max + slot] = (x16_t)vapp[0];
            vbase[(long)(lane + 64) * p.S_max + slot] = (x16_t)vapp[1];
        }
    }

    // where the appended token sits inside this split (block-uniform: slot comes from a scalar load)
    const int rel = slot - k0;                                    // 0 .. 63 in the owner block

    // ---- S^T = K Q^T: lane holds s[kb][r] = score of key k0 + 16 kb + 4 fq + r for head frow -----------------------------------
    f32x4 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 kf[KQ];
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) kf[kk] = kraw[kb][kk];
        if (owner && (rel >> 4) == kb) {                           // uniform branch: one 16-key block holds the appended token,
            const bool fresh = frow == (rel & 15);                 // and one A row of it is that token
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) kf[kk] = fresh ? kn[kk] : kf[kk];
        }
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) s[kb] = mfma16(__builtin_bit_cast(x16x8, kf[kk]), __builtin_bit_cast(x16x8, qf[kk]), s[kb]);
    }
    float m = -INFINITY;
    if (k0 + 64 > len) {                                          // uniform: only the last split has keys beyond len
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + kb * 16 + fq * 4 + r;
                const float x = (key < len) ? s[kb][r] * p.scale_log2 : -INFINITY;
                s[kb][r] = x;
                m = fmaxf(m, x);
            }
    } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = s[kb][r] * p.scale_log2;
                s[kb][r] = x;
                m = fmaxf(m, x);
            }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));                          // finite: key k0 < len exists
    float l = 0.f;
    u32x4 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float pv[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[2 * ks + h][r] - m);
                pv[h * 4 + r] = e;
                l += e;
            }
        pf[ks] = pack8(pv);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);

    // ---- O^T = V^T P^T: lane holds o[i][r] = O[head frow][d = 16 i + 4 fq + r] ---------------------------------------------------
    f32x4 o[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if (owner && (rel >> 5) == ks) {                           // uniform branch: this K-step holds the appended token: 16-bit insert of the
            const bool mine = fq == ((rel >> 2) & 3);              // fresh v into run (rel >> 4) & 1, element rel & 3 of the lanes with this fq
            const int run = (rel >> 4) & 1, w = (rel >> 1) & 1, half = rel & 1;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const unsigned nv = vnew[i] & 0xffffu;
                const int wi = run * 2 + w;                        // uniform
                const unsigned cur = wi == 0 ? vfr[ks][i][0] : (wi == 1 ? vfr[ks][i][1] : (wi == 2 ? vfr[ks][i][2] : vfr[ks][i][3]));
                const unsigned pat = half ? ((cur & 0x0000ffffu) | (nv << 16)) : ((cur & 0xffff0000u) | nv);
                const unsigned val = mine ? pat : cur;
#pragma unroll
                for (int q = 0; q < 4; ++q) vfr[ks][i][q] = (wi == q) ? val : vfr[ks][i][q];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) o[i] = mfma16(__builtin_bit_cast(x16x8, vfr[ks][i]), __builtin_bit_cast(x16x8, pf[ks]), o[i]);
    }
    if (liveq) {
        float* po = p.part_o + (pbase + frow) * D;
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(po + i * 16 + fq * 4) = o[i];
        if (fq == 0) *reinterpret_cast<float2*>(p.part_ml + (pbase + frow) * 2) = float2{m, l};
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3 (round 6): ONE launch.  A block = NW waves owns one (kv head, sample): wave w streams the 64-key splits w, w + NW, ... with v2's
// register-only body and folds them into a running (m, l, O^T) — flash attention's online rescale, so no per-split partial ever leaves
// the wave —, the NW states meet in LDS and every wave finishes its share of the O^T d-tiles and stores 16-bit output rows.  No partial
// traffic (10.5 MB written + 5 MB re-read per layer at 64 rows), no second launch (the two-launch floor is 8.9 us at 8 samples).
template <int D, int NW, bool PACKED>
__global__ __launch_bounds__(NW * 64) void decode_attn_rope_v3_kernel(Args p) {
#pragma clang fp contract(off)
    static_assert(D == 128, "fragment map below is written for 16 chunks per head");
    constexpr int KQ = D / 32, NB = D / 16, HALF = D / 2;
    extern __shared__ __attribute__((aligned(16))) float v3_lds[];
    float* sm_ml = v3_lds;                                        // [NW][64][2]
    f32x4* sm_o = reinterpret_cast<f32x4*>(v3_lds + NW * 64 * 2); // [NW][NB][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, frow = lane & 15, fq = lane >> 4;
    const int g = blockIdx.x, b = blockIdx.y;
    const int group = p.Hq / p.Hkv;
    const int slot = p.slot[b];
    const int len = slot + 1;
    const x16_t* row = p.qkv + (long)b * p.ld_qkv;
    x16_t* kbase = p.kc + ((long)b * p.Hkv + g) * p.S_max * D;
    x16_t* vbase = p.vtc + ((long)b * p.Hkv + g) * D * (long)p.S_max;
    const bool liveq = frow < group;
    float M = -INFINITY, L = 0.f;
    f32x4 o[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave * 64 < len) {
        // ---- q (this lane's head) and the fresh k, rotated in registers straight into fragments (v2)
        const x16_t* qrow = row + (long)(g * group + (liveq ? frow : 0)) * D;
        const x16_t* krow = row + (long)(p.Hq + g) * D;
        const x16_t* vrow = row + (long)(p.Hq + p.Hkv + g) * D;
        u32x4 qraw[4], knraw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qraw[j] = *reinterpret_cast<const u32x4*>(qrow + (fq + 4 * j) * 8);
            knraw[j] = *reinterpret_cast<const u32x4*>(krow + (fq + 4 * j) * 8);
        }
        float2 cs[2][8];
        {
            const float* csb = p.rope_cs + (long)b * HALF * 2;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float4 v = *reinterpret_cast<const float4*>(csb + 2 * ((fq + 4 * h) * 8 + e));
                    cs[h][e] = float2{v.x, v.y};
                    cs[h][e + 1] = float2{v.z, v.w};
                }
        }
        u32x4 qf[KQ], kn[KQ];
        rope_chunk_pair(qraw[0], qraw[2], cs[0], qf[0], qf[2]);
        rope_chunk_pair(qraw[1], qraw[3], cs[1], qf[1], qf[3]);
        rope_chunk_pair(knraw[0], knraw[2], cs[0], kn[0], kn[2]);
        rope_chunk_pair(knraw[1], knraw[3], cs[1], kn[1], kn[3]);
        if (!liveq) {
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) qf[kk] = u32x4{0u, 0u, 0u, 0u};
        }
        for (int k0 = wave * 64; k0 < len; k0 += NW * 64) {
            const bool owner = (slot >= k0) && (slot < k0 + 64);
            u32x4 kraw[4][KQ];
            u32x4 vfr[2][NB];
            if constexpr (PACKED) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int kk = 0; kk < KQ; ++kk)
                        kraw[kb][kk] = *reinterpret_cast<const u32x4*>(kbase + ((long)((k0 >> 4) + kb) * KQ + kk) * 512 + lane * 8);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < NB; ++i)
                        vfr[ks][i] = *reinterpret_cast<const u32x4*>(vbase + ((long)i * (p.S_max >> 5) + (k0 >> 5) + ks) * 512 + lane * 8);
            } else {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const int key = k0 + kb * 16 + frow;
                    const int kcl = key < p.S_max ? key : p.S_max - 1;
#pragma unroll
                    for (int kk = 0; kk < KQ; ++kk) kraw[kb][kk] = *reinterpret_cast<const u32x4*>(kbase + (long)kcl * D + kk * 32 + fq * 8);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const x16_t* vr = vbase + (long)(i * 16 + frow) * p.S_max + k0 + ks * 32 + fq * 4;
                        const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                        vfr[ks][i] = u32x4{v0[0], v0[1], v1[0], v1[1]};
                    }
            }
            unsigned vnew[NB];
            if (owner) {
                unsigned vapp[2];
#pragma unroll
                for (int i = 0; i < NB; ++i) vnew[i] = vrow[i * 16 + frow];
                vapp[0] = vrow[lane];
                vapp[1] = vrow[lane + 64];
                if constexpr (PACKED) {
                    if (frow == 0) {
#pragma unroll
                        for (int kk = 0; kk < KQ; ++kk)
                            *reinterpret_cast<u32x4*>(kbase + ((((long)(slot >> 4) * KQ + kk) * 64) + fq * 16 + (slot & 15)) * 8) = kn[kk];
                    }
                    const int r32 = slot & 31;
                    const long vcol = ((long)(slot >> 5) * 64 + ((r32 & 15) >> 2) * 16) * 8 + (r32 >> 4) * 4 + (r32 & 3);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = lane + 64 * h;
                        vbase[(long)(d >> 4) * (p.S_max >> 5) * 512 + vcol + (d & 15) * 8] = (x16_t)vapp[h];
                    }
                } else {
                    if (frow == 0) {
#pragma unroll
                        for (int kk = 0; kk < KQ; ++kk) *reinterpret_cast<u32x4*>(kbase + (long)slot * D + kk * 32 + fq * 8) = kn[kk];
                    }
                    vbase[(long)lane * p.S_max + slot] = (x16_t)vapp[0];
                    vbase[(long)(lane + 64) * p.S_max + slot] = (x16_t)vapp[1];
                }
            }
            const int rel = slot - k0;
            f32x4 s[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
                u32x4 kf[KQ];
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) kf[kk] = kraw[kb][kk];
                if (owner && (rel >> 4) == kb) {
                    const bool fresh = frow == (rel & 15);
#pragma unroll
                    for (int kk = 0; kk < KQ; ++kk) kf[kk] = fresh ? kn[kk] : kf[kk];
                }
#pragma unroll
                for (int kk = 0; kk < KQ; ++kk) s[kb] = mfma16(__builtin_bit_cast(x16x8, kf[kk]), __builtin_bit_cast(x16x8, qf[kk]), s[kb]);
            }
            float m = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kb * 16 + fq * 4 + r;
                    const float x = (key < len) ? s[kb][r] * p.scale_log2 : -INFINITY;
                    s[kb][r] = x;
                    m = fmaxf(m, x);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));                  // finite: key k0 < len exists
            const float Mn = fmaxf(M, m);
            const float al = exp2f(M - Mn);                       // first split: exp2(-inf) = 0
            M = Mn;
            L *= al;
#pragma unroll
            for (int i = 0; i < NB; ++i) o[i] *= al;
            u32x4 pf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float pv[8];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = exp2f(s[2 * ks + h][r] - Mn);
                        pv[h * 4 + r] = e;
                        L += e;                                   // lane-partial over its own keys; reduced across fq after the loop
                    }
                pf[ks] = pack8(pv);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (owner && (rel >> 5) == ks) {
                    const bool mine = fq == ((rel >> 2) & 3);
                    const int run = (rel >> 4) & 1, w = (rel >> 1) & 1, half = rel & 1;
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const unsigned nv = vnew[i] & 0xffffu;
                        const int wi = run * 2 + w;
                        const unsigned cur = wi == 0 ? vfr[ks][i][0] : (wi == 1 ? vfr[ks][i][1] : (wi == 2 ? vfr[ks][i][2] : vfr[ks][i][3]));
                        const unsigned pat = half ? ((cur & 0x0000ffffu) | (nv << 16)) : ((cur & 0xffff0000u) | nv);
                        const unsigned val = mine ? pat : cur;
#pragma unroll
                        for (int q = 0; q < 4; ++q) vfr[ks][i][q] = (wi == q) ? val : vfr[ks][i][q];
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) o[i] = mfma16(__builtin_bit_cast(x16x8, vfr[ks][i]), __builtin_bit_cast(x16x8, pf[ks]), o[i]);
            }
        }
        L += __shfl_xor(L, 16, 64);
        L += __shfl_xor(L, 32, 64);
    }
    // ---- the NW running states meet in LDS; wave w finishes the d-tiles w, w + NW, ... of every head
    sm_ml[(wave * 64 + lane) * 2] = M;
    sm_ml[(wave * 64 + lane) * 2 + 1] = L;
#pragma unroll
    for (int i = 0; i < NB; ++i) sm_o[(wave * NB + i) * 64 + lane] = o[i];
    __syncthreads();
    float Mx = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) Mx = fmaxf(Mx, sm_ml[(w * 64 + lane) * 2]);
    float wgt[NW], Ls = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const float mw = sm_ml[(w * 64 + lane) * 2];
        wgt[w] = mw == -INFINITY ? 0.f : exp2f(mw - Mx);
        Ls += wgt[w] * sm_ml[(w * 64 + lane) * 2 + 1];
    }
    const float inv = Ls > 0.f ? 1.f / Ls : 0.f;
    if (liveq) {
        const long ld = (long)p.Hq * D;
#pragma unroll
        for (int i = wave; i < NB; i += NW) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) acc += sm_o[(w * NB + i) * 64 + lane] * wgt[w];
            const int n = (g * group + frow) * D + i * 16 + fq * 4;   // 4 consecutive columns of output row b
            const long off = p.out_packed ? (long)(b >> 4) * 16 * ld + ((long)(n >> 3) * 16 + (b & 15)) * 8 + (n & 7) : (long)b * ld + n;
            *reinterpret_cast<u32x2*>(p.out + off) = u32x2{pack2x(acc[0] * inv, acc[1] * inv), pack2x(acc[2] * inv, acc[3] * inv)};
        }
    }
}

// the library's merge (csrc/attention.hip decode_combine_kernel), row-major output
template <int D>
__global__ void combine_kernel(Args p) {
#pragma clang fp contract(off)
    const int b = blockIdx.y, hq = blockIdx.x, d = threadIdx.x;
    const int group = p.Hq / p.Hkv;
    const int g = hq / group, hrow = hq % group;
    const long base0 = (((long)b * p.Hkv + g) * p.nsplit) * 16 + hrow;
    constexpr int CH = 8;
    float M = -INFINITY;
    for (int s0 = 0; s0 < p.nsplit; s0 += CH) {
        float mv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) mv[i] = (s0 + i < p.nsplit) ? p.part_ml[(base0 + (long)(s0 + i) * 16) * 2] : -INFINITY;
#pragma unroll
        for (int i = 0; i < CH; ++i) M = fmaxf(M, mv[i]);
    }
    float L = 0.f, acc = 0.f;
    for (int s0 = 0; s0 < p.nsplit; s0 += CH) {
        float mv[CH], lv[CH], ov[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const bool ok = s0 + i < p.nsplit;
            const long base = base0 + (long)(ok ? s0 + i : 0) * 16;
            mv[i] = ok ? p.part_ml[base * 2] : -INFINITY;
            lv[i] = p.part_ml[base * 2 + 1];
            ov[i] = (mv[i] == -INFINITY) ? 0.f : p.part_o[base * D + d];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (mv[i] == -INFINITY) continue;
            const float w = exp2f(mv[i] - M);
            L += w * lv[i];
            acc += w * ov[i];
        }
    }
    p.out[(long)b * p.Hq * D + hq * D + d] = f2x(L > 0.f ? acc / L : 0.f);
}

}  // namespace

extern "C" int decode_attn_rope_v2(void* stream, const void* qkv, long ld_qkv, const void* rope_cs, const int* slot, void* k_cache,
                                   void* vt_cache, void* out, void* workspace, int batch, int n_heads, int n_kv_heads, int head_dim,
                                   int s_max, int max_len, float scale, int packed) {
    if (head_dim != 128 || n_heads % n_kv_heads || n_heads / n_kv_heads > 16 || (s_max & 63) || (ld_qkv & 7) || max_len > s_max || max_len <= 0)
        return -1;
    const int nsplit = (max_len + 63) / 64;
    Args a{(const x16_t*)qkv, ld_qkv, (const float*)rope_cs, slot, (x16_t*)k_cache, (x16_t*)vt_cache, (float*)workspace, nullptr,
           (x16_t*)out, batch, n_heads, n_kv_heads, s_max, nsplit, scale * 1.4426950408889634f};
    a.part_ml = a.part_o + (long)batch * n_kv_heads * nsplit * 16 * head_dim;
    hipStream_t s = (hipStream_t)stream;
    if (packed) hipLaunchKernelGGL((decode_attn_rope_v2_kernel<128, true>), dim3(nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((decode_attn_rope_v2_kernel<128, false>), dim3(nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
    hipLaunchKernelGGL(combine_kernel<128>, dim3(n_heads, batch), dim3(128), 0, s, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// v3: one launch, NW waves per (kv head, sample); nw = 4 or 8
extern "C" int decode_attn_rope_v3(void* stream, const void* qkv, long ld_qkv, const void* rope_cs, const int* slot, void* k_cache,
                                   void* vt_cache, void* out, int batch, int n_heads, int n_kv_heads, int head_dim, int s_max, float scale,
                                   int packed, int nw, int out_packed) {
    if (head_dim != 128 || n_heads % n_kv_heads || n_heads / n_kv_heads > 16 || (s_max & 63) || (ld_qkv & 7) || (nw != 4 && nw != 8)) return -1;
    Args a{(const x16_t*)qkv, ld_qkv, (const float*)rope_cs, slot, (x16_t*)k_cache, (x16_t*)vt_cache, nullptr, nullptr,
           (x16_t*)out, batch, n_heads, n_kv_heads, s_max, 0, scale * 1.4426950408889634f, out_packed};
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)nw * 64 * 2 * 4 + (size_t)nw * 8 * 64 * 16;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_rope_v3_kernel<128, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_rope_v3_kernel<128, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr = true;
    }
    const dim3 grid(n_kv_heads, batch);
    if (nw == 8) {
        if (packed) hipLaunchKernelGGL((decode_attn_rope_v3_kernel<128, 8, true>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((decode_attn_rope_v3_kernel<128, 8, false>), grid, dim3(512), lds, s, a);
    } else {
        if (packed) hipLaunchKernelGGL((decode_attn_rope_v3_kernel<128, 4, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((decode_attn_rope_v3_kernel<128, 4, false>), grid, dim3(256), lds, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
