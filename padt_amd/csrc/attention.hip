// Varlen flash attention (prefill-style) and split-KV decode attention for gfx950, bf16 MFMA 16x16x32, fp32 softmax.
//
// attn_varlen_kernel<D, CAUSAL, QR>: one block (4 waves) = 64*QR query rows of one (segment, head); K/V tiles of 64
//   keys are staged in LDS (next tile prefetched into registers), both row-major with conflict-free row strides; V is
//   consumed through ds_read_b64_tr_b16 (MFMA wants both operands contiguous along the contraction index, which for
//   P·V is the key index: the transpose read delivers 4 keys of one d column per lane).  The whole tile is computed transposed (S^T = K Q^T, O^T = V^T P^T): a lane owns one query column, so
//   the online softmax is in-lane and P feeds the second MFMA straight from registers.  Never materialises S.
//   Replaces flash_attn_varlen_func at: HF ViT attention (28 window layers: 36 segments/img of 64/48/36 tokens; 4 full
//   layers: one 2116-token segment/img), LLM prefill (causal, GQA 16:2, d=128) and PaDTDecoderFlashAttention2.forward
//   (padt_decoder.py:55: query↔query, query→image, image→query; d=80).
// decode_attn_kernel: one wave per (64-key split, kv head, sample); the GQA group's q heads are the 16 MFMA rows; K is
//   read straight from the row-major K cache, V from the TRANSPOSED V cache (both 16-byte fragment loads, no LDS
//   staging: nothing is shared between waves); partial (m, l, O) per split are merged by decode_combine_kernel.
#include "common.h"
#include <cstdlib>
#include <cstdint>

extern "C" void padt_set_error(const char* msg);

namespace PADT_NS {

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef v4s_t __attribute__((address_space(3))) v4s_lds;

struct AttnArgs {
    const x16_t* q; long ldq;      // token stride (elements); head h at +h*D
    const x16_t* k; long ldk;      // kv head g at +g*D
    const x16_t* v; long ldv;
    x16_t* o; long ldo;
    const int* cu_q; const int* cu_k;
    int group;                      // q heads per kv head
    float scale_log2;               // softmax scale * log2(e)
    const float* rcos; const float* rsin; long ld_cs;   // optional fused rotate-half RoPE of q and k (self-attention:
                                                        // one fp32 table [token][>= D/2], same token index space)
};

template <int D> struct AttnCfg {
    static constexpr int KQ = (D + 31) / 32;          // k-steps for K Q^T
    static constexpr int NB = D / 16;                 // 16-wide d blocks for V^T P^T
    static constexpr int KROW = (D == 128) ? 144 : (D == 32 ? 48 : 80);   // K row stride (elements): the strides ≡ 16 (mod 32)
                                                      // elements are the conflict-free ones for the ds_read_b128 lane groups
    // V is staged ROW-major like K and read with the LDS transpose read (ds_read_b64_tr_b16): a 16-lane group hands in the
    // 16 8-byte chunks of a [4 keys][16 d] block and lane i gets column i (4 keys) — two of them are one A fragment of
    // O^T = V^T P^T.  Row stride ≡ 16/48/80/112 (mod 128) elements puts the 8 key rows x 32 B a half-wave touches on
    // 8 disjoint bank windows.
    static constexpr int VROW = (D == 128) ? 144 : (D == 32 ? 48 : 80);
    static constexpr int CPR = D / 8;                 // 16-byte chunks per K/V row
    static constexpr int NK = (64 * CPR + 255) / 256; // K (and V) chunks per thread per tile
    static constexpr int K_BYTES = 64 * KROW * 2;
    static constexpr int V_BYTES = 64 * VROW * 2;
    static constexpr int LDS = K_BYTES + V_BYTES;
    static constexpr int LDS2 = 2 * LDS;              // double-buffered (LDS-DMA path)
    static constexpr int PIECES = (K_BYTES + 1023) / 1024;   // 1-KiB LDS-DMA pieces per K (or V) tile
};

// Everything is computed TRANSPOSED so that a lane owns ONE query column: S^T = K Q^T puts the 16 keys of a block on the
// MFMA rows (lane: keys fq*4..+3, query frow) → the row max / row sum over keys are in-lane (plus two cross-group
// shuffles for the max; the sum is reduced once at the end), and the 8 probabilities a lane holds per 32 keys ARE a
// B-operand fragment of O^T = V^T P^T under the key permutation {32ks+4fq+j, 32ks+16+4fq+j} — P never touches LDS; the
// same permutation is applied to the V^T fragment (two 8-byte LDS reads).  O^T's rescale factor is in-lane as well.
// ROPE (window layers of the ViT, HF apply_rotary_pos_emb_vision :160-171): q fragments are rotated in registers when
// they are loaded, the staged K tile is rotated in place in LDS — same fp32 expressions and single bf16 rounding as the
// stand-alone rope_half kernel, without its extra pass over q and k in HBM.
// GQA: the block's 64 QR query rows are (token, q head of ONE kv group) pairs, row r = token r / group, head r % group, so a staged
// K / V tile is multiplied by all q heads that share it (the prompt pass: 8 q heads per kv head — one eighth of the K / V tile loads
// and LDS fills per MFMA of the head-per-block mapping); blockIdx.y is the kv head.
template <int D, bool CAUSAL, int QR, bool ROPE, bool GQA = false>
__global__ __launch_bounds__(256) void attn_varlen_kernel(AttnArgs p) {
    using C = AttnCfg<D>;
    constexpr int HALF = D / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    x16_t* Ks = reinterpret_cast<x16_t*>(smem);
    x16_t* Vt = reinterpret_cast<x16_t*>(smem + C::K_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    // causal: a block's work grows with its query position (it visits keys 0 .. its last token) — dispatch the late, long blocks first
    const int seg = blockIdx.z, h = blockIdx.y, tile = CAUSAL ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int q_beg = p.cu_q[seg], Lq = p.cu_q[seg + 1] - q_beg;
    const int k_beg = p.cu_k[seg], Lk = p.cu_k[seg + 1] - k_beg;
    constexpr int TQ = 64 * QR;                                   // query rows per block
    const int G = GQA ? p.group : 1;
    const int TPB = GQA ? TQ / G : TQ;                            // query tokens per block
    if (tile * TPB >= Lq) return;
    const int hk = GQA ? h : h / p.group;
    const int shift = Lk - Lq;                                    // causal: key j visible to query i iff j <= i + shift
    const int wr0 = wave * 16 * QR;                               // this wave's first row of the block
    // row r of the block → (query token, q head); rows past the block's last whole token are dead
    auto tok_of = [&](int r) { return GQA ? tile * TPB + r / G : tile * TQ + r; };
    auto head_of = [&](int r) { return GQA ? hk * G + r % G : h; };
    auto live = [&](int r) { return (!GQA || r / G < TPB) && tok_of(r) < Lq; };
    const int w_tok0 = tok_of(wr0);                               // this wave's first / last query token
    const int w_tok1 = GQA ? tile * TPB + (wr0 + 16 * QR - 1) / G : tile * TQ + wr0 + 16 * QR - 1;

    // ---- Q fragments (B operand): column = query row
    x16x8 qf[QR][C::KQ];
#pragma unroll
    for (int rb = 0; rb < QR; ++rb)
#pragma unroll
        for (int kk = 0; kk < C::KQ; ++kk) {
            const int r = wr0 + rb * 16 + frow;
            const int qrow = tok_of(r), hq = head_of(r);
            const int d = kk * 32 + fq * 8;
            qf[rb][kk] = (live(r) && d < D) ? ld_frag(p.q + (long)(q_beg + qrow) * p.ldq + hq * D + d) : zero_frag();
            if (ROPE && qrow < Lq && d < D) {
                const bool lo = d < HALF;
                const x16_t* qp = p.q + (long)(q_beg + qrow) * p.ldq + h * D + (lo ? d + HALF : d - HALF);
                const long ci = (long)(q_beg + qrow) * p.ld_cs + (lo ? d : d - HALF);
                float x[8], y[8], c[8], sn[8];
                unpack8(__builtin_bit_cast(u32x4, qf[rb][kk]), x);
                unpack8(*reinterpret_cast<const u32x4*>(qp), y);
                *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(p.rcos + ci);
                *reinterpret_cast<f32x4*>(c + 4) = *reinterpret_cast<const f32x4*>(p.rcos + ci + 4);
                *reinterpret_cast<f32x4*>(sn) = *reinterpret_cast<const f32x4*>(p.rsin + ci);
                *reinterpret_cast<f32x4*>(sn + 4) = *reinterpret_cast<const f32x4*>(p.rsin + ci + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = lo ? rope_lo(x[e], y[e], c[e], sn[e]) : rope_hi(y[e], x[e], c[e], sn[e]);
                qf[rb][kk] = __builtin_bit_cast(x16x8, pack8(x));
            }
        }

    f32x4 o[QR][C::NB];
    float m_run[QR], l_run[QR];
#pragma unroll
    for (int rb = 0; rb < QR; ++rb) {
        m_run[rb] = -INFINITY;
        l_run[rb] = 0.f;
#pragma unroll
        for (int i = 0; i < C::NB; ++i) o[rb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int nkt = (Lk + 63) / 64;
    if (CAUSAL) {
        const int last = tile * TPB + TPB - 1 + shift;            // last visible key for this block
        int lim = last < 0 ? 0 : (last / 64 + 1);
        nkt = lim < nkt ? lim : nkt;
    }

    // DMA (every variant but the fused-RoPE one): K / V tile kt+1 goes global → LDS by LDS-DMA into the other half of a double
    // buffer while tile kt is multiplied — no staging registers (24-32 VGPRs back: one more wave per SIMD), no ds_write pass, ONE block
    // barrier per tile instead of two.  Lane l of piece pc owns LDS bytes pc*1024 + 16 l of the row-major tile (row stride KROW): it
    // loads the matching 16 bytes of its key row, or nothing when those bytes are row padding.  Keys past Lk re-read the last key
    // (their scores are masked, their probabilities 0).
    constexpr bool DMA = !ROPE;
    constexpr int PPW = (C::PIECES + 3) / 4;                      // pieces per wave
    auto issue = [&](int kt, int buf) {
        char* kb_ = smem + buf * C::LDS;
        char* vb_ = kb_ + C::K_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + 4 * i;
            const int ob = pc * 1024 + lane * 16;                 // byte offset inside the tile
            const int row = ob / (C::KROW * 2), cb = ob % (C::KROW * 2);
            if (pc < C::PIECES && row < 64 && cb < D * 2) {
                int key = kt * 64 + row;
                key = key < Lk ? key : Lk - 1;
                const char* ks = reinterpret_cast<const char*>(p.k + (long)(k_beg + key) * p.ldk + hk * D) + cb;
                const char* vs = reinterpret_cast<const char*>(p.v + (long)(k_beg + key) * p.ldv + hk * D) + cb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks,
                                                 (__attribute__((address_space(3))) void*)(kb_ + pc * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs,
                                                 (__attribute__((address_space(3))) void*)(vb_ + pc * 1024), 16, 0, 0);
            }
        }
    };
    // (fused-RoPE variant) K/V tile kt+1 is fetched into registers while tile kt is multiplied
    u32x4 kreg[DMA ? 1 : C::NK], vreg[DMA ? 1 : C::NK];
    constexpr int NR = ROPE ? (64 * (C::CPR / 2) + 255) / 256 : 1;  // (row, half-chunk) rotation items per thread per tile
    f32x4 rc[NR][2], rs[NR][2];
    auto fetch = [&](int kt) {
        if (ROPE) {
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int idx = it * 256 + tid;
                const int row = idx / (C::CPR / 2), c = idx % (C::CPR / 2);
                const int key = kt * 64 + row;
                const bool ok = (idx < 64 * (C::CPR / 2)) && (key < Lk);
                const long ci = (long)(k_beg + (ok ? key : 0)) * p.ld_cs + c * 8;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    rc[it][e] = ok ? *reinterpret_cast<const f32x4*>(p.rcos + ci + e * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
                    rs[it][e] = ok ? *reinterpret_cast<const f32x4*>(p.rsin + ci + e * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#pragma unroll
        for (int it = 0; it < (DMA ? 0 : C::NK); ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / C::CPR, c = idx % C::CPR;
            const int key = kt * 64 + row;
            const bool ok = (idx < 64 * C::CPR) && (key < Lk);
            kreg[it] = ok ? *reinterpret_cast<const u32x4*>(p.k + (long)(k_beg + key) * p.ldk + hk * D + c * 8) : u32x4{0u, 0u, 0u, 0u};
            vreg[it] = ok ? *reinterpret_cast<const u32x4*>(p.v + (long)(k_beg + key) * p.ldv + hk * D + c * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    if (nkt > 0) { if (DMA) issue(0, 0); else fetch(0); }

    for (int kt = 0; kt < nkt; ++kt) {
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed
            __syncthreads();                                      // everyone's have; everyone is done with the other buffer (tile kt - 1)
            if (kt + 1 < nkt) issue(kt + 1, (kt + 1) & 1);
            Ks = reinterpret_cast<x16_t*>(smem + (kt & 1) * C::LDS);
            Vt = reinterpret_cast<x16_t*>(smem + (kt & 1) * C::LDS + C::K_BYTES);
        } else {
        __syncthreads();                                          // previous tile fully consumed
#pragma unroll
        for (int it = 0; it < C::NK; ++it) {
            const int idx = it * 256 + tid;
            if (idx < 64 * C::CPR) {
                *reinterpret_cast<u32x4*>(Ks + (idx / C::CPR) * C::KROW + (idx % C::CPR) * 8) = kreg[it];
                *reinterpret_cast<u32x4*>(Vt + (idx / C::CPR) * C::VROW + (idx % C::CPR) * 8) = vreg[it];
            }
        }
        __syncthreads();
        }
        if (ROPE) {                                               // rotate the staged K tile in place (pairs d, d + D/2)
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int idx = it * 256 + tid;
                if (idx < 64 * (C::CPR / 2)) {
                    const int row = idx / (C::CPR / 2), c = idx % (C::CPR / 2);
                    x16_t* k1 = Ks + row * C::KROW + c * 8;
                    float x1[8], x2[8], o1[8], o2[8];
                    unpack8(*reinterpret_cast<const u32x4*>(k1), x1);
                    unpack8(*reinterpret_cast<const u32x4*>(k1 + HALF), x2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float cc = rc[it][e >> 2][e & 3], ss = rs[it][e >> 2][e & 3];
                        o1[e] = rope_lo(x1[e], x2[e], cc, ss);
                        o2[e] = rope_hi(x1[e], x2[e], cc, ss);
                    }
                    *reinterpret_cast<u32x4*>(k1) = pack8(o1);
                    *reinterpret_cast<u32x4*>(k1 + HALF) = pack8(o2);
                }
            }
            __syncthreads();
        }
        if (!DMA && kt + 1 < nkt) fetch(kt + 1);
        if (CAUSAL && kt * 64 > w_tok1 + shift) continue;         // tile entirely above this wave's diagonal

        // ---- S^T = K Q^T: lane holds S[query = frow][key = kt*64 + kb*16 + fq*4 + r]
        f32x4 s[QR][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rb = 0; rb < QR; ++rb) s[rb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < C::KQ; ++kk) {
                const int d = kk * 32 + fq * 8;
                const x16x8 kf = (d < D) ? ld_frag(Ks + (kb * 16 + frow) * C::KROW + d) : zero_frag();
#pragma unroll
                for (int rb = 0; rb < QR; ++rb) s[rb][kb] = mfma16(kf, qf[rb][kk], s[rb][kb]);
            }
        }
        const bool edge = (kt * 64 + 64 > Lk) || (CAUSAL && kt * 64 + 63 > w_tok0 + shift);
        if (edge) {
#pragma unroll
            for (int rb = 0; rb < QR; ++rb) {
                const int qi = tok_of(wr0 + rb * 16 + frow);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + kb * 16 + fq * 4 + r;
                        bool ok = key < Lk;
                        if (CAUSAL) ok = ok && (key <= qi + shift);
                        s[rb][kb][r] = ok ? s[rb][kb][r] : -INFINITY;
                    }
            }
        }
        x16x8 pf[QR][2];
#pragma unroll
        for (int rb = 0; rb < QR; ++rb) {
            float m = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, s[rb][kb][r]);
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float mnew = fmaxf(m_run[rb], m);
            const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
            const float alpha = __builtin_amdgcn_exp2f((m_run[rb] - msafe) * p.scale_log2);    // m_run = -inf → 0
            m_run[rb] = mnew;
            const float mc = -msafe * p.scale_log2;
            float lsum = 0.f;
            float pv[16];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[rb][kb][r], p.scale_log2, mc));   // masked → 0
                    pv[kb * 4 + r] = e;
                    lsum += e;
                }
            l_run[rb] = l_run[rb] * alpha + lsum;                  // per-lane partial (this lane's 16 keys per tile)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                pf[rb][ks] = __builtin_bit_cast(x16x8, pack8(pv + ks * 8));
            // the running max rarely moves after the first tiles: skip the O rescale for the whole wave when alpha == 1
            // in every lane (multiplying by exactly 1.0f is the identity, so results do not depend on the shortcut)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int i = 0; i < C::NB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[rb][i][r] *= alpha;
            }
        }

        // ---- O^T += V^T P^T   (A = V^T rows d, B = P; contraction slots j<4: key 32ks+4fq+j, j>=4: key 32ks+16+4fq+j-4)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < C::NB; ++i) {
                // chunk handed in by this lane: key 32ks + 4fq + (frow >> 2) (second read: + 16), d = 16i + 4(frow & 3)
                const x16_t* vp = Vt + (ks * 32 + fq * 4 + (frow >> 2)) * C::VROW + i * 16 + (frow & 3) * 4;
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_lds*)(vp)));
                const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_lds*)(vp + 16 * C::VROW)));
                const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
                const x16x8 vf = __builtin_bit_cast(x16x8, vv);
#pragma unroll
                for (int rb = 0; rb < QR; ++rb) o[rb][i] = mfma16(vf, pf[rb][ks], o[rb][i]);
            }
    }

    // ---- normalise and store: lane holds O[query = frow][d = i*16 + fq*4 + r]
#pragma unroll
    for (int rb = 0; rb < QR; ++rb) {
        float l = l_run[rb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int r = wr0 + rb * 16 + frow;
        const int qi = tok_of(r);
        if (!live(r)) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        x16_t* dst = p.o + (long)(q_beg + qi) * p.ldo + head_of(r) * D + fq * 4;
#pragma unroll
        for (int i = 0; i < C::NB; ++i) {
            u32x2 w;
            w[0] = pack2x(o[rb][i][0] * inv, o[rb][i][1] * inv);
            w[1] = pack2x(o[rb][i][2] * inv, o[rb][i][3] * inv);
            *reinterpret_cast<u32x2*>(dst + i * 16) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode attention, head_dim D (multiple of 32), one new token per sample.
struct DecodeArgs {
    const x16_t* q;        // [B][Hq*D]
    const x16_t* kc;       // K cache [B][Hkv][S_max][D]
    const x16_t* vtc;      // V^T cache [B][Hkv][D][S_max]
    const int* lens;        // [B] number of valid keys (including the token just appended)
    float* part_o;          // [B][Hkv][nsplit][16][D]
    float* part_ml;         // [B][Hkv][nsplit][16][2]
    x16_t* out;            // [B][Hq*D]
    int Hq, Hkv, S_max, nsplit;
    float scale_log2;
    int out_packed = 0;     // out in the 16-row fragment-packed activation layout (row length Hq*D)
};

// decode_attn_kernel and decode_attn_rope_kernel run the SAME softmax arithmetic (the fused kernel is what a decode step launches, the
// unfused pair llm_qkv_post + decode_attn is its test reference).  `#pragma clang fp contract(off)` pins `s * scale` and the later `x - m` as
// two roundings in both (hipcc is otherwise free to contract one of them).  K / V appends and the attention outputs are bit-identical between
// the two paths in both operand types.  (Until the end of round 4 the fp16 instantiation differed in a fraction of a per cent of the outputs
// at D = 128: the fused kernel's scalar rotation was compiled to v_fma_mixlo_f16 — one rounding — where llm_qkv_post's vector path rounds
// twice; see rounded32() in common.h.)  Only the fused kernel runs in a decode step, so no product invariant (merged == un-merged decode,
// graph == eager) ever depended on it.
template <int D>
__global__ __launch_bounds__(64) void decode_attn_kernel(DecodeArgs p) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) x16_t Pw[16 * 72];
    const int lane = threadIdx.x, frow = lane & 15, fq = lane >> 4;
    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int group = p.Hq / p.Hkv;
    const int len = p.lens[b];
    const int k0 = split * 64;
    const long pbase = (((long)b * p.Hkv + g) * p.nsplit + split) * 16;
    constexpr int KQ = D / 32, NB = D / 16;

    if (k0 >= len) {                                              // empty split: neutral partial
        if (lane < 16) { p.part_ml[(pbase + lane) * 2] = -INFINITY; p.part_ml[(pbase + lane) * 2 + 1] = 0.f; }
        return;
    }
    x16x8 qf[KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk)
        qf[kk] = (frow < group) ? ld_frag(p.q + (long)b * p.Hq * D + (g * group + frow) * D + kk * 32 + fq * 8) : zero_frag();

    const x16_t* kbase = p.kc + ((long)b * p.Hkv + g) * p.S_max * D;
    const x16_t* vbase = p.vtc + ((long)b * p.Hkv + g) * D * (long)p.S_max;

    f32x4 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        int key = k0 + kb * 16 + frow;
        key = key < p.S_max ? key : p.S_max - 1;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) s[kb] = mfma16(qf[kk], ld_frag(kbase + (long)key * D + kk * 32 + fq * 8), s[kb]);
    }
    float mrow[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int key = k0 + kb * 16 + frow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = (key < len) ? s[kb][r] * p.scale_log2 : -INFINITY;
            s[kb][r] = x;
            mrow[r] = fmaxf(mrow[r], x);
        }
    }
    float lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m = mrow[r];
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        m = fmaxf(m, __shfl_xor(m, 8, 64));
        mrow[r] = m;                                              // finite: key k0 < len exists
        lrow[r] = 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = exp2f(s[kb][r] - mrow[r]);
            lrow[r] += pv;
            Pw[(fq * 4 + r) * 72 + kb * 16 + frow] = f2x(pv);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = lrow[r];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        lrow[r] = t;
    }
    __syncthreads();
    f32x4 o[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const x16x8 pf = ld_frag(Pw + frow * 72 + ks * 32 + fq * 8);
        int kk0 = k0 + ks * 32 + fq * 8;                          // 8 keys, 16-byte aligned in the V^T row
        kk0 = (kk0 + 8 <= p.S_max) ? kk0 : p.S_max - 8;           // (S_max % 64 == 0, so this never actually clamps)
#pragma unroll
        for (int i = 0; i < NB; ++i) o[i] = mfma16(pf, ld_frag(vbase + (long)(i * 16 + frow) * p.S_max + kk0), o[i]);
    }
    // partials: lane holds O[head = fq*4 + r][d = i*16 + frow]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hrow = fq * 4 + r;
        float* po = p.part_o + (pbase + hrow) * D;
#pragma unroll
        for (int i = 0; i < NB; ++i) po[i * 16 + frow] = o[i][r];
        if (frow == 0) { p.part_ml[(pbase + hrow) * 2] = mrow[r]; p.part_ml[(pbase + hrow) * 2 + 1] = lrow[r]; }
    }
}

template <int D>
__global__ void decode_combine_kernel(DecodeArgs p) {
#pragma clang fp contract(off)
    const int b = blockIdx.y, hq = blockIdx.x, d = threadIdx.x;
    const int group = p.Hq / p.Hkv;
    const int g = hq / group, hrow = hq % group;
    const long base0 = (((long)b * p.Hkv + g) * p.nsplit) * 16 + hrow;
    // loads of a chunk of splits are issued together (independent), then consumed in split order: same arithmetic and order
    // as a plain loop, two memory round trips per 8 splits instead of one per split
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
            ov[i] = (mv[i] == -INFINITY) ? 0.f : p.part_o[base * D + d];      // empty splits never wrote their O partial
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (mv[i] == -INFINITY) continue;
            const float w = exp2f(mv[i] - M);
            L += w * lv[i];
            acc += w * ov[i];
        }
    }
    const long ld = (long)p.Hq * D;
    const int n = hq * D + d;
    const long o = p.out_packed ? (long)(b >> 4) * 16 * ld + ((long)(n >> 3) * 16 + (b & 15)) * 8 + (n & 7) : (long)b * ld + n;
    p.out[o] = f2x(L > 0.f ? acc / L : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode-step attention with mRoPE + KV-cache append fused in (replaces llm_qkv_post + decode_attn for T = 1).
// Grid (64-key split, kv head, sample), one wave per block — same decomposition as decode_attn_kernel, so a layer's
// ≈5 MB of KV is spread over ≈160 CUs.  Every block rotates the group's q heads (8 x 128 values) and the new k with the
// step's cos/sin table (written once per step by greedy_step_kernel, fp32, [B][D/2][2]) into LDS; the block whose split
// contains the append slot also writes k / v to the K / V^T caches.  The fresh token's K row and V^T column are always
// taken from LDS, never read back from global, so no block depends on another block's (or its own) fresh stores.
// HF:557-599 (mRoPE), :665-666 (cache update), :641-689 (attention with Lq == 1).
struct DecodeRopeArgs {
    const x16_t* qkv; long ld_qkv;   // [B][(Hq + 2 Hkv) * D], bias already added
    const float* rope_cs;             // [B][D/2][2] cos, sin of this step's position (sections already resolved)
    const int* slot;                  // [B] append index; valid keys afterwards = slot + 1
    x16_t* kc; x16_t* vtc;
    float* part_o; float* part_ml;
    int B, Hq, Hkv, S_max, nsplit;
    float scale_log2;
};

template <int D>
__global__ __launch_bounds__(64) void decode_attn_rope_kernel(DecodeRopeArgs p) {
#pragma clang fp contract(off)
    constexpr int QROW = D + 8, KQ = D / 32, NB = D / 16, HALF = D / 2;
    __shared__ __attribute__((aligned(16))) x16_t Qs[16 * QROW];
    __shared__ __attribute__((aligned(16))) x16_t Knew[D];
    __shared__ __attribute__((aligned(16))) x16_t Vnew[D];
    __shared__ __attribute__((aligned(16))) x16_t Pw[16 * 72];
    __shared__ __attribute__((aligned(16))) x16_t Raw[18 * D];
    __shared__ __attribute__((aligned(16))) float Cs[D];
    const int lane = threadIdx.x, frow = lane & 15, fq = lane >> 4;
    const int split = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int group = p.Hq / p.Hkv;
    const int slot = p.slot[b];
    const int len = slot + 1;
    const int k0 = split * 64;
    const long pbase = (((long)b * p.Hkv + g) * p.nsplit + split) * 16;
    if (k0 >= len) {
        if (lane < 16) { p.part_ml[(pbase + lane) * 2] = -INFINITY; p.part_ml[(pbase + lane) * 2 + 1] = 0.f; }
        return;
    }
    const x16_t* row = p.qkv + (long)b * p.ld_qkv;
    x16_t* kbase = p.kc + ((long)b * p.Hkv + g) * p.S_max * D;
    x16_t* vbase = p.vtc + ((long)b * p.Hkv + g) * D * (long)p.S_max;
    const bool owner = (slot >= k0) && (slot < k0 + 64);
    const float* cs = p.rope_cs + (long)b * HALF * 2;

    // ---- all global loads of the block are issued back to back, then consumed: (1) this (b, g)'s raw q/k/v slice and
    //      the step's cos/sin table (→ LDS staging), (2) every cache fragment of the split (K: 4 key blocks x KQ,
    //      V^T: 2 k-steps x NB → registers).  One memory round trip instead of four dependent ones.  The fresh token's
    //      K row / V^T column is patched from LDS afterwards, so reading the (stale) slot here is harmless.
    constexpr int CPH = D / 8;                                    // 16-byte chunks per head
    const int n_raw = (group + 2) * CPH;                          // q heads of the group, then k, then v
    constexpr int RAW_IT = (18 * CPH + 63) / 64;
    u32x4 raw[RAW_IT];
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
        const int c = it * 64 + lane;
        const int hh = c / CPH, cc = c % CPH;
        const x16_t* src = hh < group ? row + (long)(g * group + hh) * D
                                       : (hh == group ? row + (long)(p.Hq + g) * D : row + (long)(p.Hq + p.Hkv + g) * D);
        raw[it] = (c < n_raw) ? *reinterpret_cast<const u32x4*>(src + cc * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    float2 csv[(HALF + 63) / 64];
#pragma unroll
    for (int it = 0; it < (HALF + 63) / 64; ++it) {
        const int d = it * 64 + lane;
        csv[it] = d < HALF ? *reinterpret_cast<const float2*>(cs + 2 * d) : float2{1.f, 0.f};
    }
    u32x4 kraw[4][KQ], vraw[2][NB];
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
        for (int i = 0; i < NB; ++i)
            vraw[ks][i] = *reinterpret_cast<const u32x4*>(vbase + (long)(i * 16 + frow) * p.S_max + k0 + ks * 32 + fq * 8);

    for (int i = lane; i < 16 * QROW; i += 64) Qs[i] = 0;
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
        const int c = it * 64 + lane;
        if (c < n_raw) *reinterpret_cast<u32x4*>(Raw + c * 8) = raw[it];
    }
#pragma unroll
    for (int it = 0; it < (HALF + 63) / 64; ++it) {
        const int d = it * 64 + lane;
        if (d < HALF) { Cs[2 * d] = csv[it].x; Cs[2 * d + 1] = csv[it].y; }
    }
    __syncthreads();
    for (int i = lane; i < (group + 1) * HALF; i += 64) {
        const int hh = i / HALF, d = i % HALF;
        const float c = Cs[2 * d], sn = Cs[2 * d + 1];
        const float x1 = x2f(Raw[hh * D + d]), x2 = x2f(Raw[hh * D + d + HALF]);
        const x16_t o1 = f2x(rounded32(rope_lo(x1, x2, c, sn))), o2 = f2x(rounded32(rope_hi(x1, x2, c, sn)));
        if (hh < group) {
            Qs[hh * QROW + d] = o1; Qs[hh * QROW + d + HALF] = o2;
        } else {
            Knew[d] = o1; Knew[d + HALF] = o2;
            if (owner) { kbase[(long)slot * D + d] = o1; kbase[(long)slot * D + d + HALF] = o2; }
        }
    }
    for (int d = lane; d < D; d += 64) {
        const x16_t v = Raw[(group + 1) * D + d];
        Vnew[d] = v;
        if (owner) vbase[(long)d * p.S_max + slot] = v;
    }
    __syncthreads();

    x16x8 qf[KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) qf[kk] = ld_frag(Qs + frow * QROW + kk * 32 + fq * 8);
    f32x4 s[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int key = k0 + kb * 16 + frow;
#pragma unroll
        for (int kk = 0; kk < KQ; ++kk) {
            const x16x8 kn = ld_frag(Knew + kk * 32 + fq * 8);
            const x16x8 kf = (key == slot) ? kn : __builtin_bit_cast(x16x8, kraw[kb][kk]);
            s[kb] = mfma16(qf[kk], kf, s[kb]);
        }
    }
    float mrow[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int key = k0 + kb * 16 + frow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = (key < len) ? s[kb][r] * p.scale_log2 : -INFINITY;
            s[kb][r] = x;
            mrow[r] = fmaxf(mrow[r], x);
        }
    }
    float lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m = mrow[r];
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        m = fmaxf(m, __shfl_xor(m, 8, 64));
        mrow[r] = m;
        lrow[r] = 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pv = exp2f(s[kb][r] - mrow[r]);
            lrow[r] += pv;
            Pw[(fq * 4 + r) * 72 + kb * 16 + frow] = f2x(pv);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = lrow[r];
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        lrow[r] = t;
    }
    __syncthreads();
    f32x4 o[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const x16x8 pf = ld_frag(Pw + frow * 72 + ks * 32 + fq * 8);
        const int kk0 = k0 + ks * 32 + fq * 8;                     // 8 consecutive keys of the V^T row
        const int j = slot - kk0;                                  // position of the fresh token inside this fragment
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int d = i * 16 + frow;
            u32x4 vv = vraw[ks][i];
            if (j >= 0 && j < 8) {
                const unsigned nv = Vnew[d];
                const int wsel = j >> 1;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) {
                    const unsigned cur = vv[w2];
                    const unsigned pat = (j & 1) ? ((cur & 0x0000ffffu) | (nv << 16)) : ((cur & 0xffff0000u) | nv);   // 16-bit lane insert: type-agnostic
                    vv[w2] = (w2 == wsel) ? pat : cur;
                }
            }
            o[i] = mfma16(pf, __builtin_bit_cast(x16x8, vv), o[i]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hrow = fq * 4 + r;
        float* po = p.part_o + (pbase + hrow) * D;
#pragma unroll
        for (int i = 0; i < NB; ++i) po[i * 16 + frow] = o[i][r];
        if (frow == 0) { p.part_ml[(pbase + hrow) * 2] = mrow[r]; p.part_ml[(pbase + hrow) * 2 + 1] = lrow[r]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode-step attention over FRAGMENT-PACKED caches, ONE launch (round 6).  The two-launch form above spends 18.7 us per layer at 64 rows
// (8.9 us of it the latency floor of two dependent launches; 10.5 MB of fp32 partials written and 5 MB re-read); packing the caches
// alone bought 1.13-1.23x (round 4, tools/ubench/decode_attn_v2.hip), a single launch on the row-major caches nothing (one CU cannot
// issue a sample's 307 KB of KV as 64-byte row pieces fast enough: 19.0 us) — together: 11.2 us at 64 rows (1.68x), 18.5 at 128 (1.73x),
// 26.3 at 128 rows x 951 keys (1.57x), 18.9 for PaDT_Pro_7B's 28:4 heads at 64 rows (1.70x); profiles/r06_decode_attn_v3.log.
// Cache images (every wave-wide load instruction reads 1 KiB of contiguous memory, as the packed weight image of csrc/gemm.hip):
//   K   [B][Hkv][S_max/16][D/32][64 lanes][8]: lane (frow, fq) of tile (s16, kk) holds K[16 s16 + frow][32 kk + 8 fq .. + 8)
//   V^T [B][Hkv][D/16][S_max/32][64 lanes][8]: lane (frow, fq) of tile (i, ks) holds V[32 ks + 4 fq + e][16 i + frow], e < 4, then the
//       same four keys + 16 — the key permutation under which the transposed scores' registers ARE the second MFMA's B operand.
// padt_llm_qkv_post writes the same images (cache_packed = 1).  A block = NW waves owns one (kv head, sample): wave w streams the 64-key
// splits w, w + NW, ... with a register-only body (q and the fresh k rotated in registers straight into fragments, S^T = K Q^T so a lane
// owns one head's scores: in-lane soft-max + two shuffles, probabilities fed to O^T = V^T P^T from registers) into a running (m, l, O^T)
// — no per-split partial leaves the wave —, the NW states meet in LDS and every wave finishes its share of the d-tiles and stores 16-bit
// rows.  Per-sample results do not depend on the batch (a block sees one sample) nor on S_max (splits sit at absolute key positions).
// The merge arithmetic differs from decode_combine's, so outputs are equal to the two-launch form's or one 16-bit rounding apart.
struct DecodePackedArgs {
    const x16_t* qkv; long ld_qkv;   // [B][(Hq + 2 Hkv) * D], bias already added
    const float* rope_cs;             // [B][D/2][2] cos, sin of this step's position
    const int* slot;                  // [B] append index; valid keys afterwards = slot + 1
    x16_t* kc; x16_t* vtc;           // packed images (above)
    x16_t* out;                      // [B][Hq * D] rows (or the 16-row fragment-packed activation layout)
    int B, Hq, Hkv, S_max;
    float scale_log2;
    int out_packed;
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

// DS (round 6): with Hkv x B <= 128 blocks only half of the 256 CUs stream KV (64 rows x 2 kv heads: 13.3 us for 38.7 MB, of which ≈7 us are the launch's
// fixed chain slot → q / k → rope → first split → merge).  DS = 2 gives a (kv head, sample) to TWO blocks (blockIdx.z): both compute the scores and the soft-max
// over ALL keys (K is read twice — the pair runs concurrently on one XCD, the second read hits its L2), each multiplies HALF of the d-tiles (its half of
// V^T) and stores that half of the output columns; block 0 appends k, each block the v rows of its half.  No hand-off between the two (the fresh k / v come
// from registers in both); every output bit is the DS = 1 kernel's.  Measured (8 rotating cache sets, graph replays): 13.3 → 12.0 us at 64 rows, 8.6 → 7.5 at
// 32, 8.1 → 6.9 at 8; at 128 rows (256 blocks already) 19.6 → 26.7: not taken there.  (Requesting the first split's fragments ahead of the rope prologue
// was measured too: 256 VGPRs at DS = 1, 16.7 us — dropped.)
template <int D, int NW, bool PACKED, int DS = 1>
__global__ __launch_bounds__(NW * 64) void decode_attn_rope_packed_kernel(DecodePackedArgs p) {
#pragma clang fp contract(off)
    static_assert(D == 128, "fragment map below is written for 16 chunks per head");
    constexpr int KQ = D / 32, NB = D / 16 / DS, HALF = D / 2;     // NB: the d-tiles of THIS block
    const int i0 = (DS > 1 ? (int)blockIdx.z : 0) * NB;           // its first d-tile
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
                        vfr[ks][i] = *reinterpret_cast<const u32x4*>(vbase + ((long)(i0 + i) * (p.S_max >> 5) + (k0 >> 5) + ks) * 512 + lane * 8);
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
                        const x16_t* vr = vbase + (long)((i0 + i) * 16 + frow) * p.S_max + k0 + ks * 32 + fq * 4;
                        const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr), v1 = *reinterpret_cast<const u32x2*>(vr + 16);
                        vfr[ks][i] = u32x4{v0[0], v0[1], v1[0], v1[1]};
                    }
            }
            unsigned vnew[NB];
            if (owner) {
                unsigned vapp[2];
#pragma unroll
                for (int i = 0; i < NB; ++i) vnew[i] = vrow[(i0 + i) * 16 + frow];
                vapp[0] = vrow[lane];
                vapp[1] = vrow[lane + 64];
                const bool app_k = DS == 1 || blockIdx.z == 0;    // DS = 2: block 0 appends k, block z the v rows d in [64 z, 64 z + 64)
                if constexpr (PACKED) {
                    if (frow == 0 && app_k) {
#pragma unroll
                        for (int kk = 0; kk < KQ; ++kk)
                            *reinterpret_cast<u32x4*>(kbase + ((((long)(slot >> 4) * KQ + kk) * 64) + fq * 16 + (slot & 15)) * 8) = kn[kk];
                    }
                    const int r32 = slot & 31;
                    const long vcol = ((long)(slot >> 5) * 64 + ((r32 & 15) >> 2) * 16) * 8 + (r32 >> 4) * 4 + (r32 & 3);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = lane + 64 * h;
                        if (DS == 1 || (int)blockIdx.z == h) vbase[(long)(d >> 4) * (p.S_max >> 5) * 512 + vcol + (d & 15) * 8] = (x16_t)vapp[h];
                    }
                } else {
                    if (frow == 0 && app_k) {
#pragma unroll
                        for (int kk = 0; kk < KQ; ++kk) *reinterpret_cast<u32x4*>(kbase + (long)slot * D + kk * 32 + fq * 8) = kn[kk];
                    }
                    if (DS == 1 || blockIdx.z == 0) vbase[(long)lane * p.S_max + slot] = (x16_t)vapp[0];
                    if (DS == 1 || blockIdx.z == 1) vbase[(long)(lane + 64) * p.S_max + slot] = (x16_t)vapp[1];
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
            const int n = (g * group + frow) * D + (i0 + i) * 16 + fq * 4;   // 4 consecutive columns of output row b
            const long off = p.out_packed ? (long)(b >> 4) * 16 * ld + ((long)(n >> 3) * 16 + (b & 15)) * 8 + (n & 7) : (long)b * ld + n;
            *reinterpret_cast<u32x2*>(p.out + off) = u32x2{pack2x(acc[0] * inv, acc[1] * inv), pack2x(acc[2] * inv, acc[3] * inv)};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool CAUSAL, int QR, bool ROPE, bool GQA>
static void launch_attn_k(const AttnArgs& a, dim3 grid, hipStream_t s) {
    constexpr int lds = ROPE ? AttnCfg<D>::LDS : AttnCfg<D>::LDS2;             // the LDS-DMA path double-buffers the K / V tiles
    if constexpr (lds > 64 * 1024) {
        static PerDeviceOnce once;
        once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_varlen_kernel<D, CAUSAL, QR, ROPE, GQA>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    }
    hipLaunchKernelGGL((attn_varlen_kernel<D, CAUSAL, QR, ROPE, GQA>), grid, dim3(256), lds, s, a);
}

template <int D, bool CAUSAL, int QR>
static void launch_attn_qr(const AttnArgs& a, int max_seqlen_q, int H, int nseg, hipStream_t s) {
    if (a.group > 1 && a.group <= 16 * QR && a.rcos == nullptr) {                             // q heads of a kv group share the block's K / V tiles
        const int tpb = 64 * QR / a.group;
        launch_attn_k<D, CAUSAL, QR, false, true>(a, dim3((max_seqlen_q + tpb - 1) / tpb, H / a.group, nseg), s);
        return;
    }
    const int tiles = (max_seqlen_q + 64 * QR - 1) / (64 * QR);
    if constexpr (!CAUSAL && QR == 1 && D % 16 == 0) {
        if (a.rcos) {
            launch_attn_k<D, CAUSAL, QR, true, false>(a, dim3(tiles, H, nseg), s);
            return;
        }
    }
    launch_attn_k<D, CAUSAL, QR, false, false>(a, dim3(tiles, H, nseg), s);
}

// long segments: 32 query rows per wave (each K / V^T fragment read from LDS feeds two MFMAs).  Round 1 restricted this to d <= 80
// (at d = 128 the second row block cost the second resident block per CU: prefill 36 vs 43 us); with LDS-DMA staging and MFMA results
// in VGPRs the d = 128 kernel is LDS-bound at two blocks per CU either way and the wider block wins (3B prompt 42.2 → 37.8 us, 7B
// 69.3 → 55.9 us)
template <int D, bool CAUSAL>
static void launch_attn(const AttnArgs& a, int max_seqlen_q, int H, int nseg, hipStream_t s) {
    if (max_seqlen_q >= 256) launch_attn_qr<D, CAUSAL, 2>(a, max_seqlen_q, H, nseg, s);
    else launch_attn_qr<D, CAUSAL, 1>(a, max_seqlen_q, H, nseg, s);
}

extern "C" int PADT_TWIN(padt_attn_varlen)(void* stream, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                void* o, long ldo, const int* cu_q, const int* cu_k, int nseg, int max_seqlen_q,
                                int n_heads, int n_kv_heads, int head_dim, float scale, int causal, const void* rope_cos,
                                const void* rope_sin, long ld_cs) {
    if (nseg <= 0 || max_seqlen_q <= 0) return 0;
    if (rope_cos != nullptr && (rope_sin == nullptr || causal || max_seqlen_q >= 256 || (ld_cs & 3) || ((uintptr_t)rope_cos & 15) ||
                                ((uintptr_t)rope_sin & 15) || cu_q != cu_k)) {
        padt_set_error("padt_attn_varlen: fused RoPE needs self-attention (cu_q == cu_k), non-causal, max_seqlen_q < 256, 16-byte aligned fp32 tables");
        return -1;
    }
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || n_heads % n_kv_heads) {
        padt_set_error("padt_attn_varlen: strides must be multiples of 8 elements; heads % kv_heads == 0");
        return -1;
    }
    AttnArgs a{(const x16_t*)q, ldq, (const x16_t*)k, ldk, (const x16_t*)v, ldv, (x16_t*)o, ldo, cu_q, cu_k,
               n_heads / n_kv_heads, scale * 1.4426950408889634f, (const float*)rope_cos, (const float*)rope_sin, ld_cs};
    hipStream_t s = (hipStream_t)stream;
#define PADT_ATTN_CASE(DD)                                                         \
    case DD:                                                                       \
        if (causal) launch_attn<DD, true>(a, max_seqlen_q, n_heads, nseg, s);      \
        else launch_attn<DD, false>(a, max_seqlen_q, n_heads, nseg, s);            \
        break;
    switch (head_dim) {
        PADT_ATTN_CASE(32)
        PADT_ATTN_CASE(64)
        PADT_ATTN_CASE(80)
        PADT_ATTN_CASE(128)
        default: padt_set_error("padt_attn_varlen: head_dim must be 32, 64, 80 or 128"); return -1;
    }
#undef PADT_ATTN_CASE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

#if !PADT_OP16_F16   // type-independent: compiled once
extern "C" long padt_decode_attn_workspace(int batch, int n_kv_heads, int head_dim, int s_max) {
    const long nsplit = (s_max + 63) / 64;
    return (long)batch * n_kv_heads * nsplit * 16 * (head_dim + 2) * (long)sizeof(float);
}
#endif

extern "C" int PADT_TWIN(padt_decode_attn)(void* stream, const void* q, const void* k_cache, const void* vt_cache, const int* lens,
                                void* out, void* workspace, int batch, int n_heads, int n_kv_heads, int head_dim,
                                int s_max, int max_len, float scale) {
    if (batch <= 0) return 0;
    if (n_heads % n_kv_heads || n_heads / n_kv_heads > 16 || (s_max & 63) || max_len > s_max || max_len <= 0) {
        padt_set_error("padt_decode_attn: need heads/kv_heads <= 16, s_max % 64 == 0, 0 < max_len <= s_max");
        return -1;
    }
    DecodeArgs a;
    a.q = (const x16_t*)q; a.kc = (const x16_t*)k_cache; a.vtc = (const x16_t*)vt_cache; a.lens = lens;
    a.out = (x16_t*)out; a.Hq = n_heads; a.Hkv = n_kv_heads; a.S_max = s_max;
    a.nsplit = (max_len + 63) / 64;
    a.part_o = (float*)workspace;
    a.part_ml = a.part_o + (long)batch * n_kv_heads * a.nsplit * 16 * head_dim;
    a.scale_log2 = scale * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 32:
            hipLaunchKernelGGL(decode_attn_kernel<32>, dim3(a.nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
            hipLaunchKernelGGL(decode_combine_kernel<32>, dim3(n_heads, batch), dim3(32), 0, s, a);
            break;
        case 128:
            hipLaunchKernelGGL(decode_attn_kernel<128>, dim3(a.nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
            hipLaunchKernelGGL(decode_combine_kernel<128>, dim3(n_heads, batch), dim3(128), 0, s, a);
            break;
        default: padt_set_error("padt_decode_attn: head_dim must be 32 or 128"); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

extern "C" int PADT_TWIN(padt_decode_attn_rope)(void* stream, const void* qkv, long ld_qkv, const void* rope_cs, const int* slot,
                                     void* k_cache, void* vt_cache, void* out, void* workspace, int batch, int n_heads,
                                     int n_kv_heads, int head_dim, int s_max, int max_len, float scale, int out_packed, int cache_packed) {
    if (batch <= 0) return 0;
    if (n_heads % n_kv_heads || n_heads / n_kv_heads > 16 || (s_max & 63) || (ld_qkv & 7) || max_len > s_max || max_len <= 0) {
        padt_set_error("padt_decode_attn_rope: need heads/kv_heads <= 16, s_max % 64 == 0, ld_qkv % 8 == 0, 0 < max_len <= s_max");
        return -1;
    }
    if (cache_packed) {                                        // fragment-packed caches: ONE launch, no workspace (decode_attn_rope_packed_kernel)
        if (head_dim != 128 || batch > 65535) { padt_set_error("padt_decode_attn_rope: packed caches need head_dim 128 and batch <= 65535"); return -1; }
        constexpr int NW = 8;
        constexpr int lds = NW * 64 * 2 * 4 + NW * 8 * 64 * 16;
        static PerDeviceOnce once;
        once.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_rope_packed_kernel<128, NW, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_rope_packed_kernel<128, NW, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        DecodePackedArgs pa{(const x16_t*)qkv, ld_qkv, (const float*)rope_cs, slot, (x16_t*)k_cache, (x16_t*)vt_cache, (x16_t*)out,
                            batch, n_heads, n_kv_heads, s_max, scale * 1.4426950408889634f, out_packed};
        // two blocks per (kv head, sample) while one block each would leave half of the chip without a KV stream (decode_attn_rope_packed_kernel, DS);
        // cache_packed = 2 / 3 or PADT_DECODE_ATTN_DSPLIT = 0 / 1 force the choice (tests, A/B)
        static const int env = getenv("PADT_DECODE_ATTN_DSPLIT") ? atoi(getenv("PADT_DECODE_ATTN_DSPLIT")) : -1;
        const int force = cache_packed == 2 ? 0 : (cache_packed == 3 ? 1 : env);
        const bool dsplit = force >= 0 ? force != 0 : (long)n_kv_heads * batch <= 128;
        if (dsplit) hipLaunchKernelGGL((decode_attn_rope_packed_kernel<128, NW, true, 2>), dim3(n_kv_heads, batch, 2), dim3(NW * 64), lds, (hipStream_t)stream, pa);
        else hipLaunchKernelGGL((decode_attn_rope_packed_kernel<128, NW, true, 1>), dim3(n_kv_heads, batch), dim3(NW * 64), lds, (hipStream_t)stream, pa);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
        return 0;
    }
    const int nsplit = (max_len + 63) / 64;
    DecodeRopeArgs a{(const x16_t*)qkv, ld_qkv, (const float*)rope_cs, slot, (x16_t*)k_cache, (x16_t*)vt_cache,
                     (float*)workspace, nullptr, batch, n_heads, n_kv_heads, s_max, nsplit, scale * 1.4426950408889634f};
    a.part_ml = a.part_o + (long)batch * n_kv_heads * nsplit * 16 * head_dim;
    hipStream_t s = (hipStream_t)stream;
    // A fused "last split block merges" variant (agent-scope hand-off of the partials) was measured and rejected: the
    // cross-XCD partial stores / loads cost 28 us per call against 12.7 us for the two launches below.
    DecodeArgs c;                                              // the merge reads the same partial layout
    c.q = nullptr; c.kc = nullptr; c.vtc = nullptr; c.lens = nullptr; c.part_o = a.part_o; c.part_ml = a.part_ml;
    c.out = (x16_t*)out; c.Hq = n_heads; c.Hkv = n_kv_heads; c.S_max = s_max; c.nsplit = nsplit; c.scale_log2 = a.scale_log2;
    c.out_packed = out_packed;
    switch (head_dim) {
        case 32:
            hipLaunchKernelGGL(decode_attn_rope_kernel<32>, dim3(nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
            hipLaunchKernelGGL(decode_combine_kernel<32>, dim3(n_heads, batch), dim3(32), 0, s, c);
            break;
        case 128:
            hipLaunchKernelGGL(decode_attn_rope_kernel<128>, dim3(nsplit, n_kv_heads, batch), dim3(64), 0, s, a);
            hipLaunchKernelGGL(decode_combine_kernel<128>, dim3(n_heads, batch), dim3(128), 0, s, c);
            break;
        default: padt_set_error("padt_decode_attn_rope: head_dim must be 32 or 128"); return -1;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

#if !PADT_OP16_F16   // type-independent: compiled once
// fp32 cos/sin table of one decode step: rope_cs[b][d] = (cos, sin)(pos3[axis(d)][b] * inv_freq[d]), d < D/2.
__global__ void rope_table_kernel(const int* __restrict__ pos3, const float* __restrict__ inv_freq, float* __restrict__ cs,
                                  int B, int half, int sec0, int sec1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, d = i % half;
    const int axis = d < sec0 ? 0 : (d < sec0 + sec1 ? 1 : 2);
    const float ang = (float)pos3[axis * B + b] * inv_freq[d];
    cs[2 * i] = cosf(ang);
    cs[2 * i + 1] = sinf(ang);
}

extern "C" int padt_rope_table(void* stream, const int* pos3, const void* inv_freq, void* rope_cs, int batch, int head_dim,
                               int sec0, int sec1) {
    if (batch <= 0) return 0;
    const int n = batch * (head_dim / 2);
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, pos3,
                       (const float*)inv_freq, (float*)rope_cs, batch, head_dim / 2, sec0, sec1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
#endif

}  // namespace PADT_NS
