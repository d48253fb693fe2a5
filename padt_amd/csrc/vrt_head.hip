// VRT logit head with fused logit mask + arg-max, and the greedy-loop bookkeeping kernel.
//
// Reference (padt.py:292-301, 713-757): logits = hidden @ cat([embed_tokens|lm_head, prototypes]).T, masked_fill(-inf)
// outside [text vocab ∪ the sample's own patch rows], argmax of the last position in fp32, finished rows get pad,
// EOS clears `unfinished`.  Here the two tables are read through two base pointers (the 0.6 GB concatenation per forward
// never happens), only the last position is computed, and the logits never touch HBM unless a caller asks for them.
//
// vrt_head_kernel: block = 4 waves = 16 table rows, K interleaved over the waves (same weight-streaming scheme as the
// skinny GEMM), swapped MFMA so a lane owns 4 consecutive table rows of one sample; block-local (max, argmax) per
// sample goes to a partial buffer.  greedy_step_kernel: final reduction (ties → lowest index, as torch.argmax),
// pad/EOS bookkeeping, append to the token buffer, stash the step's last-layer hidden row for parseVRTintoCompletion
// (padt_processor.py:125), advance cache slots / rope positions / step counter — all on device, so a decode step is one
// replayable hipGraph.
#include "common.h"
#include <stdlib.h>
#include <cstdint>

extern "C" void padt_set_error(const char* msg);
extern "C" long padt_vrt_head_nblk(long vocab, long n_proto);

namespace PADT_NS {

// Generation-config slots kept in DEVICE memory so a captured decode graph does not bake them in
// (HF generation_config.json: repetition_penalty, eos_token_id list; padt.py:570-580,717,756).
struct GenCfg {
    float penalty;                     // RepetitionPenaltyLogitsProcessor: score<0 ? score*p : score/p on every id already in the row
    int eos[4];                        // up to 4 EOS ids, -1 = unused
    int do_sample;                     // 0 greedy (arg-max), 1 multinomial sampling (padt.py:740-743) by sample_token_kernel
    unsigned seed;                     // counter-based RNG key (the device step counter, row and table index are the counter)
    float temperature;                 // TemperatureLogitsWarper
    int top_k;                         // TopKLogitsWarper (0 = off)
    float top_p;                       // TopPLogitsWarper (1 = off)
    int pad[2];
};

struct HeadArgs {
    const x16_t* h; long ldh;         // [B][D]
    const x16_t* E; int V;            // text rows
    const x16_t* proto; int NP;       // prototype rows
    const int* vrt_off;                // [B+1]
    const int* mode_table;             // [T] or null: 0 free, 1 text rows only, 2 own VRT rows only, 3 force EOS
    const int* step;                   // device step counter (index into mode_table) or null
    float* logits; long ldl;           // optional [B][V+NP]
    float* part_val; int* part_idx;    // [nblk][16*MT]
    int B, D, eos;
    const x16_t* Ep;                  // optional fragment-packed copy of E (PACKED kernels; h is then packed too)
    const GenCfg* gen;                 // optional generation config (repetition penalty) + per-sample seen-token bitmap
    const unsigned* seen; long seen_words;
};

// PACKED: text rows come from a fragment-packed copy of the table ([V/16][D/32][64 lanes][8], ops.pack_weight — every wave
// load is 1 KiB contiguous) and the hidden rows from the 16-row fragment-packed activation layout; prototype rows (rebuilt
// per batch) stay row-major.  Waves take groups of U consecutive K-steps; wave w finishes (row block, sample block) pairs w, w + 4, ...
// NT: 16-row table blocks per thread block — they share every hidden-row fragment a wave loads (at 64 rows a wave loads 4 KiB of hidden
// fragments per K-step: with one table block per thread block that is 4 bytes of L2 traffic per byte of table, 2.5 GB per step; with NT = 4,
// 1:1).  The K-step → wave map and the cross-wave order are those of NT = 1: a logit's bits depend neither on NT nor on the row count.
template <int MT, int NT, bool PACKED>
__global__ __launch_bounds__(256) void vrt_head_kernel(HeadArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red_raw[];
    typedef float RedT[NT * MT][64][4];
    RedT* red = reinterpret_cast<RedT*>(red_raw);                 // [4 waves]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int NTOT = p.V + p.NP;
    const int nblk = (NTOT + 15) / 16;
    const x16_t* wrow[NT];
    const x16_t* wpk[NT];
    bool in_text[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n0 = (blockIdx.x * NT + i) * 16;
        int n = n0 + frow;
        n = n < NTOT ? n : NTOT - 1;
        in_text[i] = PACKED && (n0 + 16 <= p.V);                  // whole block inside the packed text table
        wrow[i] = (n < p.V) ? p.E + (long)n * p.D : p.proto + (long)(n - p.V) * p.D;
        wpk[i] = PACKED ? p.Ep + (long)(min(n0, p.V - 16) >> 4) * (p.D >> 5) * 512 + lane * 8 : nullptr;
    }
    const x16_t* xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = j * 16 + frow;
        xok[j] = PACKED ? (j * 16 < p.B) : (m < p.B);             // packed: row blocks past the last valid row are not in the buffer
        xrow[j] = PACKED ? p.h + (long)j * 16 * p.ldh + lane * 8 : p.h + (long)(m < p.B ? m : 0) * p.ldh + fq * 8;
    }
    const int xstep = PACKED ? 512 : 32;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nks = (p.D + 31) / 32;
    constexpr int U = 4;                                          // K-steps per wave group (fixes the summation order)
    constexpr int UH = (NT * MT > 8) ? 2 : 4;                     // K-steps loaded at a time (register budget); same MFMA order either way
    for (int g0 = wave; g0 * U < nks; g0 += 4) {
#pragma unroll
        for (int uh = 0; uh < U; uh += UH) {
            x16x8 wf[UH][NT], xf[UH][MT];
#pragma unroll
            for (int u = 0; u < UH; ++u) {
                const int ks = g0 * U + uh + u;
                const int k = ks * 32 + fq * 8;
                const bool kok = (ks < nks) && (k < p.D);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (PACKED && in_text[i]) wf[u][i] = kok ? __builtin_nontemporal_load(reinterpret_cast<const x16x8*>(wpk[i] + (long)ks * 512)) : zero_frag();
                    else wf[u][i] = kok ? ld_frag(wrow[i] + k) : zero_frag();
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) xf[u][j] = (kok && xok[j]) ? ld_frag(xrow[j] + (long)ks * xstep) : zero_frag();
            }
#pragma unroll
            for (int u = 0; u < UH; ++u)
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) acc[i][j] = mfma16(wf[u][i], xf[u][j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(&red[wave][i * MT + j][lane][0]) = acc[i][j];
    __syncthreads();
    const int mode = (p.mode_table && p.step) ? p.mode_table[*p.step] : 0;
    const float pen = (p.gen && p.seen) ? p.gen->penalty : 1.0f;
    for (int q = wave; q < NT * MT; q += 4) {                     // (table block, sample block) pairs this wave finishes
        const int i = q / MT, j = q % MT;
        const int blk = blockIdx.x * NT + i;
        if (blk >= nblk) continue;
        const int n0 = blk * 16;
        f32x4 sum = *reinterpret_cast<f32x4*>(&red[0][q][lane][0]);
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<f32x4*>(&red[w][q][lane][0]);
        const int m = j * 16 + frow;                              // sample
        float best = -INFINITY;
        int bidx = 0x7fffffff;
        int lo = 0, hi = 0;
        if (m < p.B) { lo = p.vrt_off[m]; hi = p.vrt_off[m + 1]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = n0 + fq * 4 + r;
            bool ok = (m < p.B) && (row < NTOT);
            if (ok) {
                if (row < p.V) ok = (mode == 0 || mode == 1 || (mode == 3 && row == p.eos));
                else { const int jv = row - p.V; ok = (jv >= lo && jv < hi) && (mode == 0 || mode == 2); }
            }
            float sc = sum[r];
            if (pen != 1.0f && ok && ((p.seen[(long)m * p.seen_words + (row >> 5)] >> (row & 31)) & 1u)) sc = sc < 0.f ? sc * pen : sc / pen;
            const float v = ok ? sc : -INFINITY;
            if (p.logits && m < p.B && row < NTOT) p.logits[(long)m * p.ldl + row] = v;
            if (v > best) { best = v; bidx = row; }               // rows ascend with r → first max wins
        }
        // combine the 4 lanes (fq = 0..3) that hold the same sample
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bidx, off, 64);
            if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
        }
        if (fq == 0 && m < p.B) {
            p.part_val[(long)blk * p.B + m] = best;
            p.part_idx[(long)blk * p.B + m] = bidx;
        }
    }
}

#if !PADT_OP16_F16   // type-independent: compiled once
struct GreedyArgs {
    const float* part_val; const int* part_idx; int nblk;
    int B, D, eos, pad, T_max;
    int* unfinished;          // [B]
    long* tokens_out;         // [B][T_max]
    long* cur_tok;            // [B] token to feed to the next step
    int* step;                // device counter
    int* slot; int* lens;     // [B] KV append index / valid-key count for the NEXT step
    int* pos3;                // [3][B] rope positions for the NEXT step
    const x16_t* hidden;     // [B][D] last-layer hidden of this step (post final norm)
    x16_t* hidden_buf;       // [T_max][B][D]
    int advance;              // 1: bump slot/lens/pos (decode steps and after prefill)
    const GenCfg* gen;        // optional: extra EOS ids
    unsigned* seen; long seen_words;   // optional: bitmap of ids present in each row (repetition penalty), updated here
};

__global__ __launch_bounds__(256) void greedy_step_kernel(GreedyArgs p) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < p.nblk; i += 256) {
        const float v = p.part_val[(long)i * p.B + b];
        const int ix = p.part_idx[(long)i * p.B + b];
        if (v > best || (v == best && ix < bidx)) { best = v; bidx = ix; }
    }
    sv[tid] = best; si[tid] = bidx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float ov = sv[tid + s]; const int oi = si[tid + s];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    const int step = *p.step;
    if (step < p.T_max) {
        const x16_t* h = p.hidden + (long)b * p.D;
        x16_t* hb = p.hidden_buf + ((long)step * p.B + b) * p.D;
        for (int c = tid * 8; c < p.D; c += 256 * 8) *reinterpret_cast<u32x4*>(hb + c) = *reinterpret_cast<const u32x4*>(h + c);
    }
    __syncthreads();
    if (tid == 0) {
        const int unf = p.unfinished[b];
        long next = unf ? (long)si[0] : (long)p.pad;                 // padt.py:749
        // a row whose logits are all NaN (an fp16 operand overflowed upstream: the range guard flags the batch, padt_check_finite) has no
        // arg-max — bidx is still the sentinel; continue with the pad token so that every later index stays inside its table
        if (si[0] == 0x7fffffff || next < 0 || (p.seen && (next >> 5) >= p.seen_words)) next = (long)p.pad;
        if (step < p.T_max) p.tokens_out[(long)b * p.T_max + step] = next;
        p.cur_tok[b] = next;
        bool is_eos = next == p.eos;
        if (p.gen) {
#pragma unroll
            for (int k = 0; k < 4; ++k) is_eos = is_eos || (p.gen->eos[k] >= 0 && next == (long)p.gen->eos[k]);
        }
        p.unfinished[b] = unf & (is_eos ? 0 : 1);                    // padt.py:756
        if (p.seen) p.seen[(long)b * p.seen_words + (next >> 5)] |= 1u << (next & 31);   // the row now contains `next`
        if (p.advance) {
            p.slot[b] += 1; p.lens[b] += 1;
            p.pos3[b] += 1; p.pos3[p.B + b] += 1; p.pos3[2 * p.B + b] += 1;
        }
    }
}

// one-thread kernel: bump the step counter after all samples' greedy_step blocks ran
__global__ void step_inc_kernel(int* step) { *step += 1; }

extern "C" long padt_vrt_head_nblk(long vocab, long n_proto) { return (vocab + n_proto + 15) / 16; }
#endif

template <int MT, int NT, bool PACKED>
static void launch_head(const HeadArgs& a, int nblk, hipStream_t s) {
    constexpr int lds = 4 * NT * MT * 64 * 16;
    if constexpr (lds > 64 * 1024) {
        static PerDeviceOnce once;
        once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vrt_head_kernel<MT, NT, PACKED>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
    }
    hipLaunchKernelGGL((vrt_head_kernel<MT, NT, PACKED>), dim3((nblk + NT - 1) / NT), dim3(256), lds, s, a);
}

extern "C" int PADT_TWIN(padt_vrt_head)(void* stream, const void* hidden, long ldh, const void* embed_table, long vocab,
                             const void* proto, long n_proto, const int* vrt_off, const int* mode_table,
                             const int* step, void* logits_f32, long ld_logits, void* part_val, void* part_idx,
                             long batch, long D, int eos, const void* embed_table_packed, const void* gen_cfg,
                             const void* seen, long seen_words) {
    if (batch <= 0) return 0;
    if (batch > 128 || (D & 7) || (ldh & 7)) { padt_set_error("padt_vrt_head: batch <= 128, D % 8 == 0 required"); return -1; }
    if (embed_table_packed && ((D & 31) || (vocab & 15) || ((uintptr_t)embed_table_packed & 15) || ((uintptr_t)hidden & 15))) {
        padt_set_error("padt_vrt_head: the packed path needs D % 32 == 0, vocab % 16 == 0 and 16-byte aligned pointers");
        return -1;
    }
    HeadArgs a{(const x16_t*)hidden, ldh, (const x16_t*)embed_table, (int)vocab, (const x16_t*)proto, (int)n_proto,
               vrt_off, mode_table, step, (float*)logits_f32, ld_logits, (float*)part_val, (int*)part_idx, (int)batch,
               (int)D, eos, (const x16_t*)embed_table_packed, (const GenCfg*)gen_cfg, (const unsigned*)seen, seen_words};
    if (seen && seen_words * 32 < vocab + n_proto) { padt_set_error("padt_vrt_head: seen bitmap narrower than the table"); return -1; }
    const int nblk = (int)padt_vrt_head_nblk(vocab, n_proto);
    hipStream_t s = (hipStream_t)stream;
    if (embed_table_packed) {                                   // NT = 4 table blocks per thread block (2 at 128 rows): profiles/r03_head_nt_ab.log
        if (batch <= 16) launch_head<1, 4, true>(a, nblk, s);
        else if (batch <= 32) launch_head<2, 4, true>(a, nblk, s);
        else if (batch <= 64) launch_head<4, 4, true>(a, nblk, s);
        else launch_head<8, 2, true>(a, nblk, s);
    } else if (batch <= 16) launch_head<1, 1, false>(a, nblk, s);
    else if (batch <= 32) launch_head<2, 1, false>(a, nblk, s);
    else if (batch <= 64) launch_head<4, 1, false>(a, nblk, s);
    else launch_head<8, 1, false>(a, nblk, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

#if !PADT_OP16_F16   // type-independent: compiled once
extern "C" int padt_greedy_step(void* stream, const void* part_val, const void* part_idx, long nblk, long batch, long D,
                                int eos, int pad, long t_max, int* unfinished, long* tokens_out, long* cur_tok,
                                int* step, int* slot, int* lens, int* pos3, const void* hidden, void* hidden_buf,
                                int advance, const void* gen_cfg, void* seen, long seen_words) {
    if (batch <= 0) return 0;
    if (D & 7) { padt_set_error("padt_greedy_step: D % 8 == 0 required"); return -1; }
    GreedyArgs a{(const float*)part_val, (const int*)part_idx, (int)nblk, (int)batch, (int)D, eos, pad, (int)t_max,
                 unfinished, tokens_out, cur_tok, step, slot, lens, pos3, (const x16_t*)hidden, (x16_t*)hidden_buf,
                 advance, (const GenCfg*)gen_cfg, (unsigned*)seen, seen_words};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(greedy_step_kernel, dim3((unsigned)batch), dim3(256), 0, s, a);
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Sampling branch of the loop (padt.py:740-743: probs = softmax(next_token_scores); next = multinomial(probs, 1)) with HF's
// logits warpers (generation/logits_process.py: Temperature → TopK → TopP, the order _get_logits_processor builds at padt.py:570-580).
// One block per row over the fp32 logits the head kernel wrote (mask / repetition penalty / scripted mode already applied):
//   top-k: exact k-th largest by a 4-pass radix select on order-preserving keys; everything >= it stays (HF keeps ties);
//   top-p: the survivors (<= 1024: top_p needs top_k) are sorted in LDS; rank r stays iff the probability mass of the ranks
//          before it is < top_p (= HF's "remove ascending-cumulative <= 1 - top_p", at least one token kept);
//   draw:  Gumbel-max — argmax((l - max)/T + g), g = -log(-log u), u from a counter-based hash of (seed, step, row, index) —
//          an exact multinomial draw from softmax(l/T) over the survivors without normalising or building a CDF.
// The draws cannot match torch.multinomial's (different generator); the DISTRIBUTION is what the tests check.
PADT_DEV unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
PADT_DEV float gumbel_noise(unsigned seed, unsigned step, unsigned row, unsigned idx) {
    unsigned x = hash32(idx * 0x9E3779B1u + seed);
    x = hash32(x ^ (row * 0x85EBCA77u + step * 0xC2B2AE3Du + 0x68bc21ebu));
    // 23 random bits: (x >> 9) + 0.5 is exact in fp32 (24 would round 2^24 - 0.5 up to 2^24 → u = 1 → g = +inf, a uniformly random
    // token once per 2^24 candidates: 0.9 % per step over a 152k-row table with top_k = 0) → u in [2^-24, 1 - 2^-24], g finite
    const float u = ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
    return -logf(-logf(u));
}
PADT_DEV unsigned float_key(float f) {                                    // order-preserving float → uint
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void sample_token_kernel(const float* __restrict__ logits, long ld, int n, const GenCfg* __restrict__ g,
                                                            const int* __restrict__ step, float* __restrict__ out_val,
                                                            int* __restrict__ out_idx) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sel[2];                                           // chosen bin, remaining rank
    __shared__ float sval[1024];
    __shared__ int sidx[1024];
    __shared__ float red_v[1024];
    __shared__ int red_i[1024];
    __shared__ int cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (long)b * ld;
    const float T = g->temperature > 0.f ? g->temperature : 1.f;
    const int k = g->top_k;
    const float top_p = g->top_p;
    const unsigned seed = g->seed, st = step ? (unsigned)*step : 0u;
    // ---- top-k threshold (key of the k-th largest entry), or keep everything finite
    unsigned thresh = float_key(-INFINITY) + 1u;                          // any finite value
    if (k > 0 && k < n) {
        unsigned prefix = 0u, mask = 0u, remaining = (unsigned)k;
        for (int pass = 3; pass >= 0; --pass) {
            const int shift = pass * 8;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            for (int i = tid; i < n; i += 1024) {
                const unsigned key = float_key(row[i]);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0u;
                int bin = 255;
                for (; bin > 0; --bin) {
                    if (cum + hist[bin] >= remaining) break;
                    cum += hist[bin];
                }
                sel[0] = (unsigned)bin;
                sel[1] = remaining - cum;
            }
            __syncthreads();
            prefix |= sel[0] << shift;
            mask |= 0xffu << shift;
            remaining = sel[1];
            __syncthreads();
        }
        if (prefix > thresh) thresh = prefix;                             // fewer than k finite entries: -inf stays out
    }
    // ---- row maximum over the survivors (for the exponentials)
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const float v = row[i];
        if (float_key(v) >= thresh) mx = fmaxf(mx, v);
    }
    red_v[tid] = mx;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if (tid < s2) red_v[tid] = fmaxf(red_v[tid], red_v[tid + s2]);
        __syncthreads();
    }
    mx = red_v[0];
    __syncthreads();
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if (top_p < 1.0f && k > 0 && k <= 1024) {
        // ---- nucleus over the (<= 1024) top-k survivors: gather, sort descending, prefix mass, Gumbel over the kept ranks
        if (tid == 0) cnt = 0;
        sval[tid] = -INFINITY;
        sidx[tid] = 0x7fffffff;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const float v = row[i];
            if (float_key(v) >= thresh) {
                const int slot = atomicAdd(&cnt, 1);
                if (slot < 1024) { sval[slot] = v; sidx[slot] = i; }
            }
        }
        __syncthreads();
        for (int size = 2; size <= 1024; size <<= 1)                        // bitonic sort, descending by (value, then lower index first)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int j = tid ^ stride;
                if (j > tid) {
                    const bool desc = (tid & size) == 0;
                    const float a = sval[tid], c = sval[j];
                    const int ai = sidx[tid], ci = sidx[j];
                    const bool a_first = (a > c) || (a == c && ai < ci);
                    if (a_first != desc) { sval[tid] = c; sval[j] = a; sidx[tid] = ci; sidx[j] = ai; }
                }
                __syncthreads();
            }
        const float pv = sval[tid] > -INFINITY ? expf((sval[tid] - mx) / T) : 0.f;
        red_v[tid] = pv;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                          // inclusive scan
            const float add = tid >= off ? red_v[tid - off] : 0.f;
            __syncthreads();
            red_v[tid] += add;
            __syncthreads();
        }
        const float total = red_v[1023];
        const float before = (red_v[tid] - pv) / total;                   // probability mass of the ranks before this one
        const bool keep = sval[tid] > -INFINITY && (tid == 0 || before < top_p);
        if (keep) {
            best = (sval[tid] - mx) / T + gumbel_noise(seed, st, (unsigned)b, (unsigned)sidx[tid]);
            bidx = sidx[tid];
        }
    } else {
        for (int i = tid; i < n; i += 1024) {
            const float v = row[i];
            if (float_key(v) >= thresh) {
                const float sc = (v - mx) / T + gumbel_noise(seed, st, (unsigned)b, (unsigned)i);
                if (sc > best || (sc == best && i < bidx)) { best = sc; bidx = i; }
            }
        }
    }
    __syncthreads();
    red_v[tid] = best;
    red_i[tid] = bidx;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if (tid < s2) {
            const float ov = red_v[tid + s2];
            const int oi = red_i[tid + s2];
            if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) { red_v[tid] = ov; red_i[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { out_val[b] = red_v[0]; out_idx[b] = red_i[0]; }
}

// next token per row by multinomial sampling from the warped logits; writes one (value, index) "partial" per row in the layout
// padt_greedy_step reads with nblk = 1, so the bookkeeping kernel is shared with the greedy path.
extern "C" int padt_sample_token(void* stream, const void* logits_f32, long ld_logits, long n_rows_table, const void* gen_cfg,
                                 const int* step, void* part_val, void* part_idx, long batch) {
    if (batch <= 0) return 0;
    if (gen_cfg == nullptr || n_rows_table <= 0 || n_rows_table > 0x7fffffffL) { padt_set_error("padt_sample_token: gen_cfg and a table size are required"); return -1; }
    hipLaunchKernelGGL(sample_token_kernel, dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, (const float*)logits_f32, ld_logits,
                       (int)n_rows_table, (const GenCfg*)gen_cfg, step, (float*)part_val, (int*)part_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// Arg-max of fp32 score rows (round 6: the selection of the HOOKED decode loop — caller-supplied `logits_processor`s of padt.py:717 have rewritten
// the step's score rows between the head kernel and the selection, so the head's fused arg-max partials no longer describe them): one block per
// row, ties → lowest index (torch.argmax, padt.py:745), one (value, index) partial per row in the layout padt_greedy_step reads with nblk = 1.
__global__ __launch_bounds__(1024) void argmax_rows_f32_kernel(const float* __restrict__ x, long ld, int n, float* __restrict__ out_val, int* __restrict__ out_idx) {
    __shared__ float red_v[1024];
    __shared__ int red_i[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* row = x + (long)b * ld;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < n; i += 1024) {
        const float v = row[i];
        if (v > best || (v == best && i < bidx) || (v != v && best == best)) { best = v; bidx = i; }    // a NaN wins once (torch.argmax: NaN is the maximum)
    }
    red_v[tid] = best;
    red_i[tid] = bidx;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if (tid < s2) {
            const float ov = red_v[tid + s2], cv = red_v[tid];
            const int oi = red_i[tid + s2], ci = red_i[tid];
            const bool o_nan = ov != ov, c_nan = cv != cv;
            const bool take = (o_nan && !c_nan) || (o_nan == c_nan && (ov > cv || ((ov == cv || o_nan) && oi < ci)));
            if (take) { red_v[tid] = ov; red_i[tid] = oi; }
        }
        __syncthreads();
    }
    if (tid == 0) { out_val[b] = red_v[0]; out_idx[b] = red_i[0] == 0x7fffffff ? 0 : red_i[0]; }
}

extern "C" int padt_argmax_rows_f32(void* stream, const void* scores_f32, long ld, long n_cols, void* part_val, void* part_idx, long batch) {
    if (batch <= 0) return 0;
    if (scores_f32 == nullptr || n_cols <= 0 || n_cols > 0x7fffffffL || part_val == nullptr || part_idx == nullptr) {
        padt_set_error("padt_argmax_rows_f32: score rows, a column count and the partial buffers are required");
        return -1;
    }
    hipLaunchKernelGGL(argmax_rows_f32_kernel, dim3((unsigned)batch), dim3(1024), 0, (hipStream_t)stream, (const float*)scores_f32, ld, (int)n_cols,
                       (float*)part_val, (int*)part_idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// What generate()'s synchronising half needs from the device, in ONE launch and ONE small D2H copy (round 6: replaces a torch.cat of flag
// tensors, `unfinished.any()` per chunk and isin / argmax / max over the token ring — seven ATen reduce launches and three host syncs per
// batch): out = [err, any(unfinished), nf_rows[n_rows], nf_batch[n_batch], first_eos[n_rows]] with first_eos[r] = the first step t < done whose
// token is one of the EOS ids (config eos or gen->eos[0..3]), -1 if none — the reference stops right after the step in which the last
// sequence finished (padt.py:756-757), so a batch's length is max over its rows of first_eos + 1.
struct CollectArgs {
    const int* err; const int* unfinished; const int* nf_rows; const int* nf_batch; int n_rows, n_batch;
    const long* tokens; int T_max, done, eos; const GenCfg* gen; int* out;
};
__global__ __launch_bounds__(256) void collect_summary_kernel(CollectArgs p) {
    const int tid = threadIdx.x;
    int unf = 0;
    for (int r = tid; r < p.n_rows; r += 256) {
        unf |= p.unfinished[r];
        p.out[2 + r] = p.nf_rows[r];
        int first = -1;
        const long* row = p.tokens + (long)r * p.T_max;
        for (int t = 0; t < p.done && first < 0; ++t) {
            const long tok = row[t];
            bool is_eos = tok == (long)p.eos;
            if (p.gen) {
#pragma unroll
                for (int k = 0; k < 4; ++k) is_eos = is_eos || (p.gen->eos[k] >= 0 && tok == (long)p.gen->eos[k]);
            }
            if (is_eos) first = t;
        }
        p.out[2 + p.n_rows + p.n_batch + r] = first;
    }
    for (int b = tid; b < p.n_batch; b += 256) p.out[2 + p.n_rows + b] = p.nf_batch[b];
    const int any = __syncthreads_or(unf);
    if (tid == 0) { p.out[0] = p.err ? *p.err : 0; p.out[1] = any ? 1 : 0; }
}

extern "C" int padt_collect_summary(void* stream, const int* err, const int* unfinished, const int* nf_rows, const int* nf_batch, long n_rows,
                                    long n_batch, const long* tokens, long t_max, long done, int eos, const void* gen_cfg, int* out) {
    if (n_rows <= 0 || n_batch < 0 || done < 0 || done > t_max || out == nullptr) { padt_set_error("padt_collect_summary: bad arguments"); return -1; }
    CollectArgs a{err, unfinished, nf_rows, nf_batch, (int)n_rows, (int)n_batch, tokens, (int)t_max, (int)done, eos, (const GenCfg*)gen_cfg, out};
    hipLaunchKernelGGL(collect_summary_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// sequences[b] = [input_ids[b] (L) | tokens[row0 + b][0 .. n_steps)] with the session-global VRT ids of a merged decode group shifted back to
// the batch's own (id >= vocab → id - proto_row0): the `torch.cat([input_ids, next_tokens])` of padt.py:751 for one batch of the group.
__global__ void assemble_sequences_kernel(const long* __restrict__ ids, long ld_ids, int L, const long* __restrict__ tokens, long T_max,
                                          int n_steps, long vocab, long proto_row0, long* __restrict__ out, int B) {
    const int W = L + n_steps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * W; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / W), c = (int)(i % W);
        long v;
        if (c < L) v = ids[(long)b * ld_ids + c];
        else {
            v = tokens[(long)b * T_max + (c - L)];
            if (v >= vocab) v -= proto_row0;
        }
        out[i] = v;
    }
}

extern "C" int padt_assemble_sequences(void* stream, const long* input_ids, long ld_ids, long L, const long* tokens, long t_max, long n_steps,
                                       long vocab, long proto_row0, long* out, long batch) {
    if (batch <= 0 || L + n_steps <= 0) return 0;
    if (n_steps < 0 || n_steps > t_max || L < 0) { padt_set_error("padt_assemble_sequences: bad arguments"); return -1; }
    const long n = batch * (L + n_steps);
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(assemble_sequences_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, input_ids, ld_ids, (int)L, tokens, t_max,
                       (int)n_steps, vocab, proto_row0, out, (int)batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// past_logit_mask (padt.py:196-201, returned at :794): mask[b][c] = c < vocab || vrt_off[b] <= c - vocab + off0 < vrt_off[b + 1] — one byte per
// table column; vrt_off are the session's prototype row offsets of the batch's rows, off0 = the batch's first prototype row in the session.
__global__ void logit_mask_kernel(const int* __restrict__ vrt_off, long vocab, long table_rows, long off0, unsigned char* __restrict__ out, int B) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)B * table_rows; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / table_rows);
        const long c = i % table_rows;
        out[i] = (c < vocab || (c - vocab + off0 >= vrt_off[b] && c - vocab + off0 < vrt_off[b + 1])) ? 1 : 0;
    }
}

extern "C" int padt_logit_mask(void* stream, const int* vrt_off, long vocab, long table_rows, long proto_row0, void* out_u8, long batch) {
    if (batch <= 0 || table_rows <= 0) return 0;
    const long n = batch * table_rows;
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(logit_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vrt_off, vocab, table_rows, proto_row0,
                       (unsigned char*)out_u8, (int)batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// output_scores=True (padt.py:719-720: `scores += (next_token_scores,)`): dst[*step][0..n) = src[0..n) — the masked / penalised fp32 logit rows
// the head kernel just wrote, filed under the DEVICE step counter so the copy can sit inside the captured decode graph.
__global__ void stash_step_f32_kernel(const float* __restrict__ src, long n, const int* __restrict__ step, long t_max, float* __restrict__ dst) {
    const long s = *step;
    if (s < 0 || s >= t_max) return;
    float* d = dst + s * n;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) d[n4 * 4 + threadIdx.x] = src[n4 * 4 + threadIdx.x];
}

extern "C" int padt_stash_step_f32(void* stream, const void* src_f32, long n, const int* step, long t_max, void* dst_f32) {
    if (n <= 0) return 0;
    if (((size_t)src_f32 | (size_t)dst_f32) & 15 || (n & 3)) { padt_set_error("padt_stash_step_f32: 16-byte aligned buffers and n % 4 == 0 required"); return -1; }
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(stash_step_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)src_f32, n, step, t_max, (float*)dst_f32);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

// seen[row[i]] |= bit(ids[i]) for every prompt token (ids already global in the session's table); ids outside the table are
// ignored (the range assert of padt.py:203 is reported by the embedding kernel).
__global__ void seen_init_kernel(const long* __restrict__ ids, const int* __restrict__ rows, long n, unsigned* seen, long words) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long id = ids[i];
        if (id >= 0 && id < words * 32) atomicOr(&seen[(long)rows[i] * words + (id >> 5)], 1u << (id & 31));
    }
}

extern "C" int padt_seen_init(void* stream, const long* ids, const int* rows, long n, void* seen, long seen_words) {
    if (n <= 0) return 0;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(seen_init_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ids, rows, n, (unsigned*)seen, seen_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}
#endif

}  // namespace PADT_NS
