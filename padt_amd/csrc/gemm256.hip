// 256x256x64 phase-pipelined bf16 MFMA GEMM for gfx950 (large-N shapes of ViT / prefill):
//     C[M,N] = epi(A[M,K] · W[N,K]^T + bias)          same operands / epilogues as gemm_tile_kernel (gemm.hip)
//
// Why a second tile kernel: the 128^2 two-barrier kernel drains its LDS-DMA at every K-step (s_waitcnt vmcnt(0) +
// barrier) and tops out at ≈880 TFLOP/s (profiles/r01_gemm_tile_experiments.md).  This one never drains:
//   * 8 waves (2 x 4), wave tile 128 x 64 → 64 MFMA 16x16x32 per wave per K-tile, issued as 2 PHASES of 32 (phase A: the m-half-0
//     rows against both n-halves, phase B: the m-half-1 rows) — see the hazard table at the main loop;
//   * LDS holds two K-tiles as 2 x 4 half-tiles of 16 KiB: A-half h = the m-half h rows of every wave, B-half h = the
//     n-half h columns of every wave — so a half-tile is dead after the last phase that uses it and can be re-staged
//     while the rest of its K-tile is still being multiplied;
//   * phase A stages one half-tile and phase B three half-tiles of FUTURE K-tiles (2 LDS-DMA pieces per wave per half-tile) and
//     both wait with a COUNTED s_waitcnt (vmcnt(2) / vmcnt(6)): the youngest half-tiles stay in flight across barriers;
//   * a phase = load segment | raw s_barrier | 32-MFMA segment | raw s_barrier, and the two wave groups (wr = 0 / 1,
//     i.e. the two waves of every SIMD) run ONE BARRIER APART: while one multiplies, the other stages and reads LDS;
//   * s_setprio(1) around each MFMA cluster; same source-side XOR swizzle as gemm_tile_kernel (0 bank conflicts).
// (The round-1 schedule of four 16-MFMA phases and the DMA-placement variants measured against it are in the history and in
//  profiles/r02_gemm256_experiments.md.)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

// Dispatch knobs (tests force every tile variant): read from the environment ONCE when the library is loaded; tests and tools that A/B a
// variant inside one process call padt_gemm_knobs() instead.  -1 keeps a field.  ONE object for both operand-type instantiations.
struct Knobs256 { int mode, mf, peel, colsplit, group_m; };
#if !PADT_OP16_F16
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
Knobs256 padt_g_knobs256 = {env_int("PADT_GEMM256", 1), env_int("PADT_GEMM_MF", 0), env_int("PADT_GEMM_PEEL", 1), env_int("PADT_GEMM_COLSPLIT", 1),
                            env_int("PADT_GEMM_GROUP_M", 8)};
extern "C" int padt_gemm_knobs(int mode256, int mf, int peel, int colsplit, int group_m) {
    if (mode256 >= 0) padt_g_knobs256.mode = mode256;
    if (mf >= 0) padt_g_knobs256.mf = mf;
    if (peel >= 0) padt_g_knobs256.peel = peel;
    if (colsplit >= 0) padt_g_knobs256.colsplit = colsplit;
    if (group_m >= 1) padt_g_knobs256.group_m = group_m;
    return 0;
}
#else
extern Knobs256 padt_g_knobs256;
#endif
extern "C" void padt_set_error(const char* msg);

namespace PADT_NS {
static Knobs256& g_knobs = padt_g_knobs256;

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_SWIGLU = 3 };

struct Gemm256Args {
    const x16_t* A; long lda;
    const x16_t* W; long ldw;
    const x16_t* bias;
    void* C; long ldc;
    const x16_t* R; long ldr;
    int M, N, K;
    const float* rs;                // optional per-row scale of the accumulator (fused RMSNorm rstd), applied before bias
    RopeEpi rope;                   // optional fused RoPE of the leading output columns (EPI_NONE)
    int group_m;                    // tile rasterisation: ids walk down group_m tile rows, then to the next tile column
    int r_f32;                      // EPI_RESID: R is fp32 [M][ldr] (with OUT_F32)
    long lo_off;                    // bf16 output: also store lo = bf16(x - hi) at C + lo_off (padt_gemm_bf16_ex)
    x16_t* C2; long ldc2;          // fp32 output: optional bf16 mirror of C (fp32 residual stream + the next GEMM's A operand, padt_gemm_resid32)
    unsigned long long* prof;       // optional {first block start, last block end} in 100 MHz wall-clock ticks (padt_gemm_profile)
    const float* cs;                // optional per-output-column scale of the accumulator (fp8 weights: dequantisation scale of weight row n)
};

__device__ __attribute__((aligned(16))) unsigned int g_zero_page256[64];

namespace {
constexpr int TN = 256, TK = 64;                    // tile width and K-tile depth; the height is 64 * MF rows (MF = 2..4)
constexpr int HALF_BYTES = 128 * TK * 2;            // 16 KiB: 128 rows x 128 B
constexpr int OUT_PITCH = 512 + 32;                 // epilogue staging of a bf16 output tile: 256 columns + 32 B (ds_write_b64 of 16 rows x 32 B: 2-way conflicts)
constexpr int LDS_BYTES = 256 * OUT_PITCH;          // >= 8 * HALF_BYTES: 2 K-tiles x {A0, A1, B0, B1} in the main loop

PADT_DEV char* slot(char* smem, int parity, int is_b, int h) { return smem + ((parity * 4) + is_b * 2 + h) * HALF_BYTES; }

// Per-lane byte offsets (relative to the tile's first row at k = 0) of the two 1-KiB DMA pieces this wave issues for
// half-tile (is_b, h): piece c = 2*wave + i covers local rows 8c..8c+7; LDS slot (lr, lane & 7) holds source chunk
// (lane & 7) ^ (lr & 7).  Computed once per block; a K-step only moves the (wave-uniform) base pointer by 128 bytes.
template <int MF, int ES>
PADT_DEV unsigned piece_offset(int is_b, int h, int i, int wave, int lane, int row0, int nrows, long ld) {
    const int c = wave * 2 + i;
    const int lr = c * 8 + (lane >> 3);
    const int j = (lane & 7) ^ (lr & 7);
    // A half-tile h = rows [h*16*MF, (h+1)*16*MF) of each wave-row group's 32*MF rows; with MF < 4 the upper slots of the
    // 128-row LDS half are unused and their pieces re-read a valid row (keeps every wave at 2 pieces per stage, so the
    // counted vmcnt stays uniform)
    const int lra = lr < 32 * MF ? lr : 32 * MF - 1;
    int g = is_b ? (lr >> 5) * 64 + h * 32 + (lr & 31) : (lra / (16 * MF)) * (32 * MF) + h * (16 * MF) + lra % (16 * MF);
    g = (row0 + g < nrows) ? g : nrows - 1 - row0;                // clamp to the last valid row (results are not stored)
    return (unsigned)((long)g * ld * ES + j * 16);               // ES = bytes per element (2 bf16, 1 fp8); chunk j = 16 bytes of the row's K-tile
}

PADT_DEV void dma2(const char* base, unsigned off0, unsigned off1, char* dst, int wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off0),
                                     (__attribute__((address_space(3))) void*)(dst + (wave * 2) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off1),
                                     (__attribute__((address_space(3))) void*)(dst + (wave * 2 + 1) * 1024), 16, 0, 0);
}

PADT_DEV x16x8 rd(const char* half, int lr, int j) { return ld_frag(half + lr * 128 + ((j ^ (lr & 7)) << 4)); }

// fp8 (OCP e4m3) MFMA 16x16x128: a lane's operand is 32 consecutive K bytes = two 16-byte LDS chunks; both scale exponents 0 select the
// unscaled v_mfma_f32_16x16x128_f8f6f4 (checked: tools/ubench/f8probe.hip).  2048 FLOP per cycle per SIMD: twice the bf16 16x16x32 rate.
typedef __attribute__((ext_vector_type(8))) int i32x8;
PADT_DEV f32x4 mfma_f8(x16x8 a_lo, x16x8 a_hi, x16x8 b_lo, x16x8 b_hi, f32x4 c) {
    const u32x4 al = __builtin_bit_cast(u32x4, a_lo), ah = __builtin_bit_cast(u32x4, a_hi);
    const u32x4 bl = __builtin_bit_cast(u32x4, b_lo), bh = __builtin_bit_cast(u32x4, b_hi);
    const i32x8 a = {(int)al[0], (int)al[1], (int)al[2], (int)al[3], (int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]};
    const i32x8 b = {(int)bl[0], (int)bl[1], (int)bl[2], (int)bl[3], (int)bh[0], (int)bh[1], (int)bh[2], (int)bh[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}

PADT_DEV void unpack4b(u32x2 v, float* f) { unpack4x(v, f); }
// x * sigmoid(x) with the exact expf and an IEEE division: the split-output SwiGLU of precision="reference" (what padt_swiglu_split computes)
PADT_DEV float silu_exact(float x) { return x / (1.0f + expf(-x)); }
// (hi, lo) pair of four fp32 values at cp / cp + lo_off: hi = X(x), lo = X(x - hi)
PADT_DEV void store_split4(x16_t* cp, long lo_off, const float* o) {
    const u32x2 hi = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
    float hv[4];
    unpack4x(hi, hv);
    *reinterpret_cast<u32x2*>(cp) = hi;
    *reinterpret_cast<u32x2*>(cp + lo_off) = u32x2{pack2x(o[0] - hv[0], o[1] - hv[1]), pack2x(o[2] - hv[2], o[3] - hv[3])};
}
}  // namespace

// MF = 16-row MFMA blocks per wave per m-half: tile height 64*MF (256, 192 or 128 rows) x 256 columns.  The shorter tiles
// exist for wave quantisation: 4616 prompt rows x 2048 columns are 152 tiles of 256^2 on 256 CUs but 200 of 192x256.
// FP8: A and W are OCP e4m3 bytes (K-tile = 128 elements = the same 128 bytes per row: staging, LDS image and swizzle are unchanged); a
// lane's MFMA operand is the 32 consecutive K bytes [32 fq, 32 fq + 32) = LDS chunks 2 fq and 2 fq + 1, one 16x16x128 MFMA per fragment
// pair and K-tile instead of two 16x16x32; the accumulator is scaled by rs[m] * cs[n] (activation row scale x weight row scale).
// EPI_SWIGLU with OUT_F32 set (round 6) = the SPLIT SwiGLU of precision="reference": silu(gate) * up evaluated with the exact expf / division and
// stored as a (hi, lo) 16-bit pair, lo at p.lo_off — the [hi | lo] A operand of the down projection, without the fp32 gate / up rows in between.
template <int EPI, bool OUT_F32, int MF, bool FP8 = false>
__global__ __launch_bounds__(512) void gemm_tile256_kernel(Gemm256Args p) {
    constexpr int TMV = 64 * MF;
    constexpr bool SPLIT_GLU = (EPI == EPI_SWIGLU) && OUT_F32;
    constexpr int ES = FP8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    ProfScope prof_scope(p.prof, tid);
    const int ntn = (p.N + TN - 1) / TN, ntm = (p.M + TMV - 1) / TMV;
    // Each XCD owns a contiguous run of tile ids (xcd_remap) and its 32 CUs work on ~32 consecutive ids at a time; walking
    // the ids in group_m x (32 / group_m) patches makes those tiles share group_m A panels and 32/group_m W panels in the
    // XCD's 4 MiB L2 instead of 1 + 32 (measured FETCH_SIZE of the gate/up GEMM: 7x its algorithmic bytes when row-major).
    const int id = xcd_remap(blockIdx.x, ntm * ntn);
    const int gm = p.group_m;
    const int per_group = gm * ntn;
    const int first_m = (id / per_group) * gm;
    const int gsz = (ntm - first_m) < gm ? (ntm - first_m) : gm;
    const int tm = first_m + (id % per_group) % gsz, tn = (id % per_group) / gsz;
    const int m0 = tm * TMV, n0 = tn * TN;
    const int wr = wave >> 2, wc = wave & 3;
    const int nk = (p.K * ES + 2 * TK - 1) / (2 * TK);            // K-tiles of 128 bytes per row

    f32x4 acc[2 * MF][4];
#pragma unroll
    for (int i = 0; i < 2 * MF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    x16x8 af[MF][2], b0[2][2], b1[2][2];                          // A sub-tile, B n-half 0 (kept all tile), B n-half 1

    unsigned offA[2][2], offB[2][2];                              // [half][piece] per-lane byte offsets, see piece_offset
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            offA[h][i] = piece_offset<MF, ES>(0, h, i, wave, lane, m0, p.M, p.lda);
            offB[h][i] = piece_offset<MF, ES>(1, h, i, wave, lane, n0, p.N, p.ldw);
        }
    const char* tileA = reinterpret_cast<const char*>(p.A) + (long)m0 * p.lda * ES;     // wave-uniform bases
    const char* tileW = reinterpret_cast<const char*>(p.W) + (long)n0 * p.ldw * ES;
    auto stage_half = [&](int t, int is_b, int h, char* dst) {    // K % 64 == 0 (dispatcher): no tail, no zero page
        if (is_b) dma2(tileW + (long)t * (TK * 2), offB[h][0], offB[h][1], dst, wave);
        else dma2(tileA + (long)t * (TK * 2), offA[h][0], offA[h][1], dst, wave);
    };

    // ---- TWO phases of 32 MFMAs per K-tile instead of four of 16: a 16-MFMA segment is ≈270 cycles and every interval between two
    // barriers carries ≈150 cycles of synchronisation cost, so halving the barrier count is worth more than the finer DMA / ds_read
    // interleave (same-call A/B, profiles/r02_gemm256_experiments.md: 8192^3 1342 → 1437 TFLOP/s, prefill down +13 %, gate/up +5 %;
    // a variant with 4 + 4 DMA pieces per phase and the counted wait inside the load segment measured 3 % below this 2 + 6 split):
    //   phase A(t): reads B n-half 0, B n-half 1, A m-half 0 (16 ds_read_b128) | stages A1(t+1)            | MFMA (m0,n0) (m0,n1)
    //   phase B(t): reads A m-half 1 (8)                                     | stages A0, B0, B1 of t + 2 | MFMA (m1,n1) (m1,n0)
    // Hazards (g0 = leading group, g1 one barrier behind; interval numbering per K-tile t: g0 loads A(t) in I(4t), multiplies in
    // I(4t+1) while g1 loads A(t); g0 loads B(t) in I(4t+2), ...):
    //   WAR  A1(t-1) is last read by g1 in I(4t-1), re-staged from I(4t) on; A0/B0/B1(t) last read in I(4t+1), re-staged from I(4t+2);
    //   RAW  A0/B0/B1(t+1) are issued in I(4t-2)/I(4t-1), retired by the wait at the end of MFMA A(t) (g0 I(4t+1), g1 I(4t+2)) with
    //        only phase A(t)'s 2 pieces younger → vmcnt(2), first read in I(4t+4); A1(t+1) is issued in I(4t)/I(4t+1), retired at
    //        the end of MFMA B(t) (I(4t+3)/I(4t+4)) with phase B(t)'s 6 pieces younger → vmcnt(6), first read in I(4t+6).
    stage_half(0, 0, 0, slot(smem, 0, 0, 0));
    stage_half(0, 1, 0, slot(smem, 0, 1, 0));
    stage_half(0, 1, 1, slot(smem, 0, 1, 1));
    stage_half(0, 0, 1, slot(smem, 0, 0, 1));
    if (nk > 1) {
        stage_half(1, 0, 0, slot(smem, 1, 0, 0));
        stage_half(1, 1, 0, slot(smem, 1, 1, 0));
        stage_half(1, 1, 1, slot(smem, 1, 1, 1));
    }
    const int wr_u = __builtin_amdgcn_readfirstlane(wr);          // provably wave-uniform → scalar branch around s_barrier
    if (wr_u == 1) __builtin_amdgcn_s_barrier();
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // tile 0 complete (A0, B0, B1 of tile 1 may still fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // K == 64: only tile 0's eight pieces were issued
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();                                 // both groups have waited before anyone reads tile 0
    auto phase2 = [&](int t, auto ph_tag, auto steady_tag) {
        constexpr int PH = decltype(ph_tag)::value;               // 0 = A, 1 = B
        constexpr bool STEADY = decltype(steady_tag)::value;
        const int par = t & 1;
        if (PH == 0) {
            const char* bh0 = slot(smem, par, 1, 0);
            const char* bh1 = slot(smem, par, 1, 1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    b0[i][kk] = rd(bh0, wc * 32 + i * 16 + frow, FP8 ? 2 * fq + kk : kk * 4 + fq);
                    b1[i][kk] = rd(bh1, wc * 32 + i * 16 + frow, FP8 ? 2 * fq + kk : kk * 4 + fq);
                }
        }
        {
            const char* ah = slot(smem, par, 0, PH);
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) af[i][kk] = rd(ah, wr * (16 * MF) + i * 16 + frow, FP8 ? 2 * fq + kk : kk * 4 + fq);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PH == 0) { if (STEADY || t + 1 < nk) stage_half(t + 1, 0, 1, slot(smem, par ^ 1, 0, 1)); }
        else if (STEADY || t + 2 < nk) {
            stage_half(t + 2, 0, 0, slot(smem, par, 0, 0));
            stage_half(t + 2, 1, 0, slot(smem, par, 1, 0));
            stage_half(t + 2, 1, 1, slot(smem, par, 1, 1));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {                             // quadrant order (m0,n0) (m0,n1) | (m1,n1) (m1,n0)
            const int NHq = (PH == 0) ? q : 1 - q;
            if (FP8) {
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[PH * MF + i][NHq * 2 + j] = mfma_f8(NHq ? b1[j][0] : b0[j][0], NHq ? b1[j][1] : b0[j][1], af[i][0], af[i][1],
                                                                acc[PH * MF + i][NHq * 2 + j]);
            } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[PH * MF + i][NHq * 2 + j] = mfma16(NHq ? b1[j][kk] : b0[j][kk], af[i][kk], acc[PH * MF + i][NHq * 2 + j]);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (STEADY) { if (PH == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    {
        using Q0 = std::integral_constant<int, 0>;
        using Q1 = std::integral_constant<int, 1>;
        int t = 0;
        for (; t + 2 < nk; ++t) {
            phase2(t, Q0{}, std::true_type{});
            phase2(t, Q1{}, std::true_type{});
        }
        for (; t < nk; ++t) {
            phase2(t, Q0{}, std::false_type{});
            phase2(t, Q1{}, std::false_type{});
        }
    }
    if (wr_u == 0) __builtin_amdgcn_s_barrier();                  // matches the extra barrier of the lagging group

    // ---- epilogue (swapped MFMA: lane holds row m, 4 consecutive columns)
    const x16_t* zpage = reinterpret_cast<const x16_t*>(g_zero_page256);
    // FAST PATH (full column tile): straight-line code.  gfx9 counts loads and stores in ONE
    // in-order vmcnt, so an epilogue that loads (bias / residual / RoPE angles) right before every store waits for the previous
    // store's round trip 32 times per wave — measured 12-14 us per tile whatever the number of busy CUs, a third of a K = 1280 GEMM
    // (profiles/r02_gemm256_experiments.md).  Here bias and row scales are loaded once, the per-row-block operands (residual quads,
    // cos / sin pairs) are fetched one 16-row block AHEAD of the stores, nothing branches per fragment, and rows past M are masked
    // at the store only (their loads are clamped to row M - 1): the stores are fire-and-forget and drain while the CU already runs
    // its next block.
    if (n0 + TN <= p.N) {
        // bf16 tiles leave through LDS (free after the main loop) so that the global stores are whole 512-byte rows
        const bool wide = !OUT_F32 && EPI != EPI_SWIGLU && p.lo_off == 0 && (reinterpret_cast<unsigned long>(p.C) & 15) == 0 && (p.ldc & 7) == 0;
        const bool mirror = OUT_F32 && p.C2 != nullptr;           // fp32 tile to C, its bf16 image through LDS to C2 (16-byte aligned, ldc2 % 8: checked by the host)
        const int nb = n0 + wc * 64 + fq * 4;                     // this lane's first column of fragment column ni: nb + 16 * ni
        const int mb = m0 + wr * (32 * MF) + frow;                // this lane's row of fragment row mi: mb + 16 * mi
        float bv[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) unpack4b(*reinterpret_cast<const u32x2*>(p.bias ? p.bias + nb + ni * 16 : zpage), bv[ni]);
        f32x4 csc[4];                                             // per-column scales (fp8 weights), folded into the accumulator with the row scale
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) csc[ni] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (FP8 && p.cs) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) csc[ni] = *reinterpret_cast<const f32x4*>(p.cs + nb + ni * 16);
        }
        float rsc[2 * MF];
#pragma unroll
        for (int mi = 0; mi < 2 * MF; ++mi) rsc[mi] = 1.0f;
        if (p.rs) {
#pragma unroll
            for (int mi = 0; mi < 2 * MF; ++mi) rsc[mi] = p.rs[min(mb + mi * 16, p.M - 1)];
        }
        auto run = [&](auto rope_tag, auto rf32_tag) {
            constexpr bool ROPE = decltype(rope_tag)::value;
            constexpr bool RF32 = decltype(rf32_tag)::value;      // fp32 residual stream (split-precision decoder)
            u32x2 rr[2][4];                                       // residual quads of one row block, double-buffered
            f32x4 rr32[RF32 ? 2 : 1][4];
            float2 cc[2][4], ss[2][4];                            // RoPE angle pairs
            int pi[4];
            float keep[4];                                        // 0 → column is not rotated (v part): cos 1, sin 0
            if (ROPE) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    pi[ni] = ((nb + ni * 16) % p.rope.D) >> 1;
                    keep[ni] = (nb + ni * 16 < p.rope.cols) ? 1.0f : 0.0f;
                }
            }
            auto prefetch = [&](int mi, int b) {
                const long mc = min(mb + mi * 16, p.M - 1);
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    if (EPI == EPI_RESID && !RF32) rr[b][ni] = *reinterpret_cast<const u32x2*>(p.R + mc * p.ldr + nb + ni * 16);
                    if (EPI == EPI_RESID && RF32) rr32[RF32 ? b : 0][ni] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + mc * p.ldr + nb + ni * 16);
                    if (ROPE) {
                        cc[b][ni] = *reinterpret_cast<const float2*>(p.rope.cos + mc * p.rope.ld + pi[ni]);
                        ss[b][ni] = *reinterpret_cast<const float2*>(p.rope.sin + mc * p.rope.ld + pi[ni]);
                    }
                }
            };
            if (EPI == EPI_RESID || ROPE) prefetch(0, 0);
#pragma unroll
            for (int mi = 0; mi < 2 * MF; ++mi) {
                const int b = mi & 1;
                if ((EPI == EPI_RESID || ROPE) && mi + 1 < 2 * MF) prefetch(mi + 1, b ^ 1);
                const int m = mb + mi * 16;
                const bool live = m < p.M;
                if (EPI == EPI_SWIGLU) {
#pragma unroll
                    for (int ni = 0; ni < 4; ni += 2) {
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            o[r] = FP8 ? silu(acc[mi][ni][r] * (rsc[mi] * csc[ni][r]) + bv[ni][r]) * (acc[mi][ni + 1][r] * (rsc[mi] * csc[ni + 1][r]) + bv[ni + 1][r])
                                 : SPLIT_GLU ? silu_exact(acc[mi][ni][r] * rsc[mi] + bv[ni][r]) * (acc[mi][ni + 1][r] * rsc[mi] + bv[ni + 1][r])
                                             : silu(acc[mi][ni][r] * rsc[mi] + bv[ni][r]) * (acc[mi][ni + 1][r] * rsc[mi] + bv[ni + 1][r]);
                        const int no = (n0 >> 1) + wc * 32 + (ni >> 1) * 16 + fq * 4;
                        if (SPLIT_GLU) {
                            if (live) store_split4(reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + no, p.lo_off, o);
                        } else if (live) *reinterpret_cast<u32x2*>(reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + no) = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        float o[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = FP8 ? acc[mi][ni][r] * (rsc[mi] * csc[ni][r]) + bv[ni][r] : acc[mi][ni][r] * rsc[mi] + bv[ni][r];
                        if (ROPE) {
                            const float c0 = keep[ni] != 0.f ? cc[b][ni].x : 1.0f, c1 = keep[ni] != 0.f ? cc[b][ni].y : 1.0f;
                            const float s0 = keep[ni] != 0.f ? ss[b][ni].x : 0.0f, s1 = keep[ni] != 0.f ? ss[b][ni].y : 0.0f;
                            const float a0 = o[0], b0r = o[1], a1 = o[2], b1r = o[3];
                            o[0] = rope_lo(a0, b0r, c0, s0);
                            o[1] = rope_hi(a0, b0r, c0, s0);
                            o[2] = rope_lo(a1, b1r, c1, s1);
                            o[3] = rope_hi(a1, b1r, c1, s1);
                        }
                        if (EPI == EPI_GELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
                        }
                        if (EPI == EPI_RESID && !RF32) {
                            float rv[4];
                            unpack4b(rr[b][ni], rv);
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] += rv[r];
                        }
                        if (EPI == EPI_RESID && RF32) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] += rr32[RF32 ? b : 0][ni][r];
                        }
                        const long off = (long)m * p.ldc + nb + ni * 16;
                        if (OUT_F32) {
                            if (live) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + off) = f32x4{o[0], o[1], o[2], o[3]};
                            if (mirror) *reinterpret_cast<u32x2*>(smem + (wr * (32 * MF) + mi * 16 + frow) * OUT_PITCH + (wc * 64 + ni * 16 + fq * 4) * 2) =
                                u32x2{pack2x(o[0] * PADT_STREAM_SCALE, o[1] * PADT_STREAM_SCALE), pack2x(o[2] * PADT_STREAM_SCALE, o[3] * PADT_STREAM_SCALE)};
                        } else if (wide) {                        // bf16 tile → LDS in row-major order, written out as full rows below
                            *reinterpret_cast<u32x2*>(smem + (wr * (32 * MF) + mi * 16 + frow) * OUT_PITCH + (wc * 64 + ni * 16 + fq * 4) * 2) =
                                u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
                        } else if (live) {
                            const u32x2 hi = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
                            x16_t* cp = reinterpret_cast<x16_t*>(p.C) + off;
                            *reinterpret_cast<u32x2*>(cp) = hi;
                            if (p.lo_off) {                       // split-precision pair: lo = bf16(x - hi)
                                float hv[4];
                                unpack4b(hi, hv);
                                *reinterpret_cast<u32x2*>(cp + p.lo_off) = u32x2{pack2x(o[0] - hv[0], o[1] - hv[1]), pack2x(o[2] - hv[2], o[3] - hv[3])};
                            }
                        }
                    }
                }
            }
        };
        if (EPI == EPI_NONE && p.rope.cos != nullptr) run(std::true_type{}, std::false_type{});
        else if (EPI == EPI_RESID && OUT_F32 && p.r_f32) run(std::false_type{}, std::true_type{});
        else run(std::false_type{}, std::false_type{});
        if (wide || mirror) {
            x16_t* cw = mirror ? p.C2 : reinterpret_cast<x16_t*>(p.C);
            const long ldw_out = mirror ? p.ldc2 : p.ldc;
            // 8-byte fragment stores put 16 x 32-byte pieces on the wire per instruction and cost 4-8 us per tile in the memory system
            // (same instruction count into one 512-byte region: 1.5 us); full 512-byte rows, 16 bytes per lane, do not.
            __syncthreads();
            const int c16 = lane & 31;
#pragma unroll
            for (int it = 0; it < 4 * MF; ++it) {
                const int row = wave * (8 * MF) + it * 2 + (lane >> 5);
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * OUT_PITCH + c16 * 16);
                if (m0 + row < p.M) *reinterpret_cast<u32x4*>(cw + (long)(m0 + row) * ldw_out + n0 + c16 * 8) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2 * MF; ++mi) {
        const int m = m0 + wr * (32 * MF) + mi * 16 + frow;
        if (m >= p.M) continue;
        const float rsc = p.rs ? p.rs[m] : 1.0f;
        if (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int n = n0 + wc * 64 + ni * 16 + fq * 4;    // interleaved row index of the gate quad
                if (n >= p.N) continue;
                const x16_t* bp = p.bias ? p.bias + n : zpage;
                float gb[4], ub[4], o[4];
                unpack4b(*reinterpret_cast<const u32x2*>(bp), gb);
                unpack4b(*reinterpret_cast<const u32x2*>(bp + (p.bias ? 16 : 0)), ub);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[r] = SPLIT_GLU ? silu_exact(acc[mi][ni][r] * rsc + gb[r]) * (acc[mi][ni + 1][r] * rsc + ub[r])
                                     : silu(acc[mi][ni][r] * rsc + gb[r]) * (acc[mi][ni + 1][r] * rsc + ub[r]);
                const int no = (n >> 5) * 16 + (n & 15);
                if (SPLIT_GLU) store_split4(reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + no, p.lo_off, o);
                else *reinterpret_cast<u32x2*>(reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + no) =
                    u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wc * 64 + ni * 16 + fq * 4;
                if (n + 3 >= p.N) {                               // ragged N tail: scalar
                    for (int r = 0; r < 4 && n + r < p.N; ++r) {
                        float x = acc[mi][ni][r] * rsc;
                        if (p.bias) x += x2f(p.bias[n + r]);
                        if (EPI == EPI_GELU) x = gelu_erf(x);
                        if (EPI == EPI_RESID) x += p.r_f32 ? reinterpret_cast<const float*>(p.R)[(long)m * p.ldr + n + r] : x2f(p.R[(long)m * p.ldr + n + r]);
                        if (OUT_F32) {
                            reinterpret_cast<float*>(p.C)[(long)m * p.ldc + n + r] = x;
                            if (p.C2) p.C2[(long)m * p.ldc2 + n + r] = f2x(x * PADT_STREAM_SCALE);
                        } else {
                            const x16_t hb = f2x(x);
                            reinterpret_cast<x16_t*>(p.C)[(long)m * p.ldc + n + r] = hb;
                            if (p.lo_off) reinterpret_cast<x16_t*>(p.C)[(long)m * p.ldc + n + r + p.lo_off] = f2x(x - x2f(hb));
                        }
                    }
                    continue;
                }
                const x16_t* bp = p.bias ? p.bias + n : zpage;
                float bv[4], o[4];
                unpack4b(*reinterpret_cast<const u32x2*>(bp), bv);
                u32x2 rraw = u32x2{0u, 0u};
                f32x4 rf = f32x4{0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_RESID) {
                    if (p.r_f32) rf = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (long)m * p.ldr + n);
                    else rraw = *reinterpret_cast<const u32x2*>(p.R + (long)m * p.ldr + n);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[mi][ni][r] * rsc + bv[r];
                if (EPI == EPI_NONE) rope_pairs(o, m, n, p.rope);
                if (EPI == EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = gelu_erf(o[r]);
                }
                if (EPI == EPI_RESID) {
                    float rv[4];
                    unpack4b(rraw, rv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += p.r_f32 ? rf[r] : rv[r];
                }
                if (OUT_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
                    if (p.C2) *reinterpret_cast<u32x2*>(p.C2 + (long)m * p.ldc2 + n) =
                        u32x2{pack2x(o[0] * PADT_STREAM_SCALE, o[1] * PADT_STREAM_SCALE), pack2x(o[2] * PADT_STREAM_SCALE, o[3] * PADT_STREAM_SCALE)};
                } else {
                    const u32x2 hi = u32x2{pack2x(o[0], o[1]), pack2x(o[2], o[3])};
                    x16_t* cp = reinterpret_cast<x16_t*>(p.C) + (long)m * p.ldc + n;
                    *reinterpret_cast<u32x2*>(cp) = hi;
                    if (p.lo_off) {                               // split-precision pair: lo = bf16(x - hi)
                        float hv[4];
                        unpack4b(hi, hv);
                        *reinterpret_cast<u32x2*>(cp + p.lo_off) = u32x2{pack2x(o[0] - hv[0], o[1] - hv[1]), pack2x(o[2] - hv[2], o[3] - hv[3])};
                    }
                }
            }
        }
    }
}

template <int EPI, bool F32, int MF, bool FP8 = false>
static void launch256_mf(const Gemm256Args& a, hipStream_t s) {
    static PerDeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile256_kernel<EPI, F32, MF, FP8>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); });
    const int ntm = (a.M + 64 * MF - 1) / (64 * MF), ntn = (a.N + TN - 1) / TN;
    hipLaunchKernelGGL((gemm_tile256_kernel<EPI, F32, MF, FP8>), dim3(ntm * ntn), dim3(512), LDS_BYTES, s, a);
}

// Tile height by a wave-quantisation cost model: rounds of 256 co-resident tiles x tile height, with a penalty for the
// shorter tiles (fewer MFMAs per staged B half-tile).  A ragged last tile row of <= 64 rows can be PEELED off (returned to
// the caller, who streams it through the skinny kernel) when that saves a whole round: the ViT gate/up GEMM is
// 16928 = 66 x 256 + 32 rows x 27 tile columns → 1809 tiles = 8 rounds, 1782 tiles = 7 rounds without the 32-row tail.
struct Plan256 { double cost; int mf; long rows; };

static Plan256 plan256(long M, long N, long K, int force_mf, bool allow_peel, bool force_peel) {
    const long ntn = (N + TN - 1) / TN;
    Plan256 best{1e30, 4, M};
    const double penalty[5] = {0, 0, 1.35, 1.12, 1.0};           // measured at 8192^3: 980 / 1163 / 1349 TFLOP/s
    for (int mf = 4; mf >= 2; --mf) {
        if (force_mf >= 2 && force_mf <= 4 && mf != force_mf) continue;
        const long th = 64 * mf;
        const long tiles = ((M + th - 1) / th) * ntn;
        double cost = (double)((tiles + 255) / 256) * mf * penalty[mf];
        long rows = M;
        const long tail = M % th;
        if (allow_peel && tail > 0 && tail <= 64 && M > th) {
            // skinny pass over `tail` rows ≈ 6 us + N*K*2 B at 4 TB/s; one cost unit = a 64-row tile slab ≈ K * 0.0082 us
            const double skinny_units = (6.0 + (double)N * K * 2.0 / 4.0e6) / (K * 0.0082);
            const double c2 = (double)(((M / th) * ntn + 255) / 256) * mf * penalty[mf] + skinny_units;
            if (c2 < cost || force_peel) { cost = c2; rows = M - tail; }
        }
        if (cost < best.cost - 1e-9) best = Plan256{cost, mf, rows};
    }
    return best;
}

template <int EPI, bool F32>
static void run256(Gemm256Args a, int mf, hipStream_t s) {
    if (mf == 4) launch256_mf<EPI, F32, 4>(a, s);
    else if (mf == 3) launch256_mf<EPI, F32, 3>(a, s);
    else launch256_mf<EPI, F32, 2>(a, s);
}

// A second way out of a mostly empty last round: split the COLUMNS.  The prefill gate/up GEMM is 19 x 86 tiles = 6.4 rounds
// of 256-row tiles; its first 80 tile columns are 5.94 rounds and the remaining 6 columns run as one round of 128-row
// tiles (222 of them) — two launches, 6 + 0.68 round-equivalents instead of 7.
template <int EPI, bool F32>
static long launch256(Gemm256Args a, hipStream_t s) {
    const int force = g_knobs.mf;
    const bool allow_peel = g_knobs.peel != 0;                    // 0 never, 1 cost model (default), 2 always when a tail exists (tests)
    const bool force_peel = g_knobs.peel == 2;
    const int colsplit = g_knobs.colsplit;                        // 0 never, 1 cost model (default), >= 2: always peel that many tile columns (tests)
    const long ntn = (a.N + TN - 1) / TN;
    const Plan256 whole = plan256(a.M, a.N, a.K, force, allow_peel, force_peel);
    long split_cols = 0;
    Plan256 pm = whole, pr = whole;
    if (colsplit != 0 && a.rope.cos == nullptr && ntn >= 4 && a.N % TN == 0 && whole.rows == a.M) {
        double best_total = whole.cost;
        for (long c = 1; c <= 16 && c < ntn - 1; ++c) {
            if (colsplit >= 2 && c != colsplit) continue;
            const Plan256 m1 = plan256(a.M, (ntn - c) * TN, a.K, force, false, false);
            const Plan256 r1 = plan256(a.M, c * TN, a.K, force, false, false);
            const double total = m1.cost + r1.cost + 0.15;        // + a launch
            if (total < best_total - 1e-9 || colsplit >= 2) { best_total = total; split_cols = c; pm = m1; pr = r1; }
        }
    }
    if (split_cols == 0) {
        a.M = (int)whole.rows;
        run256<EPI, F32>(a, whole.mf, s);
        return whole.rows;
    }
    const long n1 = (ntn - split_cols) * TN;                      // weight rows (= pre-epilogue columns) of the first launch
    const long c1 = (EPI == EPI_SWIGLU) ? n1 / 2 : n1;            // output columns it writes
    Gemm256Args a1 = a, a2 = a;
    a1.N = (int)n1;
    a2.N = a.N - (int)n1;
    a2.W = a.W + n1 * a.ldw;
    if (a.bias) a2.bias = a.bias + n1;
    a2.C = (F32 && EPI != EPI_SWIGLU) ? (void*)((float*)a.C + c1) : (void*)((x16_t*)a.C + c1);    // the split SwiGLU (F32 tag) writes 16-bit pairs
    if (a.R) a2.R = a.r_f32 ? (const x16_t*)((const float*)a.R + c1) : a.R + c1;
    if (a.C2) a2.C2 = a.C2 + c1;
    run256<EPI, F32>(a1, pm.mf, s);
    run256<EPI, F32>(a2, pr.mf, s);
    return a.M;
}

// Called by padt_gemm_bf16's dispatcher (gemm.hip) for shapes where the 256^2 tiling pays; arguments already validated.
// Returns 0 when it took the launch, 1 when the shape should stay on the 128^2 kernel.
// *rows_done = leading rows it computed (< M when a short ragged tail is left to the caller's skinny kernel).
extern "C" int PADT_TWIN(padt_gemm256_try)(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C,
                                long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32,
                                const float* row_scale, const RopeEpi* rope, long* rows_done, int resid_f32, long lo_off,
                                void* C2, long ldc2, unsigned long long* prof) {
    const int mode = g_knobs.mode;                                // 0 off, 1 auto, 2 force
    if (mode == 0) return 1;
    if (mode == 1) {
        // auto: measured on MI355X (profiles/r01_gemm_tile_experiments.md) the phase-pipelined kernel wins or ties on every
        // ViT / prefill shape of the model once all three dimensions are a few tiles deep; tiny problems keep the 128^2
        // kernel (2 blocks/CU hide their prologue / epilogue better)
        if (M < 512 || N < 512 || K < 512) return 1;
    }
    if (K % TK) return 1;                                         // no K-tail path in this kernel
    const int group_m = g_knobs.group_m;
    Gemm256Args a{(const x16_t*)A, lda, (const x16_t*)W, ldw, (const x16_t*)bias, C, ldc, (const x16_t*)R, ldr,
                  (int)M, (int)N, (int)K, row_scale, *rope, group_m < 1 ? 1 : group_m, resid_f32, lo_off, (x16_t*)C2, ldc2, prof, nullptr};
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue * 2 + (out_f32 ? 1 : 0)) {
        case 0: *rows_done = launch256<EPI_NONE, false>(a, s); break;
        case 1: *rows_done = launch256<EPI_NONE, true>(a, s); break;
        case 2: *rows_done = launch256<EPI_GELU, false>(a, s); break;
        case 3: *rows_done = launch256<EPI_GELU, true>(a, s); break;
        case 4: *rows_done = launch256<EPI_RESID, false>(a, s); break;
        case 5: *rows_done = launch256<EPI_RESID, true>(a, s); break;
        case 6: *rows_done = launch256<EPI_SWIGLU, false>(a, s); break;
        case 7: *rows_done = launch256<EPI_SWIGLU, true>(a, s); break;      // split SwiGLU: (hi, lo) 16-bit pairs, lo at lo_off (not an fp32 tile)
        default: return 1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp8 x fp8 MFMA GEMM (BASELINE configs[4], the 7B "fp8 MFMA weight path" at prompt length): A8 = e4m3 activations [M][K] with one fp32 scale
// per row (padt_quant_rows_fp8: amax scaling, optionally x the RMSNorm rstd of the row), W8 = e4m3 weights [N][K] (nn.Linear layout) with one
// fp32 scale per weight row.  C = epi(rs[m] * cs[n] * (A8 · W8^T) + bias): epilogue 0 → bf16 C; 3 → SwiGLU (W rows interleaved gate16 | up16),
// bf16 C with N / 2 columns; 2 → the fp32 residual stream: X32 += ..., Xb = bf16(X32) (C unused).  The fp8 products are exact in fp32;
// accumulation is fp32.  K % 128 == 0, N % 256 == 0, lda / ldw % 16 == 0, 16-byte aligned operands.
template <int EPI, bool F32>
static void run256_fp8(const Gemm256Args& a, int mf, hipStream_t s) {
    if (mf == 4) launch256_mf<EPI, F32, 4, true>(a, s);
    else if (mf == 3) launch256_mf<EPI, F32, 3, true>(a, s);
    else launch256_mf<EPI, F32, 2, true>(a, s);
}

extern "C" int PADT_TWIN(padt_gemm_fp8_impl)(void* stream, const void* A8, long lda, const void* W8, long ldw, const void* row_scale, const void* col_scale,
                             const void* bias, void* C, long ldc, void* X32, long ldx, void* Xb, long ldxb, long M, long N, long K, int epilogue,
                             unsigned long long* prof) {
    if (M <= 0 || N <= 0) return 0;
    if (K <= 0 || (K & 127) || (N & 255) || (lda & 15) || (ldw & 15) || ((uintptr_t)A8 & 15) || ((uintptr_t)W8 & 15) || row_scale == nullptr ||
        col_scale == nullptr || ((uintptr_t)col_scale & 15) || ((uintptr_t)bias & 7)) {
        padt_set_error("padt_gemm_fp8: K % 128 == 0, N % 256 == 0, lda / ldw % 16 == 0, 16-byte aligned A8 / W8 / col_scale and both scale vectors required");
        return -1;
    }
    RopeEpi norope{nullptr, nullptr, 0, 0, 0};
    const Plan256 pl = plan256(M, N, K / 2, g_knobs.mf, false, false);      // cost model in K-tiles: a 128-element fp8 K-tile costs what 64 bf16 elements do
    Gemm256Args a{(const x16_t*)A8, lda, (const x16_t*)W8, ldw, (const x16_t*)bias, C, ldc, nullptr, 0, (int)M, (int)N, (int)K,
                  (const float*)row_scale, norope, g_knobs.group_m < 1 ? 1 : g_knobs.group_m, 0, 0, nullptr, 0, prof, (const float*)col_scale};
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_NONE) {
        if (C == nullptr || (ldc & 7) || ((uintptr_t)C & 15) || ldc < N) { padt_set_error("padt_gemm_fp8: bf16 C with ldc % 8 == 0, ldc >= N required"); return -1; }
        run256_fp8<EPI_NONE, false>(a, pl.mf, s);
    } else if (epilogue == EPI_SWIGLU) {
        if (C == nullptr || (ldc & 7) || ((uintptr_t)C & 15) || ldc < N / 2) { padt_set_error("padt_gemm_fp8: SwiGLU needs bf16 C with ldc >= N / 2"); return -1; }
        run256_fp8<EPI_SWIGLU, false>(a, pl.mf, s);
    } else if (epilogue == EPI_RESID) {
        if (X32 == nullptr || (ldx & 3) || ((uintptr_t)X32 & 15) || (Xb && ((ldxb & 7) || ((uintptr_t)Xb & 15)))) {
            padt_set_error("padt_gemm_fp8: epilogue 2 updates the fp32 stream X32 (ldx % 4, 16-byte aligned) and writes the optional bf16 mirror Xb (ldxb % 8)");
            return -1;
        }
        a.C = X32; a.ldc = ldx; a.R = (const x16_t*)X32; a.ldr = ldx; a.r_f32 = 1; a.C2 = (x16_t*)Xb; a.ldc2 = ldxb;
        run256_fp8<EPI_RESID, true>(a, pl.mf, s);
    } else { padt_set_error("padt_gemm_fp8: epilogue 0, 2 or 3"); return -1; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { padt_set_error(hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace PADT_NS
