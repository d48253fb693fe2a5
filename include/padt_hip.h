/* padt_hip.h — C ABI of libpadt_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for PaDT's
 * generate-with-Visual-Reference-Tokens hot path.
 *
 * The reference (Gorilla-Lab-SCUT/PaDT) has no native boundary: every kernel is reached through a third-party Python
 * call (torch ATen/BLAS, flash-attn).  Each entry below names the reference call site(s) it replaces (file:line into
 * the reference repo; "HF:" = transformers models/qwen2_5_vl/modeling_qwen2_5_vl.py as cited in SURVEY.md).
 *
 * Conventions
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it, allocates nothing and keeps no hidden
 *     state (work buffers are caller-provided; sizes from the *_workspace/_nblk query functions).  The only process-wide
 *     state lives behind two test / measurement surfaces that no product call path touches: padt_gemm_knobs (forced tile
 *     dispatch variants) and padt_gemm_profile (in-kernel launch timing).
 *   - All pointers are device pointers.  bf16 = raw 16-bit brain floats.  Strides (`ld*`) are in ELEMENTS.
 *   - OPERAND TYPES.  This header declares the bf16 instantiation of the library.  Every entry point below that reads or writes 16-bit
 *     floats exists a second time for IEEE fp16 operands (`v_mfma_f32_16x16x32_f16`: same rate, 3 more mantissa bits — the default
 *     ViT / LLM operand type since round 4): same arguments, name suffixed `_f16` (`padt_gemm_bf16` → `padt_gemm_f16`,
 *     `padt_row_rstd` → `padt_row_rstd_f16`), declared in the generated `include/padt_hip_f16.h`.  Entry points that only MOVE 16-bit
 *     words (padt_gather_rows, padt_pack_rows, padt_embed_tokens, padt_greedy_step's hidden-row stash) serve both types.  The one
 *     numerical difference: the fp16 instantiation writes the 16-bit mirror of an fp32 residual stream (padt_gemm_resid32,
 *     padt_gemm_packed_resid32, padt_gemm_fp8 epilogue 2) scaled by PADT_F16_STREAM_SCALE — see there.
 *   - Return 0 on success, -1 for rejected arguments, -2 for a HIP launch error; text via padt_last_error().
 *   - Row-major everywhere; weights in nn.Linear layout [out_features][in_features].
 */
#ifndef PADT_HIP_H
#define PADT_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- plumbing --------------------------------------------------------------------------------------------------- */
int         padt_abi_version(void);                                  /* 4: round 6 (collect summary, output scores, f32-MFMA attention, padt_gemm_split_rows, padt_argmax_rows_f32, split SwiGLU); 3: round 5; 2: round 4 (fp16 twins) */
/* The fp16 instantiation stores mirror = fp16(PADT_F16_STREAM_SCALE * X32): a residual stream is un-normalised (checkpoints carry "massive
 * activations" of 1e3-1e4 in a few channels), fp16 ends at 65504, and every consumer of a mirror is scale-invariant — padt_row_rstd_f16 and the
 * fused RMSNorm statistics of padt_gemm_packed_f16 / padt_quant_rows_fp8_f16 return rstd / scale when called with eps * scale^2, which the
 * projection's row scale then applies to (scale * x) · W.  Returns the bf16 instantiation's factor (1) for f16 = 0, 2^-4 for f16 = 1. */
float       padt_stream_scale(int f16);
const char* padt_last_error(void);
int         padt_device_info(int device, char* name, int name_len, int* n_cu, long* hbm_bytes);
int         padt_memset(void* stream, void* dst, int value, long bytes);
int         padt_event_create(void** ev);
int         padt_event_record(void* ev, void* stream);
int         padt_event_elapsed_ms(void* start, void* stop, float* ms);
int         padt_event_destroy(void* ev);

/* ---- GEMM: C[M,N] = epi(row_scale[m] * (A[M,K] · W[N,K]^T) + bias) ---------------------------------------------------
 * epilogue: 0 none, 1 exact-erf GELU, 2 += R (residual), 3 SwiGLU (W rows interleaved gate16|up16; C has N/2 columns).
 * out_f32: C is float instead of bf16.  M <= 64 takes the weight-streaming (HBM-bound) kernel, else the phase-pipelined
 * 256/192/128x256x64 (or the 128x128x64) LDS-DMA MFMA tile kernel.  row_scale (fp32 [M], may be null) scales the
 * accumulator before the bias: with padt_row_rstd and the norm weight folded into W this is RMSNorm → Linear
 * (HF:74-79 + the projections of HF:219-220, 85-96, 727-757) without materialising the normalised activations.  Replaces every nn.Linear / Conv3d-as-GEMM: HF:116-122 (patch embed), HF:219-220,85-96,
 * 141-151 (ViT qkv/proj/MLP/merger), HF:630-633,545-553 (LLM), padt.py:189 (vis_proj), padt_decoder.py:15-18,82-86,
 * 142-184 (decoder projections, MLPs, heads). */
int padt_gemm_bf16(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C, long ldc,
                   const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32, const void* row_scale);
/* C = rope(row_scale[m] * (A · W^T) + bias), bf16: rotate-half RoPE (HF:160-171) of output columns [0, rope_cols) fused into the
 * epilogue.  Those columns must be PAIR-INTERLEAVED per head — column 2i / 2i+1 of a head = its d = i / i + head_dim/2 —
 * which the caller gets by permuting W's rows (and bias) once at load; q·k scores are invariant under the common
 * permutation.  rope_cos / rope_sin: fp32 [M][ld_cs], column i < head_dim/2 = angle of pair i.  Replaces
 * {qkv Linear → apply_rotary_pos_emb_vision} of the ViT block (HF:219-220,160-171) without the extra pass over q and k. */
int padt_gemm_rope_bf16(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C, long ldc,
                        long M, long N, long K, const void* row_scale, const void* rope_cos, const void* rope_sin, long ld_cs,
                        long rope_cols, int head_dim);
/* fp32 RESIDUAL STREAM: X32[M,N] (fp32, in place) += A[M,K] · W[N,K]^T + bias, and Xb = bf16(X32) (row-major mirror, ldxb % 8 == 0; may be
 * null).  The residual adds of a ViT block (HF:318-320: x + attn.proj(..), x + mlp.down_proj(..)) and of an LLM layer at prompt length
 * (HF:741,757) with the stream carried in fp32 between kernels — what the reference's fp32 CPU path does — while the same epilogue emits the
 * bf16 A operand of the next projection.  Every tile kernel of padt_gemm_bf16 serves it. */
int padt_gemm_resid32(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* X32, long ldx, void* Xb,
                      long ldxb, long M, long N, long K);
/* Dispatch knobs of the 256-row tile kernel (tests force every tile variant; tools A/B them): mode256 0 off / 1 auto / 2 forced,
 * mf 0 auto / 2..4 tile height in 64-row units, peel 0 never / 1 cost model / 2 always, colsplit 0 never / 1 cost model / n columns,
 * group_m rasterisation patch height.  -1 keeps a field.  The defaults are read once from PADT_GEMM256 / PADT_GEMM_MF / PADT_GEMM_PEEL /
 * PADT_GEMM_COLSPLIT / PADT_GEMM_GROUP_M when the library loads.  Process-wide, not thread-safe (a test / tuning surface). */
int padt_gemm_knobs(int mode256, int mf, int peel, int colsplit, int group_m);
/* Measurement surface (bench.py's in-situ roofline): while a slot array (capacity pairs of uint64, initialised to {~0, 0} by the caller) is
 * registered, every tile-GEMM call (M > 64) takes the next slot and its kernels record {first block start, last block end} in 100 MHz
 * wall-clock ticks — no event packets, no serialisation of the stream.  slots = NULL stops.  Returns the number of calls recorded since the
 * previous registration.  Process-wide, not thread-safe. */
long padt_gemm_profile(void* slots_u64, long capacity);
/* out[row] = rsqrt(mean(x[row]^2) + eps), fp32 — the statistics half of a folded RMSNorm (see row_scale above): Qwen2RMSNorm.forward's
 * variance / rsqrt in front of the ViT block's two sub-layers (HF:318-320) and the LLM layer's (HF:727,744); the weight multiply lives in W. */
int padt_row_rstd(void* stream, const void* x, long ldx, void* out_f32, long rows, long D, float eps);

/* Decode-sized (M <= 64) projection with the preceding RMSNorm fused into the prologue:
 * C = epi(rstd(A)[m] * (A · W^T)[m] + bias), rstd = rsqrt(mean(A[m]^2)+eps); W carries the norm weight (W·diag(g), folded
 * at load time).  epilogue 0 or 3 (SwiGLU).  Replaces {input_layernorm → q/k/v_proj} and {post_attention_layernorm →
 * gate/up_proj} of a decode step (HF:727-757). */
int padt_gemm_rmsnorm_bf16(void* stream, const void* A, long lda, float eps, const void* W, long ldw, const void* bias,
                           void* C, long ldc, long M, long N, long K, int epilogue);

/* Decode-step projection over FRAGMENT-PACKED weights, M <= 128 (HBM-bound weight streaming): same math as
 * padt_gemm_bf16 / padt_gemm_rmsnorm_bf16, but W is stored as [N/16][Kp/32][64 lanes][8] so every wave instruction
 * reads 1 KiB contiguous bytes (row-major fragments load 16 rows x 64 B and run the address unit at a quarter rate).
 * Kp = padded K (multiple of 32) of the packed image, N must be a multiple of 16.  epilogue 0 none, 2 += R, 3 SwiGLU;
 * norm_eps >= 0 fuses the preceding RMSNorm as a row scale (folded norm weight), < 0 disables it.  HF:727-757, T = 1.
 * split_k in [2, 8] (epilogue 0 / 2, no norm) spreads K over that many blocks per 16 output columns — for projections
 * with N/16 < #CUs (down_proj: 128 column blocks) — using `workspace` (padt_gemm_splitk_workspace(N, split_k) bytes,
 * private to one stream, first 256*ceil(N/16*4/256) bytes ZERO before the first call; the kernel leaves them zero).
 * split_k <= 1: workspace may be null.
 * act_packed: bit 0 = A, bit 1 = C and R are in the FRAGMENT-PACKED ACTIVATION layout: rows in blocks of 16, element (m, k)
 * of a [rows][ld] matrix at (m/16)*16*ld + ((k/8)*16 + m%16)*8 + k%8, buffers sized for whole 16-row blocks — the x
 * fragment of a K-step is then 1 KiB contiguous per wave instruction (row-major: 16 rows x 64 B, quarter-rate address
 * unit, the limiter when 32 rows decode together).  padt_pack_rows converts either way. */
long padt_gemm_splitk_workspace(long N, int split_k);
int padt_gemm_packed_bf16(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* bias, void* C,
                          long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps,
                          int split_k, void* workspace, int act_packed);
int padt_pack_rows(void* stream, const void* src, long ld_src, void* dst, long ld_dst, long M, long K, int to_packed);
/* padt_gemm_packed_bf16 over FP8 weights (BASELINE configs[4]: PaDT_Pro_7B, "fp8 MFMA weight path"): Wq = OCP e4m3 bytes stored
 * [N/16][Kp/64][64 lanes][16 B] — a lane's 16 bytes = its 8 elements of K-step 2t followed by its 8 elements of K-step 2t+1, so one
 * 1-KiB wave load feeds two 16x16x32 MFMA K-steps and the decode step streams half the weight bytes; bytes are converted to bf16
 * fragments in registers (exact), fp32 accumulation, scales fp32 [N] (one per weight row) applied to the accumulator before the
 * bias: C = epi(rstd?(A) * scale[n] * (A · Wq^T) + bias).  Kp % 64 == 0, N % 16 == 0.  HF:727-757 at T = 1 with W ≈ scale * Wq. */
int padt_gemm_packed_fp8(void* stream, const void* A, long lda, const void* Wq, long Kp, const void* scales, const void* bias, void* C,
                         long ldc, const void* R, long ldr, long M, long N, long K, int epilogue, float norm_eps, int split_k,
                         void* workspace, int act_packed);
/* fp8 x fp8 MFMA GEMM at prompt length (BASELINE configs[4]: the 7B "fp8 MFMA weight path"; v_mfma_f32_16x16x128_f8f6f4, twice the bf16 MFMA
 * rate): A8 = OCP e4m3 activations [M][K] with one fp32 scale per row (padt_quant_rows_fp8), W8 = e4m3 weights [N][K] in nn.Linear layout with
 * one fp32 scale per weight row.  acc = A8 · W8^T in fp32 (the fp8 products are exact), then by `epilogue`:
 *   0: C (bf16)  = row_scale[m] * col_scale[n] * acc + bias          2: X32 (fp32, in place) += row_scale * col_scale * acc; Xb = bf16(X32)
 *   3: SwiGLU over gate16 | up16 interleaved weight rows → C (bf16, N / 2 columns)
 * K % 128 == 0, N % 256 == 0, lda / ldw % 16 == 0, 16-byte aligned operands.  Same phase-pipelined 256-row tile kernel as padt_gemm_bf16
 * (the LDS image of a 128-byte K-tile row is identical).  HF:630-633,545-553 (q/k/v/o, gate/up/down Linear) with W ≈ col_scale * W8 and
 * x ≈ row_scale * A8. */
int padt_gemm_fp8(void* stream, const void* A8, long lda, const void* W8, long ldw, const void* row_scale, const void* col_scale,
                  const void* bias, void* C, long ldc, void* X32, long ldx, void* Xb, long ldxb, long M, long N, long K, int epilogue);
/* x8[row] = e4m3(x[row] / s) with s = the smallest power of two >= amax(row) / 448; row_scale[row] = s, or s * rsqrt(mean(x^2) + eps) when
 * norm_eps >= 0 (the row feeds a projection whose RMSNorm weight is folded into W: HF:727,744).  bf16 in, bytes out. */
int padt_quant_rows_fp8(void* stream, const void* x, long ldx, void* x8, long ld8, void* row_scale_f32, long rows, long K, float norm_eps);
/* Decode-step residual projection over the fp32 residual stream (o_proj / down_proj, HF:741,757 at one token per row):
 * X32[M,N] (fp32 row-major, in place) += scale?[n] * (A · W^T); Xb = bf16(X32) in the fragment-packed activation layout (required;
 * ldxb % 8 == 0) = the A operand of the next projection.  Wp bf16 fragment-packed, or with scales != null the fp8 image.  a_packed: A is
 * in the fragment-packed activation layout.  split_k / workspace as padt_gemm_packed_bf16. */
int padt_gemm_packed_resid32(void* stream, const void* A, long lda, const void* Wp, long Kp, const void* scales, void* X32, long ldx,
                             void* Xb, long ldxb, long M, long N, long K, int split_k, void* workspace, int a_packed);

/* ---- attention ------------------------------------------------------------------------------------------------------
 * Varlen flash attention, fp32 online softmax, non-causal or causal (bottom-right aligned), GQA by head index.
 * q: token t head h at q + t*ldq + h*head_dim (k, v likewise with kv head h / (n_heads/n_kv_heads)).
 * Replaces flash_attn_varlen_func at padt_decoder.py:55 and in HF's ViT (HF:225-291) / LLM prefill (HF:641-689).
 * rope_cos / rope_sin (nullable, fp32 [token][ld_cs], first head_dim/2 columns): fuse the rotate-half RoPE of q and k
 * (HF:160-171) into the kernel — self-attention only (cu_q == cu_k), non-causal, max_seqlen_q < 256 (ViT window layers). */
int padt_attn_varlen(void* stream, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o,
                     long ldo, const int* cu_q, const int* cu_k, int nseg, int max_seqlen_q, int n_heads,
                     int n_kv_heads, int head_dim, float scale, int causal, const void* rope_cos, const void* rope_sin,
                     long ld_cs);
/* Single-token decode attention over the KV cache (K row-major [B][Hkv][S_max][D], V transposed [B][Hkv][D][S_max]),
 * split over 64-key chunks + combine.  lens[b] = valid keys incl. the token just appended; max_len bounds them.
 * Replaces the Lq==1 case of HF:641-689 with DynamicCache.  Workspace: padt_decode_attn_workspace() bytes, private to
 * one stream (partial O / (m, l) of the splits). */
long padt_decode_attn_workspace(int batch, int n_kv_heads, int head_dim, int s_max);
int  padt_decode_attn(void* stream, const void* q, const void* k_cache, const void* vt_cache, const int* lens, void* out,
                      void* workspace, int batch, int n_heads, int n_kv_heads, int head_dim, int s_max, int max_len,
                      float scale);

/* Decode-step attention with mRoPE + KV-cache append fused in (T = 1).
 * rope_cs = this step's fp32 (cos, sin) table [B][head_dim/2][2] from padt_rope_table; slot[b] = append index (keys
 * visible afterwards = slot[b]+1); out_packed: write `out` in the fragment-packed activation layout (see padt_gemm_packed_bf16).
 * cache_packed = 0: row-major K / transposed V caches (above), split attention over 64-key chunks + a combine launch, workspace as
 *   padt_decode_attn_workspace.
 * cache_packed = 1 (round 6; head_dim 128): FRAGMENT-PACKED caches — K [B][Hkv][S_max/16][D/32][64 lanes][8] (lane (frow, fq) of tile
 *   (s16, kk): K[16 s16 + frow][32 kk + 8 fq ..+8)), V^T [B][Hkv][D/16][S_max/32][64 lanes][8] (lane (frow, fq) of tile (i, ks):
 *   V[32 ks + 4 fq + e][16 i + frow], e < 4, then the same keys + 16) — every wave-wide load is 1 KiB contiguous — and ONE launch: a block
 *   of 8 waves per (kv head, sample) streams the sample's keys, merges on chip and writes the output rows; no workspace (may be null).
 *   11.2 us per layer at 64 rows against 18.7 (profiles/r06_decode_attn_v3.log).  Same results up to one 16-bit rounding (different
 *   merge order).  padt_llm_qkv_post writes the same images when given cache_packed = 1.  While Hkv x batch <= 128 a (kv head, sample) is
 *   given to TWO blocks: both score all keys, each multiplies half of the d-tiles with its half of V^T and stores that half of the output columns
 *   (all 256 CUs stream KV: 13.3 → 12.0 us at 64 rows x 2 kv heads with rotating caches, 8.1 → 6.9 at 8 rows); bit-identical to one block.  cache_packed = 2 / 3 force one / two blocks.
 * HF:557-599, 641-689, 665-666. */
int padt_decode_attn_rope(void* stream, const void* qkv, long ld_qkv, const void* rope_cs, const int* slot, void* k_cache,
                          void* vt_cache, void* out, void* workspace, int batch, int n_heads, int n_kv_heads, int head_dim,
                          int s_max, int max_len, float scale, int out_packed, int cache_packed);
/* rope_cs[b][d] = (cos, sin)(pos3[axis(d)][b] * inv_freq[d]) with mRoPE sections (sec0, sec1, rest).  HF:525-538,589-595. */
int padt_rope_table(void* stream, const int* pos3, const void* inv_freq, void* rope_cs, int batch, int head_dim, int sec0,
                    int sec1);

/* ---- row kernels ---------------------------------------------------------------------------------------------------- */
/* y = act(w * (x [+ add[row/add_div]]) * rsqrt(mean(.^2)+eps)), act: 0 none, 1 exact GELU.
 * HF:74-79; padt_decoder.py:71-74,143,156,168-172,220. */
int padt_rmsnorm(void* stream, const void* x, long ldx, const void* add, long ld_add, int add_div, const void* w, void* y,
                 long ldy, long rows, long D, float eps, int act);
/* nn.LayerNorm (vis_norm), padt.py:121,188. */
int padt_layernorm(void* stream, const void* x, long ldx, const void* w, const void* b, void* y, long ldy, long rows,
                   long D, float eps);
/* in-place rotate-half rotary over n_heads consecutive heads; fp32 cos/sin tables [T][ld_cs] (first head_dim/2 cols).
 * HF:160-171 (ViT q,k), padt_decoder.py:38-51 (decoder image side). */
int padt_rope_half(void* stream, void* x, long ldx, const void* cos_t, const void* sin_t, long ld_cs, long T, int n_heads,
                   int head_dim);
/* dst[i] = src[idx[i]] (bf16 / f32 rows).  padt.py:70-75,103-104 (window order), padt.py:365-373 (per-object copies). */
int padt_gather_rows(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst, long n, long D);
int padt_gather_rows_f32(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst, long n, long D);
/* dst[idx[i]] = src[i], fp32 rows, distinct indices: the KV-cache update of a decode step (HF:641-689) for the fp32 cache of the reference-precision mode. */
int padt_scatter_rows_f32(void* stream, const void* src, long ld_src, const int* idx, void* dst, long ld_dst, long n, long D);
/* y = a + b[row % b_rows].  padt_decoder.py:30-31 (additive positional queries), :202 (vp_embedding). */
int padt_add_rows(void* stream, const void* a, long lda, const void* b, long ldb, long b_rows, void* y, long ldy, long n,
                  long D);
/* y = bf16(scale * x), fp32 → bf16 with zero-padded row tail (pixel_values.type(visual.dtype), padt.py:184: scale 1; the first mirror of an
 * fp32 residual stream: scale = padt_stream_scale()). */
int padt_cast_f32_bf16(void* stream, const void* x, long ldx, void* y, long ldy, long rows, long D, long D_pad, float scale);
/* bf16 → fp32 rows (token embeddings entering the fp32 residual stream; padt.py:212-219 feed HF:790-872). */
int padt_cast_bf16_f32(void* stream, const void* x, long ldx, void* y, long ldy, long rows, long D);
/* y(bf16) = w * x * rsqrt(mean(x^2)+eps) for fp32 rows x: the norms that read the fp32 residual stream (ViT merger ln_q HF:141-148,
 * LLM final norm HF:867). */
int padt_rmsnorm_f32(void* stream, const void* x_f32, long ldx, const void* w, void* y, long ldy, long rows, long D, float eps);
/* flags[row / rows_per_flag] |= 1 when a row of x holds +-inf or NaN; kind 0 fp32, 1 bf16, 2 fp16; rows of whole 16-byte vectors.  The
 * fp16-operand safety net (round 5): an overflow of an un-normalised fp16 tensor upstream (SwiGLU hidden HF:85-96,545-553, q / k / v
 * HF:630-633, merger hidden HF:141-151) arrives as inf / NaN in the rows the product path checks with this — ViT output rows
 * (padt.py:99-104), prototypes (padt.py:187-191), the post-norm hidden rows of the prompt pass and of every decode step (padt.py:732-737) —
 * and is reported (PaDTHipError) or answered by a re-run on the bf16 instantiation (operands="auto") instead of returned. */
int padt_check_finite(void* stream, const void* x, long ldx, long rows, long cols, int kind, int* flags, long rows_per_flag);
/* in-place fp32 sigmoid (bbox head, padt_decoder.py:164). */
int padt_sigmoid_f32(void* stream, void* x, long n);
/* inputs_embeds from the two-pointer table [E ‖ proto] + image-embed scatter.  padt.py:193-219, 226-229.
 * err_flag (optional) is set to 1 when an id falls outside the table (padt.py:203 assert). */
int padt_embed_tokens(void* stream, const long* ids, const int* img_index, const void* embed_table, const void* proto,
                      const void* image_embeds, void* out, long T, long vocab, long n_proto, long D, int* err_flag);
/* mRoPE on q,k + KV-cache append (K row-major, V transposed) (+ packed roped K for prefill attention).
 * HF:557-599 (apply_multimodal_rotary_pos_emb), HF:665-666 (cache update), padt.py:256-277 (positions). */
int padt_llm_qkv_post(void* stream, const void* qkv, long ld_qkv, const int* pos3, const int* sample, const int* slot,
                      const int* lens, const void* inv_freq, void* q_out, long ld_q, void* k_pack, long ld_kp,
                      void* k_cache, void* vt_cache, long T, int n_heads, int n_kv_heads, int head_dim, int s_max,
                      int sec0, int sec1, int cache_packed);
/* PaDT mask head tail: per-patch 4x4 dot with the object's mask token, scattered to (n_obj, 4H, 4W) fp32.
 * padt_decoder.py:241-274. */
int padt_mask_scatter(void* stream, const void* e2, long ld_e2, const void* mask_tok, long ld_tok, const int* cu_patch,
                      const int* obj_w, void* masks_f32, int n_obj, long total_patches, int Hm4, int Wm4, int dm);

/* ---- split-precision ("hp") PaDT decoder (padt_decoder.py:20-276 at fp32-class activation precision) -------------------
 * The north star's 1e-3 box / mask-logit tolerance is not reachable with bf16 activation storage between the decoder's kernels
 * (tests/study_decoder_precision.py: each stored tensor of the query path costs 1-2.5e-3 by itself).  The hp decoder keeps
 * fp32 residual streams and hands every GEMM its A operand as a bf16 (hi, lo) pair, hi = bf16(x), lo = bf16(x - hi), stored
 * [hi(K) | lo(K)] per row against a weight image [W | W]: padt_gemm_bf16_ex below IS padt_gemm_bf16 at K' = 2K (fp32
 * accumulation of hi*W + lo*W), with two extra epilogue options:
 *   resid_f32: epilogue 2 adds an fp32 residual R[M][ldr] (out_f32 must be 1) — the fp32 residual stream, in place allowed;
 *   lo_off:    bf16 output stored as a pair, hi at C[m][n], lo at C[m][lo_off + n] (lo_off >= N, ldc >= lo_off + N); with epilogue 3 (round 6:
 *              the split SwiGLU of precision="reference" — exact expf / division as padt_swiglu_split, no fp32 gate / up rows in between) the
 *              pair is silu(gate) * up, n < N / 2, lo_off >= N / 2. */
int padt_gemm_bf16_ex(void* stream, const void* A, long lda, const void* W, long ldw, const void* bias, void* C, long ldc,
                      const void* R, long ldr, long M, long N, long K, int epilogue, int out_f32, const void* row_scale,
                      int resid_f32, long lo_off);
/* The same split-precision projection for FEW rows (M <= 64: the decode steps of precision="reference"), bound by the weight stream:
 * A rows = [hi(K) | lo(K)] pairs with the lo half a_lo_off elements into the row, W [N][ldw] with its first K columns used and read ONCE
 * (every weight fragment multiplies the hi and the lo fragment of a row block) — pass padt_gemm_bf16_ex's doubled image with ldw = 2K.
 * epilogue 0 / 2: C fp32 = A_hi W^T + A_lo W^T + bias (+ R_f32, in place allowed), c_lo_off = 0; epilogue 3: SwiGLU over [gate16 | up16]-interleaved
 * weight rows (N % 32 == 0), C = bf16 split rows — silu(gate) * up as (hi, lo) pairs, hi at C[m][n], lo at C[m][c_lo_off + n], n < N / 2 (exact expf and
 * division, as padt_swiglu_split).  layout: bit 0 — A_split is the 16-row fragment-packed image of the split rows (padt_pack_rows over their 2K
 * columns: lda = 2K, a_lo_off = 16 K); bit 1 — W is the fragment-packed image of padt_gemm_packed_bf16 ([N/16][ldw/32][64 lanes][8], ldw = K padded to
 * 32, N % 16 == 0): every wave load of either operand is then 1 KiB contiguous.  HF:641-757 at one token per row. */
int padt_gemm_split_rows(void* stream, const void* A_split, long lda, long a_lo_off, const void* W, long ldw, const void* bias, void* C,
                         long ldc, long c_lo_off, const void* R_f32, long ldr, long M, long N, long K, int epilogue, int layout);
/* Row kernel: y0 = f(x), y1 = f(x) + pos[r % pos_rows], f(x) = act(RMSNorm_w(x[idx[r]] + add[r / add_div])), every stage
 * optional (null pointer); x bf16 or fp32 (x_f32); each output off (mode 0), fp32 rows (1) or bf16 split rows (2) laid out
 * [hi(chunk) lo(chunk)] x D/chunk.  padt_decoder.py:71-74 (RMSNorm), :30-31 (+ positional query), :220 (repeat4(low) + high),
 * :168-172 (Linear → RMSNorm → GELU of mask_output_upscaling1), padt.py:365-373 (per-object row gather). */
int padt_norm_split(void* stream, const void* x, long ldx, int x_f32, const int* idx, const void* add_f32, long ld_add, int add_div,
                    const void* w, float eps, int act, const void* pos_f32, long ld_pos, long pos_rows, void* y0, long ld_y0,
                    int y0_mode, void* y1, long ld_y1, int y1_mode, long rows, long D, long chunk);
/* Reference-precision mode of ViT / LLM (round 5; padt_amd/reference.py: the SAME split-precision machinery — fp32 streams, (hi, lo) bf16 GEMM
 * operands at K' = 2K — applied to the 68 upstream layers, so that every float output meets the north star's 1e-3):
 * y = split(silu(g) * u) for fp32 rows [g(I) | u(I)] (the SwiGLU of HF:85-96 / :545-553 between two split GEMMs; y rows are [hi(I) | lo(I)]) */
int padt_swiglu_split(void* stream, const void* gu_f32, long ld_gu, long I, void* y_split, long ld_y, long rows);
/* fp32 LayerNorm with bf16 weight and bias → fp32 rows (vis_norm of the prototype projection, padt.py:187-191). */
int padt_layernorm_f32(void* stream, const void* x_f32, long ldx, const void* w, const void* b, float eps, void* y_f32, long ldy, long rows, long D);
/* In-place rotate-half rotary on fp32 rows (padt_decoder.py:38-51, flash-attn apply_rotary_emb, non-interleaved). */
int padt_rope_half_f32(void* stream, void* x, long ldx, const void* cos_t, const void* sin_t, long ld_cs, long T, int n_heads,
                       int head_dim);
/* fp32 varlen attention (padt_decoder.py:52-58): q/k/v fp32 rows with the heads contiguous, exact expf softmax,
 * output as split rows (chunk as above) for the out-projection.  head_dim 32 / 64 / 80 / 128.  Round 5 (the reference-precision LLM, HF:641-689):
 * kv_group (q head h reads kv head h / kv_group; 1 = the decoder's), causal (bottom-right aligned mask of the prompt pass; 0 = the decoder's),
 * len_k (nullable: per-segment key counts for segments at fixed strides cu_k[s] with room to grow — the fp32 KV cache of the decode steps). */
int padt_attn_f32(void* stream, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out_split, long ldo,
                  long chunk, const int* cu_q, const int* cu_k, int nseg, int max_seqlen_q, int max_seqlen_k, int n_heads,
                  int head_dim, float scale, int kv_group, int causal, const int* len_k);
/* The same attention on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums — an fmaf chain — at the fp32
 * vector peak, VALU free for the softmax): same arguments and semantics, results equal up to fp32 summation order; head_dim 80 or 128.
 * The reference-precision mode's ViT windows / full layers (HF:211-245), causal GQA prompt pass and decode steps over the fp32 cache
 * (HF:641-689); with kv_group > 1 and fewer than 16 queries per segment the G heads of a group share one tile and the block's four
 * waves split the keys. */
int padt_attn_f32_mfma(void* stream, const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* out_split, long ldo,
                  long chunk, const int* cu_q, const int* cu_k, int nseg, int max_seqlen_q, int max_seqlen_k, int n_heads,
                  int head_dim, float scale, int kv_group, int causal, const int* len_k);
/* padt_mask_scatter on fp32 e2 / mask tokens. */
int padt_mask_scatter_f32(void* stream, const void* e2, long ld_e2, const void* mask_tok, long ld_tok, const int* cu_patch,
                          const int* obj_w, void* masks_f32, int n_obj, long total_patches, int Hm4, int Wm4, int dm);

/* ---- VRT head + greedy bookkeeping ------------------------------------------------------------------------------------
 * logits = hidden · [embed_table ‖ proto]^T through two base pointers, -inf outside text ∪ own patch rows, per-block
 * (max, argmax) partials; optional dense fp32 logits.  padt.py:292-301.  mode_table/step: scripted logits-processor
 * slot for synthetic weights (0 free, 1 text rows, 2 own VRT rows, 3 force EOS), padt.py:717.
 * embed_table_packed (nullable): fragment-packed copy of embed_table ([vocab/16][D/32][64][8], as padt_gemm_packed_bf16's
 * weights); when given, `hidden` must be in the 16-row fragment-packed activation layout (ldh = D) and text rows are
 * streamed as 1 KiB contiguous wave loads (0.62 GB per step is the decode step's largest single read).
 * gen_cfg (nullable, DEVICE memory: {float repetition_penalty; int eos[4]; sampling fields, see padt_sample_token}) + seen (nullable, device bitmap
 * [batch][seen_words] of 32-bit words, bit r = table row r already occurs in that sample's input_ids): HF's
 * RepetitionPenaltyLogitsProcessor (score < 0 ? score*p : score/p) fused in front of the mask/arg-max, padt.py:570-580,717.
 * Kept in device memory so a captured decode graph does not bake the values in. */
long padt_vrt_head_nblk(long vocab, long n_proto);
int  padt_vrt_head(void* stream, const void* hidden, long ldh, const void* embed_table, long vocab, const void* proto,
                   long n_proto, const int* vrt_off, const int* mode_table, const int* step, void* logits_f32,
                   long ld_logits, void* part_val, void* part_idx, long batch, long D, int eos,
                   const void* embed_table_packed, const void* gen_cfg, const void* seen, long seen_words);
/* argmax reduction (ties → lowest id), pad/EOS bookkeeping, token append, hidden-row stash, slot/len/position/step
 * advance — all on device.  padt.py:745-757, 732-737.  gen_cfg / seen as in padt_vrt_head: extra EOS ids stop a row too
 * (generation_config's eos_token_id list), and the chosen token's bit is set in the row's seen bitmap. */
int  padt_greedy_step(void* stream, const void* part_val, const void* part_idx, long nblk, long batch, long D, int eos,
                      int pad, long t_max, int* unfinished, long* tokens_out, long* cur_tok, int* step, int* slot,
                      int* lens, int* pos3, const void* hidden, void* hidden_buf, int advance, const void* gen_cfg,
                      void* seen, long seen_words);
/* Sampling branch (padt.py:740-743 multinomial over softmax of the warped scores): one token per row drawn from the fp32 logits
 * padt_vrt_head wrote (logits_f32), after HF's Temperature → TopK → TopP warpers (generation/logits_process.py) with the
 * parameters in gen_cfg (DEVICE: {float penalty; int eos[4]; int do_sample; unsigned seed; float temperature; int top_k;
 * float top_p; int pad[2]}); exact top-k by radix select, nucleus over the sorted survivors (top_p < 1 needs 0 < top_k <= 1024),
 * Gumbel-max draw keyed by (seed, *step, row, index) so a captured decode graph draws fresh numbers every replay.  Writes one
 * (value, index) pair per row in padt_greedy_step's partial layout (nblk = 1). */
int  padt_sample_token(void* stream, const void* logits_f32, long ld_logits, long n_rows_table, const void* gen_cfg, const int* step,
                       void* part_val, void* part_idx, long batch);
/* Arg-max of fp32 score rows → one (value, index) pair per row in padt_greedy_step's partial layout (nblk = 1), ties → lowest index
 * (torch.argmax, padt.py:745): the selection of the HOOKED decode loop, where caller-supplied logits processors (padt.py:717) have rewritten
 * the rows padt_vrt_head wrote. */
int  padt_argmax_rows_f32(void* stream, const void* scores_f32, long ld, long n_cols, void* part_val, void* part_idx, long batch);
/* The synchronising half of generate() (padt.py:745-757 stop rule, :203 table assert, the range guard's flags) in ONE launch + one small
 * D2H copy: out[0] = *err, out[1] = any(unfinished[0..n_rows)), out[2..] = nf_rows[n_rows], nf_batch[n_batch], first_eos[n_rows] — the first
 * step t < done at which a row's token is an EOS id (eos or gen_cfg's list), -1 if none.  out holds 2 + 2 n_rows + n_batch int32. */
int  padt_collect_summary(void* stream, const int* err, const int* unfinished, const int* nf_rows, const int* nf_batch, long n_rows,
                          long n_batch, const long* tokens, long t_max, long done, int eos, const void* gen_cfg, int* out);
/* sequences[b] = [input_ids[b] (L) | tokens[b][0..n_steps)] (padt.py:751), session-global VRT ids (>= vocab) shifted back by proto_row0. */
int  padt_assemble_sequences(void* stream, const long* input_ids, long ld_ids, long L, const long* tokens, long t_max, long n_steps,
                             long vocab, long proto_row0, long* out, long batch);
/* past_logit_mask (padt.py:196-201,794), one byte per column: c < vocab, or vrt_off[b] <= c - vocab + proto_row0 < vrt_off[b + 1]. */
int  padt_logit_mask(void* stream, const int* vrt_off, long vocab, long table_rows, long proto_row0, void* out_u8, long batch);
/* output_scores=True (padt.py:719-720): dst[*step][0..n) = src[0..n) for the fp32 score rows padt_vrt_head wrote (logits_f32), indexed by the
 * DEVICE step counter (a captured decode graph files every replay's rows under its own step); no-op once *step >= t_max.  n % 4 == 0. */
int  padt_stash_step_f32(void* stream, const void* src_f32, long n, const int* step, long t_max, void* dst_f32);
/* seen[rows[i]] |= bit(ids[i]) for the prompt tokens of a generate call (ids global in the session's table): the `input_ids` HF's
 * RepetitionPenaltyLogitsProcessor gathers over (generation/logits_process.py, reached from padt.py:717) — prompt and padding ids included. */
int  padt_seen_init(void* stream, const long* ids, const int* rows, long n, void* seen, long seen_words);

/* ---- caller-side post-processing --------------------------------------------------------------- */
/* out[o][y][x] = sigmoid(bilinear(masks[o][:src_h[o]][:src_w[o]] → dst_h[o] x dst_w[o], align_corners=False))[y][x] > 0.5,
 * one byte per pixel; masks fp32 logits as returned by vl_decode (src_h = 4*H, src_w = 4*W of pred_mask_valid_hw).
 * up_f32 (optional) receives the up-sampled logits.  eval/evaluation_scripts/utils.py:262, eval/test_demo.py:153. */
int padt_mask_upsample_binarize(void* stream, const void* masks_f32, long ld_obj, long ld_row, const int* src_h,
                                const int* src_w, const int* dst_h, const int* dst_w, void* out_u8, long out_ld_obj,
                                long out_ld_row, void* up_f32, long up_ld_obj, long up_ld_row, int n_obj, int max_dst_h,
                                int max_dst_w);
/* COCO RLE of binarised masks on the device: what the eval loop's `cocomask.encode(np.asfortranarray(mask))` + `rle['counts'].decode()`
 * produce (eval/evaluation_scripts/utils.py:263-264; pycocotools is third-party and absent here: cocoapi common/maskApi.c rleEncode +
 * rleToString restated — column-major runs, zero run first, 5-bit groups + 48).  mask_u8: n_obj binary images [max_h][ld_row] (the
 * output of padt_mask_upsample_binarize), object o uses its leading dst_h[o] x dst_w[o] pixels.  Per object: counts[o][0 .. n_counts[o])
 * (scratch AND result, capacity cap_counts >= h*w + 1 in the worst case; n_counts < 0 when exceeded) and the counts string str[o][0 .. str_len[o])
 * (capacity cap_str; str_len = -1 when exceeded).  With packed_u8 != null the strings are also written back to back into packed_u8 behind
 * offsets[0 .. n_obj] (offsets[n_obj] = total bytes, -1 on a capacity error), so that one small device-to-host copy carries a batch's RLEs
 * instead of its masks.  max_h <= 7680; an object with dst_h > max_h or dst_w > ld_row is reported like a capacity error (n_counts = str_len = -1). */
int padt_mask_rle(void* stream, const void* mask_u8, long ld_obj, long ld_row, const int* dst_h, const int* dst_w, int n_obj, int max_h,
                  int* counts, long cap_counts, void* str_u8, long cap_str, int* n_counts, int* str_len, void* packed_u8, long cap_packed,
                  int* offsets);

/* ---- image front-end tail ------------------------------------------------------------------ */
/* uint8 (H, W, 3) image (already resized; H, W multiples of patch*merge) → (H/patch * W/patch) rows of 3*temporal*patch*patch
 * values in (h/merge, w/merge, merge, merge) block-major patch order, each = lut[c][byte] (fp32 3 x 256 table holding the
 * processor's rescale + normalize), fp32 or bf16.  HF Qwen2-VL image processor: _preprocess / patchify. */
int padt_patchify_normalize(void* stream, const void* img_u8, int H, int W, const void* lut_f32, void* out, long ld_out,
                            int out_bf16, int patch, int merge, int temporal);
/* One separable pass (horizontal: in_h == out_h, vertical: in_w == out_w) of Pillow's 8-bit ImagingResample over an interleaved
 * (H, W, channels) uint8 image: out = clip8(((1 << 21) + sum_k in[first + k] * kk[k]) >> 22), bounds int32 [out][2] = (first, taps),
 * kk int32 [out][ksize] fixed-point coefficients built on the host as Pillow's precompute_coeffs / normalize_coeffs_8bpc do
 * (padt_amd/preprocess.py) — integer arithmetic, bit-exact with PIL.Image.resize for BICUBIC (HF Qwen2-VL image processor) and
 * LANCZOS (eval/test_demo.py:67-73, eval/evaluation_scripts/utils.py:205-218). */
int padt_resample_pass_u8(void* stream, const void* in_u8, int in_h, int in_w, int channels, void* out_u8, int out_h, int out_w,
                          const int* bounds, const int* kk, int ksize, int horizontal);

/* ---- data-parallel result exchange ---------------------------------------------------------------------------------------------
 * One batch's vl_decode output as ONE fixed-capacity int32 record (floats as bit patterns):
 *   [n, cap, mask_hw, has_mask | sample_idx (cap) | valid_hw (2 cap) | boxes (4 cap) | scores (cap) | mask logits (cap * mask_hw^2)]
 * words = 4 + 8 cap + cap mask_hw^2; zero outside the n objects / the H x W window.  masks_f32 may be null (no mask head).  The records of
 * every rank then move with one RCCL all-gather (the reference gathers per-rank JSONL files on disk: eval/evaluation_scripts/utils.py:
 * 249-266 writes them, eval_refcoco.py / eval_coco.py read all eight). */
int padt_pack_results(void* stream, void* out_i32, long words, int n, int cap, int mask_hw, const int* sample_idx, const long* valid_h,
                      const long* valid_w, const void* boxes_f32, long ld_box, const void* scores_f32, long ld_score, const void* masks_f32,
                      long ld_obj, long ld_row, int H, int W);

#ifdef __cplusplus
}
#endif
#endif /* PADT_HIP_H */
