/* e2k -- C ABI of the MI355X (gfx950) kernels behind the E2-TTS flow-matching transformer hot path.
 *
 * The reference (lucidrains/e2-tts-pytorch) has no FFI: its hot path is Python calling ATen ops.  This
 * header is the boundary a maintainer would bind instead (ctypes stub in INTEGRATION.md); every entry
 * point cites the reference call site(s) it replaces (file:line into /root/reference/e2_tts_pytorch).
 *
 * Conventions
 *   - every function returns 0 on success, E2K_ERR_* (or 1000 + hipError_t) otherwise; no exceptions;
 *   - the COMPUTE entry points keep no state between calls, are re-entrant and allocate or free no device memory.
 *     The launch-plan entry points (e2k_plan_*, bottom of this header) are the one exception: a recording in progress is
 *     per-thread state (every compute call made by that thread is appended to it), and finished plans live in a
 *     process-wide registry behind a mutex, addressed by the integer handle e2k_query_plan_end returns, until e2k_plan_free;
 *   - the caller owns every buffer; all pointers are device pointers (HBM) unless stated otherwise;
 *   - `stream` is a hipStream_t (NULL = default stream); kernels are only enqueued, never synchronised;
 *   - "bf16" buffers hold raw bfloat16 bits (uint16_t); leading dimensions (ld*) are in ELEMENTS;
 *   - 16-byte vector access: bf16 base pointers must be 16-B aligned and ld* multiples of 8;
 *   - one tuning instrument is read from the environment once per process: E2K_GEMM_T256_MIN (smallest number of 256 x 256 output
 *     tiles for which the NT GEMM takes its 256 x 256 kernel; a choice between launch geometries of the same arithmetic).
 */
#ifndef E2K_H
#define E2K_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2K_OK 0
#define E2K_ERR_SHAPE 1   /* unsupported / inconsistent sizes */
#define E2K_ERR_ALIGN 2   /* pointer or leading dimension not vector-aligned */
#define E2K_ERR_ARG 3     /* null pointer / bad flag */

int e2k_version(void);

/* C[M,N] = ((([A1|A2] . B^T) + bias[n]) * colscale[(m / rows_per_batch) * lds + n]) * rowmask[m] + resid[m][n]
 * A1 (M,K1), A2 (M,K2) optional second K-panel (K2 = 0: none), B (N,K1+K2), all bf16 row-major.
 * C bf16 (out_f32 = 0) or fp32 (out_f32 = 1; accumulate = 1 adds into C).  bias/colscale fp32, rowmask u8,
 * resid bf16 (any of them NULL = skipped).  K1, K2 multiples of 8.
 * Replaces: nn.Linear in x_transformers.Attention / FeedForward (e2_tts.py:875,881,911,937), the output
 * mask + AdaLNZero gate (e2_tts.py:346-351,913,938), skip_proj on cat(x, skip) (e2_tts.py:895-896),
 * TextAudioCrossCondition on cat(audio, text) + residual add (e2_tts.py:508-513), proj_in/cond_proj_in/to_pred
 * (e2_tts.py:1267-1277,1296) and all of their dgrad GEMMs. */
int e2k_gemm_nt_bf16(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                     const void* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int accumulate,
                     int M, int N, const float* bias, const float* colscale, int64_t lds,
                     int rows_per_batch, const uint8_t* rowmask, const void* resid, int64_t ldr, int flags,
                     float* ws, int64_t ws_bytes, void* stream);
/* ws: optional fp32 scratch (e2k_query_gemm_nt_ws_bytes() bytes cover every shape; NULL = never split).  When the tile
 * count leaves a partial last round of the 512 resident workgroups, those tiles are split over K into partials in ws
 * and finished by a second small kernel.  One ws per stream: launches on the same stream serialise on it. */
int e2k_query_gemm_nt_ws_bytes(void);
/* The same product with TWO outputs: columns [0, nsplit) of [A1|A2] . B^T (+ resid) go to C (M, nsplit), columns
 * [nsplit, N) (+ resid2) to C2 (M, N - nsplit); bf16 outputs, nsplit a multiple of 256, resid and resid2 both given or both
 * NULL.  TextAudioCrossCondition (e2_tts.py:503-513) computes text_to_audio(cat(audio, text)) + audio and
 * audio_to_text(cat(audio, text)) + text from the SAME concatenated operand: one launch over the stacked weight rows
 * [text_to_audio; audio_to_text] instead of two (the same for the two halves of its dgrad). */
int e2k_gemm_nt2_bf16(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                      const void* B, int64_t ldb, int M, int N, int nsplit, void* C, int64_t ldc, void* C2, int64_t ldc2,
                      const void* resid, int64_t ldr, const void* resid2, int64_t ldr2, int flags,
                      float* ws, int64_t ws_bytes, void* stream);
#define E2K_GEMM_NO_GLDS 1   /* flags: stage operands through VGPRs instead of global_load_lds (A/B benchmarking) */
#define E2K_GEMM_PROBE_NO_LOADS 4 /* flags: bottleneck probe, K loop without its global loads (WRONG results) */
#define E2K_GEMM_PROBE_NO_MATH 8  /* flags: bottleneck probe, K loop without its LDS reads + MFMAs (WRONG results) */
#define E2K_GEMM_T256 128        /* flags: 256 x 256 x 64 tile, 8-wave 8-phase kernel for EVERY shape (default: shapes with >= 64 such tiles; E2K_GEMM_T256_MIN overrides) */
#define E2K_GEMM_NO_T256 256     /* flags: never use the 256 x 256 kernel (A/B) */
#define E2K_GEMM_NO_STAGE 64     /* flags: 256 x 256 kernel stores its C tile straight from the accumulator registers (16 rows x 32 bytes per wave instruction) instead of through LDS in whole-line row segments (A/B) */
#define E2K_GEMM_NO_SPLIT 16     /* flags: never split remainder tiles over K (overrides E2K_GEMM_SPLIT) */
#define E2K_GEMM_SPLIT 512       /* flags: cut the tiles of the last, partial round into K ranges + a fix-up launch.  Pays when the GEMM has the chip to itself (+2-3 % alone-timed); off by default since round 6: next to the launch lanes it costs the step 0.5 ms */
#define E2K_GEMM_TEST_SLOTS8 32  /* flags: pretend the chip holds 8 workgroups (lets small shapes exercise the remainder split in tests) */

/* C[N,K] += A[M,N]^T . B[M,K]  (weight gradients; C fp32, A = dY, B = X, bf16).  The token dimension M is
 * split over `splits` workgroups per tile (0 = choose); partial tiles go to `ws` and are combined by a reduce kernel.
 * use_tr: 0 = the general kernel with plain 16-bit LDS gathers; 1 = the library chooses per shape between the 128 x 128
 * and the 256 x 256 kernel (both read fragments with ds_read_b64_tr_b16); 2 = always 128 x 128; 3 = 256 x 256 wherever
 * it can run (M a multiple of 64).  Same results up to summation order.
 * colsum (optional, fp32 [N]): colsum[n] += sum_m A[m][n] for n >= cs_from (even) -- the bias gradient of the same
 * Linear (e.g. FeedForward proj bias, e2_tts.py:937), computed in the same pass over dY by one extra MFMA per 16
 * columns with an all-ones operand. */
int e2k_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                     int M, int N, int K, int splits, int use_tr, float* ws, float* colsum, int cs_from, void* stream);
/* The same weight gradient with column-split operands, in ONE launch of the 256 x 256 kernel:
 *   C[N1+N2, K1+K2] += [A1 | A2]^T . [B1 | B2]        (A2 / B2 may be NULL with N2 / K2 = 0)
 * TextAudioCrossCondition's two Linears both read cat(audio, text) and their gradients come from (d audio, d text): the
 * four blocks of the (D + Dt, D + Dt) weight gradient are one product of the concatenations (e2_tts.py:494-513), and the
 * skip projection's (D, 2D) gradient is dX^T . [x | skip] (e2_tts.py:895-896) -- neither concatenation is materialised.
 * N1 and K1 multiples of 256 (a tile lies in one source), M a multiple of 64; ws = e2k_query_gemm_tn_ws_floats(M, N1+N2,
 * K1+K2, splits, 3) floats. */
int e2k_gemm_tn_dual_bf16(const void* A1, int64_t lda1, int N1, const void* A2, int64_t lda2, int N2,
                          const void* B1, int64_t ldb1, int K1, const void* B2, int64_t ldb2, int K2,
                          float* C, int64_t ldc, int M, int splits, float* ws, void* stream);
/* A GROUP of up to 8 independent weight gradients with the same token count M in one launch of the 256 x 256 kernel:
 *   C_i[N_i, K_i] += A_i[M, N_i]^T . B_i[M, K_i]     (colsum_i optional: the bias gradient of the same dY_i, as above)
 * The weight gradients of one layer (attention out / qkv / FeedForward 1 and 2, audio and text stream: 4-128 tiles of
 * 256 x 256 each) fill the chip together instead of one after the other; nothing on the backward chain reads them before
 * the optimizer.  M a multiple of 64; ws = e2k_query_gemm_tn_group_ws_floats(...) floats (partial tiles in fragment order
 * over the group's tile list); splits = 0 lets the library choose one split count for the group. */
typedef struct {
    const void* A; int64_t lda; const void* B; int64_t ldb; float* C; int64_t ldc;
    int32_t N, K;
    float* colsum; int32_t cs_from; int32_t reserved;
} e2k_tn_problem;
int e2k_gemm_tn_group_bf16(const e2k_tn_problem* problems, int n, int M, int splits, float* ws, void* stream);
int64_t e2k_query_gemm_tn_group_ws_floats(const e2k_tn_problem* problems, int n, int M, int splits);
/* upper bound on the token-dimension splits the call above uses for (M, N, K, splits), whichever kernel it selects
 * (use_tr = 1 means "the library chooses per shape", not "transposing reads"); when it is > 1 the caller passes
 * ws = scratch of e2k_query_gemm_tn_ws_floats(...) floats (partial tiles are stored there and combined afterwards).  The
 * exact count for a given use_tr mode: e2k_query_gemm_tn_splits_mode. */
int e2k_query_gemm_tn_splits(int M, int N, int K, int splits);
/* the same for a given use_tr mode: 1 = the library chooses the kernel per shape, 2 = always the 128 x 128 x 64 kernel,
 * 3 = the 256 x 256 x 64 8-phase kernel wherever it can run (M a multiple of 64) */
int e2k_query_gemm_tn_splits_mode(int M, int N, int K, int splits, int use_tr);
/* floats of ws the call needs for (M, N, K, splits, use_tr): the partial tiles of the splits are stored in the MFMA fragment
 * order of the selected kernel (contiguous 1-KB runs per wave store), padded to whole tiles; 0 = nothing is split */
int64_t e2k_query_gemm_tn_ws_floats(int M, int N, int K, int splits, int use_tr);

/* ---- hyper-connections (hyper_connections.HyperConnections; reference call sites e2_tts.py:870-882,900-939) ----
 * Streams are stored token-major: X[token][4][D] bf16.  coef: per-token fp32 record (e2k_query_hc_coef_width()
 * floats: a[4][5], b[4], pre-tanh dots[4][6], 1/|r_s|[4]).  D in {128,256,512,768,1024,1536,2048}.
 *
 * forward:  r = has_depth ? Xin + b_prev * yprev : Xin            (depth connection of the previous instance)
 *           has_width ? (bin, Mout, coef) = width(r) : Mout = r   (width connection of this instance)   */
int e2k_query_hc_coef_width(void);
int e2k_query_hc_bwd_blocks(int Mtok, int D); /* rows of `partial` the backward needs */
int e2k_query_hc_partial_stride(int D);       /* floats per row of `partial` */
int e2k_hc_fwd(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
               float* coef, const float* static_beta, const float* static_alpha,
               const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
               const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
               int has_width, void* stream);
/* The width connection with the (Adaptive)RMSNorm that follows it in every branch of the backbone (e2_tts.py:875,881,908-914,926,937:
 * `x, add_residual = hc(x)` then `x = norm(x[, cond])`; x_transformers RMSNorm / AdaptiveRMSNorm) applied while the branch
 * input is in registers: xn (Mtok, D) bf16 = bin / |bin| sqrt(D) (norm_gamma[row / rows_per_batch][:] + gamma_off), rn (Mtok) = 1 / |bin|
 * (NULL: not wanted).  |bin|^2 comes from the Gram matrix of the four streams inside the one reduction round the coefficients need.
 * bin may be NULL (a no-grad forward needs only xn): replaces e2k_hc_fwd(has_width = 1) + e2k_rmsnorm_fwd. */
int e2k_hc_fwd_norm(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
                    float* coef, const float* static_beta, const float* static_alpha,
                    const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
                    const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
                    const float* norm_gamma, int64_t ldg, float gamma_off, int rows_per_batch, void* xn, float* rn, void* stream);
/* backward of the same fused pair.  G = grad wrt Mout (has_width) or wrt the materialised X (!has_width);
 * dbin = grad wrt bin; ycur = this instance's branch output.  Writes dR = grad wrt r (== grad wrt Xin) and,
 * if has_depth, dyprev = sum_s b_prev[s] * dR[s].  Parameter gradients are ACCUMULATED into g_* (fp32).
 * partial: scratch [e2k_query_hc_bwd_blocks(Mtok)][e2k_query_hc_partial_stride(D)] fp32. */
int e2k_hc_bwd(const void* Xin, const void* yprev, const float* coef_prev, const void* G,
               const void* dbin, const void* ycur, const float* coef, void* dR, void* dyprev,
               const float* static_beta, const float* static_alpha, const float* dyn_alpha_fn,
               const float* dyn_alpha_scale, const float* dyn_beta_fn, const float* dyn_beta_scale,
               const float* gamma, float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn,
               float* g_dyn_alpha_scale, float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma,
               float* partial, int Mtok, int D, int has_depth, int has_width, void* stream);
/* has_width = 2 in the call above leaves the per-workgroup partials in `partial` and does NOT launch the reduction into the
 * parameter gradients; this is that second half (same Mtok, D, partial).  Nothing on the backward chain reads these
 * gradients, so a schedule can run it off the chain (the WGRAD launch lane). */
int e2k_hc_bwd_reduce(const float* partial, const float* dyn_alpha_fn, const float* dyn_beta_fn, const float* gamma,
                      float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn, float* g_dyn_alpha_scale,
                      float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma, int Mtok, int D, void* stream);
/* the same for up to 8 hyper-connections in ONE launch (the six of a layer, e2_tts.py:870-872,908-939: their reductions are
 * independent and small).  `items` is a HOST array, copied during the call. */
typedef struct {
    const float* partial; const float* dyn_alpha_fn; const float* dyn_beta_fn; const float* gamma;
    float* g_static_beta; float* g_static_alpha; float* g_dyn_alpha_fn; float* g_dyn_alpha_scale;
    float* g_dyn_beta_fn; float* g_dyn_beta_scale; float* g_gamma;
    int32_t Mtok, D;
} e2k_hc_reduce_item;
int e2k_hc_bwd_reduce_batch(const e2k_hc_reduce_item* items, int n, void* stream);

/* ---- RMSNorm / AdaptiveRMSNorm (x_transformers; e2_tts.py:615,637,645,688,691,729,908,937) ----
 * y[m] = x[m] / max(|x[m]|, 1e-12) * sqrt(D) * (gamma[m / rows_per_batch] + gamma_off);  rn[m] = 1 / max(|x[m]|, 1e-12)
 * gamma fp32 (nb rows, ldg floats apart; dgamma uses the same stride): nb = 1 (plain RMSNorm `g`, gamma_off = 0) or one row per batch element
 * (AdaptiveRMSNorm: to_gamma(cond), gamma_off = 1). */
int e2k_rmsnorm_fwd(const void* x, const float* gamma, int64_t ldg, float gamma_off, int rows_per_batch,
                    void* y, float* rn, int M, int D, void* stream);
/* dx and dgamma (ACCUMULATED, fp32 (nb, D)) */
int e2k_rmsnorm_bwd(const void* dy, const void* x, const float* rn, const float* gamma, int64_t ldg,
                    float gamma_off, int rows_per_batch, void* dx, float* dgamma, int M, int D,
                    void* stream);

/* AdaLN-Zero gate backward (AdaLNZero, e2_tts.py:346-351; the forward multiply is the colscale epilogue of
 * e2k_gemm_nt_bf16):  dao = dy * g[b];  gsum[b][d] += sum_rows dy * y   (y = gated output, g = sigmoid gate) */
int e2k_gate_bwd(const void* dy, const void* y, const float* g, void* dao, float* gsum, int64_t ldg,
                 int M, int D, int rows_per_batch, void* stream);

/* GEGLU (x_transformers.FeedForward(glu=True): `x, gate = proj(x).chunk(2); x * gelu(gate)` + Dropout, exact erf GELU).
 * H (M, 2F) bf16 with row stride ldh; out (M, F).  p_drop = 0 disables dropout; the keep mask is the counter hash
 * rand_u32(seed, stream_id, row, col/2) (e2k_device.h, restated in oracle/dropout_hash.py).  seed_dev (device
 * pointer, may be NULL) overrides `seed` when set: a captured HIP graph then draws a fresh mask on every replay. */
int e2k_geglu_fwd(const void* H, int64_t ldh, void* out, int M, int F, float p_drop, uint32_t seed,
                  const uint32_t* seed_dev, uint32_t stream_id, void* stream);
int e2k_geglu_bwd(const void* dout, const void* H, int64_t ldh, void* dH, int M, int F, float p_drop,
                  uint32_t seed, const uint32_t* seed_dev, uint32_t stream_id, void* stream);

/* FeedForward's first Linear with the GEGLU (+ Dropout) as the GEMM's epilogue (x_transformers.FeedForward(glu=True),
 * e2_tts.py:646,692):  h = A (M,K) . W1 (2F,K)^T + bias;  out (M,F) = h[:, :F] * gelu(h[:, F:]) * keep.
 * One launch instead of e2k_gemm_nt_bf16 + e2k_geglu_fwd: h is rounded to bf16 before the product, so out is
 * e2k_geglu_fwd of the stored H up to the erf approximation of the epilogue (|error| <= 1.5e-7, below bf16 rounding), and H
 * is bit-identical to e2k_gemm_nt_bf16's wherever both sum K in one pass;
 * h is stored to H (M, 2F; what e2k_geglu_bwd reads) unless H is NULL (inference).  The 256 x 256 kernel stages the
 * value rows and the gate rows of W1 as its two B half tiles, which needs F % 128 == 0 and K % 64 == 0:
 * e2k_query_gemm_nt_geglu returns 1 for shapes it takes, other shapes are refused with E2K_ERR_SHAPE.
 * flags / ws / ws_bytes as e2k_gemm_nt_bf16 (remainder split); dropout arguments as e2k_geglu_fwd. */
int e2k_gemm_nt_geglu_bf16(const void* A, int64_t lda, int K, const void* W1, int64_t ldb, const float* bias,
                           void* H, int64_t ldh, void* out, int64_t ldo, int M, int F, float p_drop, uint32_t seed,
                           const uint32_t* seed_dev, uint32_t stream_id, int flags, float* ws, int64_t ws_bytes,
                           void* stream);
int e2k_query_gemm_nt_geglu(int M, int F, int K);
/* FeedForward's second Linear in the backward pass with the GEGLU backward as the GEMM's epilogue (e2_tts.py:646,692,937; replaces
 * e2k_gemm_nt_bf16 for d(act) = dY W2 followed by e2k_geglu_bwd): dY (M, K = dim), W2T (F, K) = the transposed weight, H (M, 2F) the
 * stored pre-activation [u | g] -> dH (M, 2F) = [d(act) keep gelu(g) | d(act) keep u gelu'(g)]; d(act) itself is never written.
 * 256 x 256 kernel only: F % 256 == 0, K % 64 == 0, K >= 256 (e2k_query_gemm_nt_geglu_bwd: 1 = taken and recommended, 2 = taken, but the
 * output has fewer 256 x 256 tiles than e2k_gemm_nt_bf16 asks for before it picks that kernel -- the two-launch pair is the better
 * choice there --, 0 = refused with E2K_ERR_SHAPE); flags / ws / ws_bytes as e2k_gemm_nt_bf16, dropout arguments as e2k_geglu_bwd (same keep mask). */
int e2k_gemm_nt_geglu_bwd_bf16(const void* dY, int64_t ldy, int K, const void* W2T, int64_t ldb, const void* H, int64_t ldh,
                               void* dH, int64_t lddh, int M, int F, float p_drop, uint32_t seed, const uint32_t* seed_dev,
                               uint32_t stream_id, int flags, float* ws, int64_t ws_bytes, void* stream);
int e2k_query_gemm_nt_geglu_bwd(int M, int F, int K);
/* x_transformers.Attention's fused input projection with the rotary embedding of q and k as the GEMM's epilogue (call sites
 * e2_tts.py:875,911; replaces e2k_gemm_nt_bf16 followed by the q / k half of e2k_qkv_post_fwd): A (B*Ntok, K) bf16, W (N, K) with
 * N >= 3 H 64 columns ordered as e2k_qkv_post_fwd's qkvg.  Columns [0, 2 H 64) of A W^T (+ bias) leave the kernel rounded to bf16, rotated
 * by the table cosb / sinb (Ntok, 32) and HEAD-MAJOR at Q / Kh (B, H, Ntok, 64); the other columns (v, gate logits, mix logits) go to
 * C (B*Ntok, ldc) and columns [0, 2 H 64) of C are NOT written.  e2k_qkv_post_fwd is then called with Q = K = NULL.  Bit-identical to
 * the two-launch form.  256 x 256 kernel only: H even, N % 8 == 0, K % 64 == 0, K >= 256 (e2k_query_gemm_nt_qkrot: 1 = taken and
 * recommended, 2 = taken, but the output has fewer 256 x 256 tiles than e2k_gemm_nt_bf16 asks for before it picks that kernel -- the
 * two-launch pair is the better choice there --, 0 = refused with E2K_ERR_SHAPE); never splits a remainder. */
int e2k_gemm_nt_qkrot_bf16(const void* A, int64_t lda, int K, const void* W, int64_t ldb, const float* bias, void* C, int64_t ldc,
                           void* Q, void* Kh, const float* cosb, const float* sinb, int B, int H, int Ntok, int N, void* stream);
int e2k_query_gemm_nt_qkrot(int M, int N, int K, int H);

/* out[n] += sum_m x[m][n]   (bias gradients; x bf16 (M,N), out fp32) */
int e2k_colsum_bf16(const void* x, int64_t ldx, float* out, int M, int N, void* stream);

/* LinearFourierEmbed's activation (e2_tts.py:368-386; Transformer(attn_fourier_embed_input = True), :639,909): the bias-free
 * projection h (M, nf + nrest) is an e2k_gemm_nt_bf16 call; this is  y (M, 2 nf + nrest) = [sin h[:nf] | cos h[:nf] | h[nf:]]
 * and its backward  dh[:nf] = dy[:nf] cos h - dy[nf:2nf] sin h,  dh[nf:] = dy[2nf:].  h is the GEMM's FP32 output (an angle
 * of a few radians rounded to bf16 moves its sine by percents); y, dy, dh bf16; nf / nrest % 8 == 0. */
int e2k_fourier_cat_fwd(const float* h, int64_t ldh, void* y, int64_t ldy, int64_t M, int nf, int nrest, void* stream);
int e2k_fourier_cat_bwd(const void* dy, int64_t ldy, const float* h, int64_t ldh, void* dh, int64_t lddh, int64_t M, int nf,
                        int nrest, void* stream);

/* fp32 master parameters -> bf16 compute shadows (flat, and (R,C) -> transposed (C,R) with row stride ldd) */
int e2k_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int e2k_cast_transpose_bf16(const float* src, void* dst, int R, int C, int64_t ldd, void* stream);
/* every transposed shadow of a module in one launch: desc (DEVICE, n x 6 int64) = {src element offset into `flat`, dst
 * element offset into `flatT`, R, C, ldd, index of the matrix's first 64 x 64 tile}, total_blocks = sum of the tiles */
int e2k_cast_transpose_batch(const float* flat, void* flatT, const int64_t* desc, int n, int total_blocks, void* stream);

/* DepthwiseConv (e2_tts.py:295-328): channels-last x (B,N,C) bf16, mask (B,N) u8 or NULL, w (C,ks) fp32, bias (C):
 *   pre = conv1d(mask * x) + bias ;  y = mask * silu(pre).   ks in {3,7,15,31}, C multiple of 64.  pre may be NULL (only the backward
 * pass reads it). */
int e2k_dwconv_fwd(const void* x, const uint8_t* mask, const float* w, const float* bias, void* pre,
                   void* y, int B, int N, int C, int ks, void* stream);
/* dx, and dw / dbias ACCUMULATED (fp32).  ws: scratch of e2k_query_dwconv_bwd_ws_floats(B, N, C, ks) floats for the
 * per-workgroup (dw, dbias) partials (NULL: global fp32 atomics instead, ~1.3x slower at the cfg3 shapes).
 * split: bit 0 = leave the partials in ws and do not sum them: the caller runs e2k_dwconv_bwd_reduce(ws, ...) with the same
 * B, N, C, ks, split later (off the backward chain); bits 1..: tuning / ablation (2 = no gradient flush,
 * 4 = loads and staging only -- WRONG results; >> 8: workgroups per channel tile and batch) */
int e2k_query_dwconv_bwd_ws_floats(int B, int N, int C, int ks);
int e2k_dwconv_bwd(const void* dy, const void* pre, const void* x, const uint8_t* mask, const float* w,
                   void* dx, float* dw, float* dbias, float* ws, int B, int N, int C, int ks, int split, void* stream);
int e2k_dwconv_bwd_reduce(const float* ws, float* dw, float* dbias, int B, int N, int C, int ks, int split, void* stream);

/* ---- attention (x_transformers.Attention, call sites e2_tts.py:875,911; dim_head = 64) ----
 * qkvg (B*N, ldq) bf16 = fused projection output, columns [q (H*64) | k | v | head-gate logits (H) | value-residual
 * mix logits (H)].  cosb/sinb: rotary table (N, 32) fp32 (theta_j = 10000^(-2j/64), interleaved pairs).
 * vfirst: first layer's un-mixed values (B,H,N,64), NULL on the first layer (then V = v is what later layers use).
 * Outputs head-major Q,K,V (B,H,N,64), transposed QT,KT,VT (B,H,64,Npad; Npad = N rounded up to 64, zero padded; QT, KT and V may
 * be NULL: the forward kernels read V^T, only the backward pass and the later layers' value residual read V),
 * gate = sigmoid(gate logits), mix = sigmoid(mix logits) (B,H,N) fp32.
 * Q = K = NULL (both, and then QT = KT = NULL): q and k were already written by e2k_gemm_nt_qkrot_bf16; only the value path and the
 * gates run.
 * laser_clamp > 0: LASER attention (Transformer(attn_laser = True, attn_laser_softclamp_value = c), e2_tts.py:543-544,641):
 * V / VT hold exp(c tanh(v / c)) of the (mixed) values and v_orig (first layer; may be NULL) receives the values before
 * that map, i.e. what later layers take as `vfirst`.  laser_clamp = 0 (then v_orig = NULL): the default attention. */
int e2k_qkv_post_fwd(const void* qkvg, int64_t ldq, const float* cosb, const float* sinb, const void* vfirst,
                     void* Q, void* K, void* V, void* QT, void* KT, void* VT, float* gate, float* mix,
                     void* v_orig, float laser_clamp, int B, int H, int N, int Npad, void* stream);
/* dqkvg (B*N, ldq) from dQ,dK,dV (B,H,N,64): inverse rotary, value-residual mix backward (dvfirst (B,H,N,64) fp32
 * is ACCUMULATED on later layers and consumed when first_layer = 1), gate / mix logit gradients; laser_clamp as above
 * (dV is then the gradient of the mapped values).  Columns [3 H 64 + H (+ H with vfirst), ldq) of every dqkvg row -- the
 * padding of a rounded-up row stride, read as zero K padding by the dgrad GEMM -- are written as zeros. */
int e2k_qkv_post_bwd(const void* dQ, const void* dK, const void* dV, const float* dgate_pre,
                     const void* qkvg, int64_t ldq, const float* cosb, const float* sinb,
                     const void* vfirst, const float* mix, float* dvfirst, int first_layer, void* dqkvg,
                     float laser_clamp, int B, int H, int N, void* stream);
/* LASER output map between the attention and the head gates (`out = log(out)`, clamped at 1e-20): O = e2k_attn_fwd's
 * un-gated output (B*N, H*64), Og = log(max(O, 1e-20)) gate, 0 on masked query rows.  Backward: dOin = dOg / O (0 where
 * O <= 1e-20) is what e2k_attn_bwd takes as its dOg; dgate_pre (B,H,N) REPLACES the one e2k_attn_bwd computes. */
int e2k_laser_out_fwd(const void* O, const float* gate, const uint8_t* kmask, void* Og, int B, int H, int N, int Npad, void* stream);
int e2k_laser_out_bwd(const void* dOg, const void* O, const float* gate, const uint8_t* kmask, void* dOin, float* dgate_pre,
                      int B, int H, int N, int Npad, void* stream);
/* O = softmax(mask(50*tanh(Q.K^T/8/50))) . V  (dropout on the probabilities if p_drop > 0), per (b,h).
 * kmask (B, Npad) u8: 1 = attend; also used as the query-row mask of the output (reference: where(mask, out, 0)).
 * O / Og: token-major (B*N, H*64) un-gated / gated by `gate`;  lse2 (B,H,N): log2-domain log-sum-exp. */
int e2k_attn_fwd(const void* Q, const void* K, const void* VT, const uint8_t* kmask, const float* gate,
                 void* O, void* Og, float* lse2, void* dropbits, int B, int H, int N, int Npad, float p_drop,
                 uint32_t seed, const uint32_t* seed_dev, uint32_t stream_id, int flags, void* stream);
/* flags: 0, or bottleneck probes of the forward kernel (tools/probes/attn_ablate.py; the RESULTS ARE WRONG on purpose) */
#define E2K_ATTN_PROBE_NO_KMASK 1    /* no key-mask load per tile */
#define E2K_ATTN_PROBE_NO_QK 2       /* no K LDS reads + score MFMAs */
#define E2K_ATTN_PROBE_NO_SOFTMAX 4  /* no soft-clamp / exp2 / dropout */
#define E2K_ATTN_PROBE_NO_PV 8       /* no V LDS reads + second MFMAs */
#define E2K_ATTN_PROBE_NO_LOADS 16   /* no global K / V tile loads after the first */
#define E2K_ATTN_PROBE_NO_BARRIER 32 /* no workgroup barriers */
#define E2K_ATTN_NO_RING 128         /* (both calls) the register-staged kernels instead of the LDS-DMA ring kernels (A/B and the path of rows longer than 4096 positions; same results) */
#define E2K_ATTN_PLAIN_WG 256        /* (both calls) ring kernels: plain workgroup numbering instead of the XCD-aware one (same results) */
/* dropbits (optional, both calls; NULL = every kernel re-derives the dropout mask from the counter hash): scratch of
 * e2k_query_attn_dropbits_bytes(B, H, N) bytes in which the forward leaves its keep decisions as 64-bit wave ballot words
 * and from which the backward kernels read them back.  Same mask either way (bit-identical results). */
int e2k_query_attn_dropbits_bytes(int B, int H, int N);
/* Which transposed copies the backward needs: bit 0 = KT (the register-staged dQ kernel: flag E2K_ATTN_NO_RING, or
 * Npad > 4096), bit 1 = QT and dOT (the register-staged dK,dV kernel: that flag).  The default
 * (LDS-DMA ring) kernels read K^T / Q^T / dO^T out of the row-major tiles with ds_read_b64_tr_b16: e2k_qkv_post_fwd
 * then takes QT = KT = NULL and e2k_attn_bwd dOT = NULL (17 MB less written per transposed copy at the cfg3 shape). */
int e2k_query_attn_bwd_transposes(int Npad, int flags);

/* backward: dOg (B*N, H*64) -> dQ, dK, dV (B,H,N,64), dgate_pre (B,H,N).  dO, dOT, delta are scratch outputs. */
int e2k_attn_bwd(const void* dOg, const void* O, const float* gate, const float* lse2, const void* Q,
                 const void* K, const void* V, const void* QT, const void* KT, const uint8_t* kmask,
                 const void* dropbits, void* dO, void* dOT, float* delta, float* dgate_pre, void* dQ, void* dK, void* dV,
                 int B, int H, int N, int Npad, float p_drop, uint32_t seed, const uint32_t* seed_dev,
                 uint32_t stream_id, int flags, void* stream);

/* ---- attention across the frequency tokens of a frame (Transformer(has_freq_axis = True), e2_tts.py:653-656,920-932) ----
 * The reference rearranges '(b f) n d -> (b n) f d' and runs a default-keyword x_transformers.Attention (dim_head = 64: no
 * mask, dropout, soft-clamp or head gates; rotary over the token index; value residual mixed at 0.5) over the F <= 8 tokens.
 * qkv (B*F*N, ld) bf16 = [q (H*64) | k | v] in the backbone's own token order (row (b F + j) N + n): no rearrangement is
 * materialised.  vfirst (row stride ldv): the first layer's v columns, NULL on the first layer.  cosb / sinb: rotary table
 * (F, 32).  out (B*F*N, H*64).  Backward: dqkv (B*F*N, lddq) from dout; dvfirst (B*F*N, H*64) fp32 is ACCUMULATED on the
 * later layers and consumed when first_layer = 1 (as in e2k_qkv_post_bwd). */
int e2k_freq_attn_fwd(const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb, const float* sinb,
                      void* out, int B, int F, int N, int H, void* stream);
int e2k_freq_attn_bwd(const void* dout, const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb,
                      const float* sinb, float* dvfirst, int first_layer, void* dqkv, int64_t lddq, int B, int F, int N,
                      int H, void* stream);

/* ---- MelSpec (e2_tts.py:248-290 -> torchaudio MelSpectrogram(n_fft=1024, hop, power=1, center, htk, norm=None)) ----
 * wave (B, nw) fp32 -> out (B, n_mels, 1 + nw/hop) fp32 = log(clamp(mel, 1e-5)).  window (n_fft) periodic Hann,
 * fb (n_fft/2+1, n_mels) filterbank, twc/tws (n_fft/2): cos/sin(2*pi*k/n_fft).  n_fft must be 1024.
 * bands (optional, int32 [n_mels][2]): bins [lo, hi) outside of which column m of fb is zero -- the htk triangles of
 * torchaudio's melscale_fbanks overlap only their neighbours, so the 513 x 100 contraction has ~1 k non-zero terms of
 * 51 k; NULL = dense contraction. */
int e2k_melspec(const float* wave, int64_t nw, const float* window, const float* fb, const float* twc,
                const float* tws, float* out, int B, int n_fft, int hop, int n_mels, const int32_t* bands, void* stream);

/* Ragged batch (the dataset side): wave (B, nw) holds clips of different length, zero-padded; lens[b]
 * = valid samples of row b (n_fft/2 < lens[b] <= nw).  Row b is transformed exactly as if it were alone (reflection
 * about ITS end, 1 + lens[b]/hop frames); the remaining frames of the (B, n_mels, 1 + nw/hop) output get pad_value -- what
 * the reference's per-clip MelSpec in HFDataset.__getitem__ followed by collate_fn's zero padding produces
 * (trainer.py:61-82,101-131), in one launch on the device. */
int e2k_melspec_ragged(const float* wave, int64_t nw, const int32_t* lens, const float* window, const float* fb,
                       const float* twc, const float* tws, float* out, float pad_value, int B, int n_fft, int hop,
                       int n_mels, const int32_t* bands, void* stream);

/* Sample-rate conversion of the dataset path (trainer.py:116-118: torchaudio.transforms.Resample(sample_rate, target) per clip, here for
 * the whole ragged batch on the device in front of e2k_melspec_ragged).  orig / nw: source / target rate divided by their gcd; kernel
 * (nw, taps) fp32, taps = 2 width + orig: the windowed-sinc polyphase filter (built on the host, data.Resample); x (B, ldx) holds n_in
 * samples per row of which lens[b] are valid (NULL: all); out (B, ldo): out[b][j nw + ph] = sum_k kernel[ph][k] xpad_b[j orig + k] for the
 * first ceil(lens[b] nw / orig) outputs, zeros up to n_out. */
int e2k_resample_sinc(const float* x, int64_t ldx, int64_t n_in, const int32_t* lens, const float* kernel, float* out, int64_t ldo,
                      int64_t n_out, int B, int orig, int nw, int taps, int width, void* stream);

/* ---- optimizer side over flat fp32 buffers (reference trainer.py:272-279) ----
 * out[0] += sum x^2 (fp64 accumulation): the global gradient norm of accelerator.clip_grad_norm_ (trainer.py:272-273). */
int e2k_sumsq_f32(const float* x, int64_t n, double* out, void* stream);
/* One ADOPT step (adam_atan2_pytorch.adopt.Adopt, trainer.py:183,275) on n elements:
 *   g' = g * min(1, max_grad_norm / (sqrt(*gsumsq) + 1e-6))      (gsumsq NULL or max_grad_norm <= 0: no clipping)
 *   step 0: v = g'^2.   step >= 1: u = clamp(g' / max(sqrt(v), eps), +-step^0.25); m += (1-beta1)(u - m);
 *   p = p (1 - lr weight_decay) - lr m;  v += (1-beta2)(g'^2 - v).
 * shadow_bf16 (optional): the bf16 compute copy of p, refreshed in the same pass. */
int e2k_adopt_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                   const double* gsumsq, int step, void* stream);
/* The same step with TWO parameter groups in one flat buffer: Adopt keeps `steps` per parameter and skips parameters whose
 * .grad is None (trainer.py:183,275) -- the text stream's parameters on steps whose classifier-free-guidance coin drops
 * the text (e2_tts.py:1261).  Elements inside one of the `nranges` (<= 128) sorted [start, end) element ranges
 * (int32 pairs on the device, bounds multiples of 4) belong to group b: they use step_b, or are left untouched
 * (parameter, moments, shadow) when active_b == 0; all other elements use `step`. */
int e2k_adopt_step_groups(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                          const double* gsumsq, int step, int step_b, int active_b, const int32_t* ranges, int nranges,
                          void* stream);
/* e2k_adopt_step_groups followed by e2k_ema_update(ema, p, n, ema_decay) in ONE pass: trainer.py:275 and :279 on the steps on which
 * ema_pytorch.EMA.update() moves the average (every `update_every`-th step).  The average takes the NEW parameter value while it is
 * in registers (8 B per element instead of a 12-B pass of its own); elements of a skipped group keep their parameter and still
 * move their average.  nranges may be 0 (ranges NULL). */
int e2k_adopt_step_ema(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                       const double* gsumsq, int step, int step_b, int active_b, const int32_t* ranges, int nranges,
                       float* ema, float ema_decay, void* stream);
/* ema += (1 - decay) (p - ema)   (ema_pytorch.EMA.update, trainer.py:170,279) */
int e2k_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream);
/* Data-parallel gradient exchange in bf16 (replaces the implicit DDP reducer of trainer.py:155-162,190-192,270 for the
 * backbone's flat gradient slabs): wire[i] = bf16(g[i] * scale) into a preallocated buffer, and g[i] = float(wire[i]) after
 * the all-reduce.  One pass each, 16-byte aligned pointers. */
int e2k_grad_pack_bf16(const float* g, void* wire_bf16, int64_t n, float scale, void* stream);
int e2k_grad_unpack_bf16(const void* wire_bf16, float* g, int64_t n, void* stream);
/* The same exchange with fp32 accumulation (one rounding instead of world - 1; the reference's reducer sums fp32 gradients,
 * trainer.py:155-162): after an all-to-all of the slab's shards a rank holds (world, per) bf16 = every peer's copy of its shard;
 * out[i] = bf16(sum over r of float(recv[r][i])), summed in rank order.  per a multiple of 8, 16-byte aligned pointers. */
int e2k_shard_sum_bf16(const void* recv_bf16, void* out_bf16, int64_t per, int world, void* stream);


/* ---- launch plans: the native scheduler of the backbone (csrc/plan.h) ----
 * The reference's hot loop is Python calling one ATen op at a time (e2_tts.py:825-939); replayed naively through this ABI
 * a dim-1024 / depth-24 step is ~3000 calls whose Python / ctypes cost equals the kernels' own time.  A plan records the
 * calls once and re-issues them from C++.
 *   e2k_plan_begin()            start recording on the calling thread: every compute entry point called from now on is
 *                               executed AND appended (its arguments by value: the buffers must stay where they are)
 *   e2k_query_plan_recorded()   number of calls recorded so far (-1: not recording) -- segment boundaries
 *   e2k_query_plan_end()        stop recording, returns the plan handle (> 0) or a negative error
 *   e2k_plan_abort()            drop a recording in progress
 *   e2k_plan_run(plan, first, count, stream)   re-issue calls [first, first + count) (count < 0: to the end) on `stream`
 *   e2k_plan_profile(...)       the same with a HIP event after every call; ms_host[i] (HOST pointer, count floats)
 *                               receives the time between the events around call first + i; synchronises the stream
 *   e2k_plan_op_name(...)       name of recorded call `index` into the HOST buffer buf_host (e.g. "gemm_nt_bf16")
 *   e2k_plan_free(plan)
 * The registry of plans is the one piece of process-global state in the library (mutex protected); the recording state
 * is per thread.  Device-side values that change between replays are read through pointers (seed_dev, adopt's gsumsq). */
int e2k_plan_begin(void);
int e2k_query_plan_recorded(void);
int e2k_query_plan_end(void);
int e2k_plan_abort(void);
int e2k_plan_free(int plan);
int e2k_query_plan_size(int plan);
int e2k_plan_run(int plan, int first, int count, void* stream);
/* Launch lanes (csrc/plan.h): a recorded call belongs to the lane that was current when it was recorded (0 = the caller's
 * stream; the text stream's branches and the weight-gradient GEMMs of the backbone go to lanes 1 and 2), and the
 * recording holds explicit ordering points between lanes.  While nothing is being recorded the three calls below are
 * no-ops (the caller orders its own streams); e2k_plan_run_lanes replays with streams_host[lane] (HOST array of
 * nstreams <= 4 hipStream_t; calls of lanes >= nstreams run on streams_host[0]; nstreams = 1 is e2k_plan_run). */
int e2k_plan_lane(int lane);
int e2k_plan_event_record(int lane, int ev);
int e2k_plan_event_wait(int lane, int ev);
int e2k_plan_run_lanes(int plan, int first, int count, void** streams_host, int nstreams);
int e2k_plan_profile(int plan, int first, int count, float* ms_host, void* stream);
/* HIP-graph form of a replay (round 6): e2k_query_plan_graph_capture runs the loop of e2k_plan_run_lanes over calls [first, first + count)
 * under stream capture -- streams_host[0] captures, the side lanes are forked from it and joined back, the recorded ordering points
 * become graph edges; NOTHING executes -- and returns a graph handle (> 0) or a negative error; e2k_plan_graph_launch re-issues the whole
 * range with one hipGraphLaunch on `stream`.  For ranges no other stream interleaves with (a forward pass; a backward pass without a
 * gradient exchange); buffers, like the plan's, must stay where they were. */
int e2k_query_plan_graph_capture(int plan, int first, int count, void** streams_host, int nstreams);
int e2k_plan_graph_launch(int graph, void* stream);
int e2k_plan_graph_free(int graph);
int e2k_plan_op_name(int plan, int index, char* buf_host, int nbuf);
int e2k_query_plan_op_lane(int plan, int index);      /* launch lane of recorded call `index` (-1: no such call) */

/* ---- stream pack / unpack, masks, time conditioning: the glue around the depth loop (csrc/glue.hip) ----
 * byte fill (hipMemsetAsync as a recordable call) and a strided form: `rows` rows of `width` bytes, `pitch` bytes apart */
int e2k_fill_bytes(void* dst, int value, int64_t nbytes, void* stream);
int e2k_fill_bytes_2d(void* dst, int64_t pitch, int value, int64_t width, int64_t rows, void* stream);
int e2k_cast_f32(const void* src_bf16, float* dst, int64_t n, void* stream);
int e2k_sigmoid_f32(const float* src, float* dst, int64_t n, void* stream);
/* key / row masks of a batch (e2_tts.py:771: registers are always attended): mask (B,T) u8 / bool or NULL (= all ones)
 * -> kmask (B,Npad): [1 x R | mask | 0 x (Npad - N)],  mask_n (B,N) (optional): the first N columns, contiguous */
int e2k_build_masks(const uint8_t* mask, uint8_t* kmask, uint8_t* mask_n, int B, int T, int R, int Npad, void* stream);
/* residual-stream pack (e2_tts.py:760-768,818: abs-pos added, registers prepended, expanded to 4 hyper-connection
 * streams):  X[b][n][s][:] = bf16( n < R ? regs[n] : x[b][n-R] + abs_pos[n-R] ),  x fp32 (B,T,D), abs_pos (>=T, D) or NULL */
int e2k_stream_pack_fwd(const float* x, const float* abs_pos, const float* regs, void* X, int B, int T, int R, int D,
                        void* stream);
/* backward: dx[b][t] = sum_s dX[b][R+t][s];  dregs[n] += sum_b sum_s dX[b][n][s];  dabs[t] += sum_b dx[b][t] (optional) */
int e2k_stream_pack_bwd(const void* dX, float* dx, float* dregs, float* dabs, int B, int T, int R, int D, void* stream);
/* reduce (e2_tts.py:947: sum of the streams) without the registers (e2_tts.py:949): xsum[b*T+t] = bf16(sum_s X[b][R+t][s]) */
int e2k_stream_unpack_fwd(const void* X, void* xsum, int B, int T, int R, int D, void* stream);
/* backward: dX[b][n][s] = n < R ? 0 : dxs[b*T + n - R]   (all four streams) */
int e2k_stream_unpack_bwd(const void* dxs, void* dX, int B, int T, int R, int D, void* stream);
/* time conditioning (e2_tts.py:355-364,621-625,782): out[b][j] = silu(pre), pre = W[j][:] . [t_b, sin(2 pi t_b w), cos(2 pi t_b w)] + bias[j]
 * times (B) fp32, fw (D/2) the fixed random Fourier weights, W (D, D+1) fp32, bias (D); `four` (B, D+1) and `pre` (B, D)
 * are kept for the backward.  bwd: dW += dpre^T four, dbias += sum_b dpre, dpre = dout * silu'(pre). */
int e2k_time_cond_fwd(const float* times, const float* fw, const float* W, const float* bias, float* four, float* pre,
                      float* out, int B, int D, void* stream);
int e2k_time_cond_bwd(const float* dout, const float* four, const float* pre, float* dW, float* dbias, int B, int D,
                      void* stream);
/* backward of the hoisted time-conditioning block (rows [layer][4][D] of dcond, gates = sigmoid(condall)):
 * gate slots (1, 3) of dcond are multiplied by (1 - gate) in place, dcb = bf16(dcond), dct (4LD, KB) = bf16(dcond)^T zero
 * padded to KB columns, gbias[l][1|3][:] += sum_b dcond (the AdaLN-Zero biases; slots 0, 2 have no bias) */
int e2k_cond_bwd_prep(float* dcond, const float* gates, void* dcb, void* dct, float* gbias, int B, int L, int D, int KB,
                      void* stream);
/* The prologue of E2TTS.forward (e2_tts.py:1519-1543) in one pass: x0 (noise), x1 (mel) (B T, C) fp32 contiguous, t (B) the flow times,
 * span_mask (B T) bytes = the random span to be infilled -> flow = x1 - x0, cond = span ? 0 : x1 (both (B T, C) fp32, returned to the caller
 * by the reference), and the input projection's GEMM operands w = (1 - t) x0 + t x1 and cond as bf16 (B T, ldp) zero-padded to Cpad columns. */
int e2k_flow_pack(const float* x0, const float* x1, const float* t, const uint8_t* span_mask, void* w_bf16, void* cond_bf16, int64_t ldp,
                  float* flow, float* cond, int B, int T, int C, int Cpad, void* stream);
/* fp32 (R, C), rows lds floats apart -> bf16 (R, ldd) with columns C .. Cpad-1 zeroed: operands of the 100-channel input /
 * output projections (e2_tts.py:1267-1277,1296: proj_in, cond_proj_in, to_pred), whose K is padded to a multiple of 8 */
int e2k_cast_pad_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int R, int C, int Cpad, void* stream);
/* CharacterEmbed (e2_tts.py:390-412): tok (B, nt) int64 byte tokens with -1 padding -> out (B, T, D) fp32 =
 * W[n < nt ? tok + 1 : 0], W (V, D) = nn.Embedding(num_embeds + 1, dim).weight.  Backward: dW (V, D) += scatter of dout (ACCUMULATES). */
int e2k_char_embed_fwd(const int64_t* tok, const float* W, float* out, int B, int nt, int T, int D, int V, void* stream);
int e2k_char_embed_bwd(const int64_t* tok, const float* dout, float* dW, int B, int nt, int T, int D, int V, void* stream);
/* Duration head (e2_tts.py:1098-1111: maybe_masked_mean :212-224 + HLGaussLayer regression mode with Softplus):
 * embed (B, T, D) fp32, mask (B, T) u8 or NULL, w (D) = hl_gauss_layer.to_pred.weight -> pooled (B, D), z (B) = w . pooled,
 * pred (B) = softplus(z).  Backward from dpred (B): dembed (B, T, D) written, dw (D) ACCUMULATED. */
int e2k_duration_head_fwd(const float* embed, const uint8_t* mask, const float* w, float* pooled, float* z, float* pred,
                          int B, int T, int D, void* stream);
int e2k_duration_head_bwd(const float* dpred, const float* z, const float* pooled, const uint8_t* mask, const float* w,
                          float* dembed, float* dw, int B, int T, int D, void* stream);
/* Classifier-free-guidance combine of E2TTS.sample (e2_tts.py:1303-1330 with `project`, :113-124), per sample b over its L
 * = frames * channels elements, fp32 in / out, the projection in fp64 as the reference does it:
 *   u = pred - null;  par = (u . unit) unit with unit = pred / max(|pred|, 1e-12);  orth = u - par
 *   out = pred + (orth + par * keep_parallel_frac) * cfg_strength      (remove_parallel = 0: out = pred + u * cfg_strength) */
int e2k_cfg_combine(const float* pred, const float* null_pred, float* out, int B, int64_t L, float cfg_strength,
                    float keep_parallel_frac, int remove_parallel, void* stream);
/* flow-matching loss (e2_tts.py:1578-1582: F.mse_loss(pred, flow, reduction = 'none')[rand_span_mask].mean()) without the
 * boolean-index gather: acc[0] = sum over masked rows of sum_c (pred - flow)^2, acc[1] = number of masked rows (acc is
 * zeroed by the call); loss = acc[0] / (acc[1] * C).  bwd: dpred = dloss[0] * 2 (pred - flow) mask / (acc[1] * C) */
int e2k_masked_mse_fwd(const float* pred, const float* flow, const uint8_t* mask, float* acc, int M, int C, void* stream);
int e2k_masked_mse_bwd(const float* pred, const float* flow, const uint8_t* mask, const float* acc, const float* dloss,
                       float* dpred, int M, int C, void* stream);
/* out (C, R) = in (R, ld >= C)^T, fp32, first C columns */
int e2k_transpose_f32(const float* in, int64_t ld, float* out, int R, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
