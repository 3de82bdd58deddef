/* e2k -- C ABI of the MI355X (gfx950) kernels behind the E2-TTS flow-matching transformer hot path.
 *
 * The reference (lucidrains/e2-tts-pytorch) has no FFI: its hot path is Python calling ATen ops.  This
 * header is the boundary a maintainer would bind instead (ctypes stub in INTEGRATION.md); every entry
 * point cites the reference call site(s) it replaces (file:line into /root/reference/e2_tts_pytorch).
 *
 * Conventions
 *   - every function returns 0 on success, E2K_ERR_* (or 1000 + hipError_t) otherwise; no exceptions,
 *     no global mutable state, re-entrant, nothing is allocated or freed inside;
 *   - the caller owns every buffer; all pointers are device pointers (HBM) unless stated otherwise;
 *   - `stream` is a hipStream_t (NULL = default stream); kernels are only enqueued, never synchronised;
 *   - "bf16" buffers hold raw bfloat16 bits (uint16_t); leading dimensions (ld*) are in ELEMENTS;
 *   - 16-byte vector access: bf16 base pointers must be 16-B aligned and ld* multiples of 8.
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

/* C[M,N] = ((([A1|A2] . B^T) + bias[n]) * colscale[m / rows_per_batch][n]) * rowmask[m] + resid[m][n]
 * A1 (M,K1), A2 (M,K2) optional second K-panel (K2 = 0: none), B (N,K1+K2), all bf16 row-major.
 * C bf16 (out_f32 = 0) or fp32 (out_f32 = 1; accumulate = 1 adds into C).  bias/colscale fp32, rowmask u8,
 * resid bf16 (any of them NULL = skipped).  K1, K2 multiples of 8.
 * Replaces: nn.Linear in x_transformers.Attention / FeedForward (e2_tts.py:875,881,911,937), the output
 * mask + AdaLNZero gate (e2_tts.py:346-351,913,938), skip_proj on cat(x, skip) (e2_tts.py:895-896),
 * TextAudioCrossCondition on cat(audio, text) + residual add (e2_tts.py:508-513), proj_in/cond_proj_in/to_pred
 * (e2_tts.py:1267-1277,1296) and all of their dgrad GEMMs. */
int e2k_gemm_nt_bf16(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                     const void* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int accumulate,
                     int M, int N, const float* bias, const float* colscale, int rows_per_batch,
                     const uint8_t* rowmask, const void* resid, int64_t ldr, void* stream);

/* C[N,K] += A[M,N]^T . B[M,K]  (weight gradients; C fp32, A = dY, B = X, bf16).  The token dimension M is
 * split over `splits` workgroups per tile (0 = choose), partial tiles are combined with fp32 atomics.
 * use_tr = 1 reads MFMA fragments with ds_read_b64_tr_b16, 0 = plain 16-bit LDS gathers (same results). */
int e2k_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                     int M, int N, int K, int splits, int use_tr, void* stream);

#ifdef __cplusplus
}
#endif
#endif
