// bf16 MFMA GEMMs for the transformer hot path (gfx950, v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
//   e2k_gemm_nt_bf16 : C[M,N]  = epilogue( [A1|A2][M,K1+K2] . B[N,K1+K2]^T )      forward + dgrad
//   e2k_gemm_tn_bf16 : C[N,K] += A[M,N]^T . B[M,K]   (fp32 C, split over M)         wgrad
//
// Replaces the nn.Linear / F.linear calls of the reference hot loop: q/k/v/out projections
// (x_transformers.Attention, called at e2_tts.py:875,911), GEGLU feed-forward (e2_tts.py:881,937),
// skip projection on cat(x, skip) (e2_tts.py:895-896) and TextAudioCrossCondition on cat(audio, text)
// (e2_tts.py:508-513).  The two concatenations are never materialised: the NT kernel walks K over two
// source matrices ("dual-source A").
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.  Operands are staged
// global -> VGPR -> LDS (16 B per lane, XOR-swizzled 128-B rows => conflict-free ds_read_b128), the loads
// of tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after them
// (one barrier per K step).  MFMA operands are swapped (srcA = weight fragment, srcB = activation fragment) so
// that a lane's 4 accumulator registers are 4 consecutive output columns -> 8-byte bf16 stores.
#include <type_traits>
#include <cstdlib>
#include "e2k_device.h"
#include "plan.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"

using namespace e2k;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int DH_ROT = 64;       // dim_head of the rotary epilogue (the attention kernels' 64)
constexpr int NT_SLOTS = 512;        // resident NT workgroups on the chip: 256 CUs x 2 (64 KB LDS each)

__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous run of tiles (bijective)
    int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

__device__ __forceinline__ void tile_coords(int wgid, int tm, int tn, int& tile_m, int& tile_n, int group = 8) {
    int width = group * tn;
    int gid = wgid / width;
    int first_m = gid * group;
    int gsize = min(tm - first_m, group);
    int in_g = wgid - gid * width;
    tile_m = first_m + in_g % gsize;
    tile_n = in_g / gsize;
}

struct NTArgs {
    const bf16_t* A1; long lda1; int K1;
    const bf16_t* A2; long lda2; int K2;
    const bf16_t* B; long ldb;
    void* C; long ldc; int accumulate;
    int M, N;
    const float* bias; const float* colscale; long lds; int rows_per_batch;
    const uint8_t* rowmask; const bf16_t* resid; long ldr;
    // two-output form (e2k_gemm_nt2_bf16): columns [nsplit, N) of the product go to a second matrix.  C2 / resid2 are stored
    // SHIFTED by -nsplit columns, so that the epilogues index them with the same global column as C / resid; nsplit is a
    // multiple of 256, i.e. a tile lies in one output and the choice is wave-uniform (nt_bind_output)
    int nsplit; void* C2; long ldc2; const bf16_t* resid2; long ldr2;
    int group;          // row tiles per group of the tile order (tile_coords): chosen so that a group is about one XCD's share

    int probe;          // E2K_GEMM_PROBE_* bits (bottleneck probes: results are wrong on purpose)
    // remainder split: workgroups [0, full) own whole tiles; the last T - full tiles (a partial round of the 512
    // resident workgroups) are cut into `split` K ranges each, fp32 partials go to `ws`, gemm_nt_fixup_kernel finishes
    int full, split; float* ws;
    int staged;         // 256 x 256 kernel: C tile written through LDS in whole 512-byte row segments (nt_epilogue_staged)
    // GEGLU epilogue (gemm_nt_256_kernel<false, true> only): N = 2F, C = pre-activation H (may be NULL), glu_out (M, F)
    bf16_t* glu_out; long ldg; unsigned seed, stream_id, thresh; float inv_keep; const unsigned* seed_dev;
    // GEGLU BACKWARD as the epilogue (e2k_gemm_nt_geglu_bwd_bf16): the product is d(activation) (M, N = F); with the stored
    // pre-activation gb_H (M, 2F) = [u | g] the epilogue writes gb_dH (M, 2F) = [d keep gelu(g) | d keep u gelu'(g)] and C is not written
    const bf16_t* gb_H; long gb_ldh; bf16_t* gb_dH; long gb_lddh;
    // rotary q / k as the epilogue of the attention's input projection (e2k_gemm_nt_qkrot_bf16, gemm_nt_256_kernel<false, false, 2>):
    // tiles left of column rot_2I = 2 H 64 hold q | k; their values leave rotated and HEAD-MAJOR, (B, H, rot_N, 64) at rot_Q / rot_K
    // (what qkv_post_fwd_kernel would have made of them), the other tiles (v, head gates, value-residual mix) go to C as always
    bf16_t *rot_Q, *rot_K; const float *rot_cos, *rot_sin; int rot_I, rot_2I, rot_N, rot_H;
};

// two-output form: rebinds the (by-value) argument block of a workgroup whose tile lies in the second output
__device__ __forceinline__ void nt_bind_output(NTArgs& p, int n0) {
    if (p.nsplit > 0 && n0 >= p.nsplit) {
        p.C = p.C2; p.ldc = p.ldc2; p.resid = p.resid2; p.ldr = p.ldr2;
    }
}

// GEGLU backward on NC consecutive columns n .. n + NC - 1 of row m (x: the product = d(activation), fp32): the arithmetic of
// geglu_kernel<true> (elementwise.hip; FeedForward, e2_tts.py:646,692) with the A&S erfc of the forward epilogue (gelu_erf_pair below:
// one v_rcp_f32 + one v_exp_f32 for gelu AND its derivative; |error| <= 1.5e-7), fed the fp32 accumulator instead of a bf16-rounded copy
__device__ __forceinline__ void gelu_erf_pair(float x, float& gel, float& dgel) {
    const float ax = fabsf(x);
    const float t = fast_rcp(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
    float y = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    y = fmaf(t, y, 0.5f * 1.421413741f);
    y = fmaf(t, y, 0.5f * -0.284496736f);
    y = fmaf(t, y, 0.5f * 0.254829592f);
    const float e = fast_exp2(x * x * (-0.5f * 1.4426950408889634f));         // exp(-x^2 / 2)
    y = y * t * e;                                                              // erfc(|x| / sqrt 2) / 2
    const float phi = x < 0.f ? y : 1.f - y;                                    // Phi(x)
    gel = x * phi;
    dgel = fmaf(x * e, 0.3989422804014327f, phi);                               // Phi(x) + x phi(x)
}
template <int NC>
__device__ __forceinline__ void nt_store_geglu_bwd(const NTArgs& p, int m, int n, const float* x) {
    static_assert(NC == 4 || NC == 8, "");
    const int F = p.N;
    float u[NC], gt[NC], du[NC], dg[NC];
    const bf16_t* hrow = p.gb_H + (long)m * p.gb_ldh + n;
    if (NC == 8) { unpack8(ld<u32x4>(hrow), u); unpack8(ld<u32x4>(hrow + F), gt); }
    else { unpack4(ld<u32x2>(hrow), u); unpack4(ld<u32x2>(hrow + F), gt); }
    const unsigned seed = p.seed_dev ? *p.seed_dev : p.seed;
#pragma unroll
    for (int e = 0; e < NC; ++e) {
        const float ks = p.thresh ? keep_scale(seed, p.stream_id, m, n + e, p.thresh, p.inv_keep) : 1.f;
        float gel, dgel;
        gelu_erf_pair(gt[e], gel, dgel);
        const float dd = x[e] * ks;
        du[e] = dd * gel;
        dg[e] = dd * u[e] * dgel;
    }
    bf16_t* drow = p.gb_dH + (long)m * p.gb_lddh + n;
    if (NC == 8) { st<u32x4>(drow, pack8(du)); st<u32x4>(drow + F, pack8(dg)); }
    else { st<u32x2>(drow, pack4(du)); st<u32x2>(drow + F, pack4(dg)); }
}

// epilogue of an interior tile (all 128 x 128 outputs exist, rows 8-byte aligned): no per-element bounds checks, the
// optional operands are selected once per tile (wave-uniform), bias kept in registers across the 4 row groups
template <bool OUT_F32, bool HAS_CS, bool HAS_RES, int NI, int NJ>
__device__ __forceinline__ void nt_epilogue_full(const NTArgs& p, f32x4 (&acc)[NI][NJ], int mw, int nw, int l15, int g) {
    const int nb = nw + 4 * g;
    f32x4 bias4[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[j][r] = p.bias[nb + j * 16 + r];
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = mw + i * 16 + l15;
        const float rm = p.rowmask ? (p.rowmask[m] ? 1.f : 0.f) : 1.f;
        const float* cs = HAS_CS ? p.colscale + (long)(m / p.rows_per_batch) * p.lds : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nb + j * 16;
            f32x4 x = acc[i][j] + bias4[j];
            if (HAS_CS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] *= cs[n + r];
            }
            x *= rm;
            if (HAS_RES) {
                float rs[4];
                unpack4(ld<u32x2>(p.resid + (long)m * p.ldr + n), rs);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] += rs[r];
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                if (p.accumulate) x += ld<f32x4>(c);
                st<f32x4>(c, x);
            } else {
                float v[4] = {x[0], x[1], x[2], x[3]};
                st<u32x2>((bf16_t*)p.C + (long)m * p.ldc + n, pack4(v));
            }
        }
    }
}

// epilogue shared by the NT kernels.  (mw, nw) = first row / column of this wave's (16 NI) x (16 NJ) sub-tile; lane holds
// C[m][n..n+3], m = mw + i*16 + l15 (i < NI), n = nw + j*16 + 4g (j < NJ)
template <bool OUT_F32, int NI, int NJ = 4, bool GB = false>      // GB: a separate instantiation, so that the plain kernels carry none of it
__device__ __forceinline__ void nt_epilogue(const NTArgs& p, f32x4 (&acc)[NI][NJ], int mw, int nw, int l15, int g) {
    if (GB) {            // GEGLU backward epilogue (N = F is a multiple of 256: every column of a tile exists)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mw + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float x[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                nt_store_geglu_bwd<4>(p, m, nw + j * 16 + 4 * g, x);
            }
        }
        return;
    }
    const bool vec_ok = (p.ldc & 3) == 0 && (p.resid == nullptr || (p.ldr & 3) == 0);
    if (vec_ok && mw + NI * 16 <= p.M && nw + NJ * 16 <= p.N) {       // wave-uniform: the whole sub-tile exists
        if (p.colscale) {
            if (p.resid) nt_epilogue_full<OUT_F32, true, true, NI, NJ>(p, acc, mw, nw, l15, g);
            else nt_epilogue_full<OUT_F32, true, false, NI, NJ>(p, acc, mw, nw, l15, g);
        } else {
            if (p.resid) nt_epilogue_full<OUT_F32, false, true, NI, NJ>(p, acc, mw, nw, l15, g);
            else nt_epilogue_full<OUT_F32, false, false, NI, NJ>(p, acc, mw, nw, l15, g);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = mw + i * 16 + l15;
        if (m >= p.M) continue;
        const float rm = p.rowmask ? (p.rowmask[m] ? 1.f : 0.f) : 1.f;
        const float* cs = p.colscale ? p.colscale + (long)(m / p.rows_per_batch) * p.lds : nullptr;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nw + j * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            const bool full = (n + 3 < p.N) && vec_ok;
            float rs[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.resid) {
                if (full) {
                    unpack4(ld<u32x2>(p.resid + (long)m * p.ldr + n), rs);
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) rs[r] = bf2f(p.resid[(long)m * p.ldr + n + r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r < p.N) {
                    float x = v[r];
                    if (p.bias) x += p.bias[n + r];
                    if (cs) x *= cs[n + r];
                    x = x * rm + rs[r];
                    v[r] = x;
                }
            }
            if (OUT_F32) {
                float* c = (float*)p.C + (long)m * p.ldc + n;
                if (full) {
                    f32x4 o = {v[0], v[1], v[2], v[3]};
                    if (p.accumulate) { f32x4 old = ld<f32x4>(c); o += old; }
                    st<f32x4>(c, o);
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) c[r] = (p.accumulate ? c[r] : 0.f) + v[r];
                }
            } else {
                bf16_t* c = (bf16_t*)p.C + (long)m * p.ldc + n;
                if (full) {
                    st<u32x2>(c, pack4(v));
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) c[r] = f2bf(v[r]);
                }
            }
        }
    }
}

// Epilogue of the 256 x 256 kernel through LDS.  The direct epilogue above stores what a lane holds -- 4 consecutive
// columns of one row, i.e. 16 rows x 32 bytes per wave instruction: sixteen partial cache lines.  Measured on MI355X
// (profiles/r03_nt_store_burst.json): a round of 256 tiles at K = 1024 takes 35.4 us with those stores and 25.6 us
// without, whether or not the workgroups' epilogues coincide -- 10 us for 128 KB per CU is the per-CU cost of partial-line
// stores, not an HBM burst.  Here half a tile (128 rows x 256 columns, fp32, rows padded by 16 B so that the 16 rows of a
// ds_write_b128 group tile all 64 banks) goes to LDS -- the K-tile ring is free by then -- and is read back row-wise:
// 32 consecutive lanes own 128 consecutive columns, so every global access (C, residual, bias, per-batch gate) covers
// whole 128-byte lines.  Same arithmetic, in the same order, as nt_epilogue: ((acc + bias) * gate) * rowmask + residual.
constexpr int QSTAGE_ROW = 256 * 4 + 16, QSTAGE_BYTES = 128 * QSTAGE_ROW;
constexpr int GSTAGE_ROW = 128 * 4 + 16, GSTAGE_BYTES = 128 * GSTAGE_ROW;       // 128 x 128 kernel: the whole tile at once

// read-back half of the staged epilogue: ROWS x COLS fp32 staged at S (row stride COLS * 4 + 16 bytes), C rows m_base + r,
// columns n0 + ...  A lane owns 8 consecutive columns of a row (COLS / 8 lanes per row): bf16 results leave as ONE 16-byte
// store per lane -- the epilogue is bound by the number of store instructions per CU, not by bytes (8-byte stores in
// whole-line order measured no faster than the direct epilogue, profiles/r03_gemm_epilogue_ab_8byte_stores.json).  The
// residual rows of all passes are fetched up front (their latency would otherwise be paid once per pass).
// EPI: 0 plain, 1 GEGLU backward, 2 rotary q / k (each its own instantiation: the plain kernels carry none of it)
template <bool OUT_F32, int ROWS, int COLS, int THREADS, int EPI = 0>
__device__ __forceinline__ void nt_stage_readback(const NTArgs& p, const unsigned char* S, int m_base, int n0, int tid) {
    constexpr int LPR = COLS / 8, RPP = THREADS / LPR, NP = ROWS / RPP, ROWB = COLS * 4 + 16;
    const int c = tid % LPR, rsub = tid / LPR;
    const int n = n0 + 8 * c;
    if (n >= p.N) return;                                // (N is a multiple of 8 on this path: a chunk is in or out as a whole)
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias) { b0 = ld<f32x4>(p.bias + n); b1 = ld<f32x4>(p.bias + n + 4); }
    if (EPI == 2 && n0 < p.rot_2I) {       // (tile-uniform: rot_2I is a multiple of the tile width)
        // a lane's 8 columns are 4 rotation pairs of one head: column n = which * I + h * 64 + d0.  The product is rounded to bf16
        // first, as if it had gone through C: bit-identical to e2k_gemm_nt_bf16 + the q / k half of e2k_qkv_post_fwd
        const int which = n >= p.rot_I ? 1 : 0, nn = n - which * p.rot_I, h = nn >> 6, d0 = nn & 63;
        bf16_t* const base = (which ? p.rot_K : p.rot_Q) + (long)h * p.rot_N * DH_ROT + d0;
        const int mf = m_base + rsub;
        int b = mf / p.rot_N, tok = mf - b * p.rot_N;
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int r = pass * RPP + rsub;
            if (m_base + r < p.M) {
                const f32x4 x0 = ld<f32x4>(S + r * ROWB + c * 32) + b0, x1 = ld<f32x4>(S + r * ROWB + c * 32 + 16) + b1;
                const f32x4 cs = ld<f32x4>(p.rot_cos + (long)tok * 32 + (d0 >> 1)), sn = ld<f32x4>(p.rot_sin + (long)tok * 32 + (d0 >> 1));
                float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                unpack8(pack8(v), v);
#pragma unroll
                for (int j = 0; j < 4; ++j) rot_pair(v[2 * j], v[2 * j + 1], cs[j], sn[j]);
                st<u32x4>(base + ((long)b * p.rot_H * p.rot_N + tok) * DH_ROT, pack8(v));
            }
            tok += RPP;
            while (tok >= p.rot_N) { tok -= p.rot_N; ++b; }
        }
        return;
    }
    u32x4 rs[NP];
    if (p.resid) {
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int m = min(m_base + pass * RPP + rsub, p.M - 1);
            rs[pass] = ld<u32x4>(p.resid + (long)m * p.ldr + n);
        }
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int r = pass * RPP + rsub, m = m_base + r;
        if (m >= p.M) continue;
        if (EPI == 1) {      // GEGLU backward epilogue: 8 columns of d(activation) -> 8 + 8 columns of dH
            const f32x4 y0 = ld<f32x4>(S + r * ROWB + c * 32), y1 = ld<f32x4>(S + r * ROWB + c * 32 + 16);
            const float x[8] = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
            nt_store_geglu_bwd<8>(p, m, n, x);
            continue;
        }
        const float rm = p.rowmask ? (p.rowmask[m] ? 1.f : 0.f) : 1.f;
        f32x4 x0 = ld<f32x4>(S + r * ROWB + c * 32) + b0, x1 = ld<f32x4>(S + r * ROWB + c * 32 + 16) + b1;
        if (p.colscale) {
            const float* cs = p.colscale + (long)(m / p.rows_per_batch) * p.lds + n;
            x0 *= ld<f32x4>(cs);
            x1 *= ld<f32x4>(cs + 4);
        }
        x0 *= rm;
        x1 *= rm;
        if (p.resid) {
            float f[8];
            unpack8(rs[pass], f);
            x0 += f32x4{f[0], f[1], f[2], f[3]};
            x1 += f32x4{f[4], f[5], f[6], f[7]};
        }
        if (OUT_F32) {
            float* cp = (float*)p.C + (long)m * p.ldc + n;
            if (p.accumulate) { x0 += ld<f32x4>(cp); x1 += ld<f32x4>(cp + 4); }
            st<f32x4>(cp, x0);
            st<f32x4>(cp + 4, x1);
        } else {
            float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            st<u32x4>((bf16_t*)p.C + (long)m * p.ldc + n, pack8(v));
        }
    }
}

template <bool OUT_F32, int EPI = 0>
__device__ __forceinline__ void nt_epilogue_staged(const NTArgs& p, f32x4 (&acc)[2][2][4][2], unsigned char* S, int m0, int n0,
                                                   int tid, int wr, int wc, int l15, int g) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    st<f32x4>(S + (wr * 64 + i * 16 + l15) * QSTAGE_ROW + (b * 128 + wc * 32 + j * 16 + 4 * g) * 4, acc[a][b][i][j]);
        __syncthreads();
        nt_stage_readback<OUT_F32, 128, 256, 512, EPI>(p, S, m0 + a * 128, n0, tid);
        if (a == 0) __syncthreads();                     // the second half overwrites the staging rows
    }
}

// the same for the 128 x 128 kernel (4 waves of 64 x 64, acc[i][j]: row wm*64 + i*16 + l15, columns wn*64 + j*16 + 4g ..)
template <bool OUT_F32>
__device__ __forceinline__ void nt_epilogue_staged_128(const NTArgs& p, f32x4 (&acc)[4][4], unsigned char* S, int m0, int n0,
                                                       int tid, int wm, int wn, int l15, int g) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            st<f32x4>(S + (wm * 64 + i * 16 + l15) * GSTAGE_ROW + (wn * 64 + j * 16 + 4 * g) * 4, acc[i][j]);
    __syncthreads();
    nt_stage_readback<OUT_F32, 128, 128, 256>(p, S, m0, n0, tid);
}

// General NT kernel (any K multiple of 8, e.g. the B x (4 L D) conditioning GEMM): operands staged through VGPRs
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(NTArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BM * BK * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tm, tn, tile_m, tile_n, p.group);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    nt_bind_output(p, n0);
    const int K = p.K1 + p.K2;
    const int nk = (K + BK - 1) / BK;

    u32x4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int chunk = tid + c * 256;
            int row = chunk >> 3, slot = chunk & 7;
            int k = k0 + slot * 8;
            int m = m0 + row, n = n0 + row;
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (k < K) {
                if (m < p.M) {
                    const bf16_t* src = (k < p.K1) ? p.A1 + (long)m * p.lda1 + k : p.A2 + (long)m * p.lda2 + (k - p.K1);
                    va = ld<u32x4>(src);
                }
                if (n < p.N) vb = ld<u32x4>(p.B + (long)n * p.ldb + k);
            }
            ra[c] = va;
            rb[c] = vb;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int chunk = tid + c * 256;
            int row = chunk >> 3, slot = chunk & 7;
            int off = row * 128 + ((slot ^ (row & 7)) << 4);
            st<u32x4>(&smem[buf][0][off], ra[c]);
            st<u32x4>(&smem[buf][1][off], rb[c]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const unsigned char* As = smem[buf][0];
        const unsigned char* Bs = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bw[4];
            const int slot = kk * 4 + g;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wm * 64 + i * 16 + l15;
                af[i] = ld<bf16x8>(As + row * 128 + ((slot ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = wn * 64 + j * 16 + l15;
                bw[j] = ld<bf16x8>(Bs + row * 128 + ((slot ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    nt_epilogue<OUT_F32, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, l15, g);
}

// Default NT kernel: global_load_lds staging, two 32-KB LDS buffers, BK = 64, one barrier per K step.
// All per-lane addressing is hoisted out of the K loop: the 8 source pointers (4 A rows + 4 B rows per lane, swizzled
// source column) advance by one scalar add per step, the 16 fragment-read offsets are loop-invariant VGPRs and the
// loop is unrolled by two so that the LDS buffer select is an immediate offset (the first version of this loop spent
// ~50 VALU instructions per 32 MFMAs on address arithmetic: PMC showed 3.6 VALU per MFMA and MFMA busy at 25 %).
template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(NTArgs p) {
    // two K-tile buffers (64 KB); the staged epilogue needs 66 KB (two workgroups per CU still fit the 160 KB)
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[GSTAGE_BYTES > 4 * BM * BK * 2 ? GSTAGE_BYTES : 4 * BM * BK * 2];
    unsigned char (*smem)[2][BM * BK * 2] = reinterpret_cast<unsigned char (*)[2][BM * BK * 2]>(smem_raw);
    lds_declare(smem_raw, sizeof(smem_raw));
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    int tile_m, tile_n;
    const int nk1 = p.K1 / BK, nk = (p.K1 + p.K2) / BK;
    int kb = 0, ke = nk, part = -1;          // K-step range of this workgroup; part >= 0: partial result slot in ws
    if ((int)blockIdx.x < p.full) {
        tile_coords(xcd_remap(blockIdx.x, p.full), tm, tn, tile_m, tile_n, p.group);
    } else {
        part = blockIdx.x - p.full;
        const int r = part / p.split, sidx = part - r * p.split;
        tile_coords(p.full + r, tm, tn, tile_m, tile_n, p.group);
        kb = (int)((long)nk * sidx / p.split);
        ke = (int)((long)nk * (sidx + 1) / p.split);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    nt_bind_output(p, n0);

    // per-lane 32-bit byte offsets of the 4 (A) + 4 (B) wave instructions of a K step at k = 0; the K advance goes
    // into the scalar base, so a load is `global_load_lds_dwordx4 voff, s[base]` with no per-step vector arithmetic
    unsigned va1[4], va2[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (row & 7);
        const int m = min(m0 + row, p.M - 1), n = min(n0 + row, p.N - 1);
        va1[i] = (unsigned)(((long)m * p.lda1 + c * 8) * 2);
        va2[i] = p.K2 ? (unsigned)(((long)m * p.lda2 + c * 8) * 2) : 0u;
        vb[i] = (unsigned)(((long)n * p.ldb + c * 8) * 2);
    }
    // fragment read offsets (bytes inside one operand tile): [kk][i]
    int offa[2][4], offb[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = wm * 64 + i * 16 + l15, rb = wn * 64 + i * 16 + l15;
            offa[kk][i] = ra * 128 + (((kk * 4 + g) ^ (ra & 7)) << 4);
            offb[kk][i] = rb * 128 + (((kk * 4 + g) ^ (rb & 7)) << 4);
        }

    auto gissue = [&](int kt, int buf) {
        const char* sb = (const char*)p.B + (long)kt * (BK * 2);
        if (kt < nk1) {                       // wave-uniform branch; no register-array select (would go to scratch)
            const char* sa = (const char*)p.A1 + (long)kt * (BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(sa + va1[i], &smem[buf][0][(wave * 4 + i) * 1024]);
        } else {
            const char* sa = (const char*)p.A2 + (long)(kt - nk1) * (BK * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(sa + va2[i], &smem[buf][0][(wave * 4 + i) * 1024]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(sb + vb[i], &smem[buf][1][(wave * 4 + i) * 1024]);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const unsigned char* As = smem[buf][0];
        const unsigned char* Bs = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = ld<bf16x8>(As + offa[kk][i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[j] = ld<bf16x8>(Bs + offb[kk][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    gissue(kb, 0);
    __syncthreads();
    int kt = kb;
    if (p.probe == 0) {
        for (; kt + 1 < ke; kt += 2) {
            gissue(kt + 1, 1);
            compute(0);
            __syncthreads();
            if (kt + 2 < ke) gissue(kt + 2, 0);
            compute(1);
            __syncthreads();
        }
    } else {
        // bottleneck probes: the same loop without its loads (LDS + MFMA side alone) or without its MFMAs (load side)
        const bool loads = !(p.probe & E2K_GEMM_PROBE_NO_LOADS), math = !(p.probe & E2K_GEMM_PROBE_NO_MATH);
        for (; kt + 1 < ke; kt += 2) {
            if (loads) gissue(kt + 1, 1);
            if (math) compute(0);
            __syncthreads();
            if (loads && kt + 2 < ke) gissue(kt + 2, 0);
            if (math) compute(1);
            __syncthreads();
        }
    }
    if (kt < ke) compute(0);
    if (part >= 0) {        // K-range partial of a remainder tile: raw accumulators, 64 floats per thread
        float* w = p.ws + ((long)part * 16 * 256 + tid) * 4;       // [part][i*4+j][tid] x 4 floats: 1-KB runs per wave store
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) st<f32x4>(w + (i * 4 + j) * 1024, acc[i][j]);
        return;                                               // (gemm_nt_fixup_kernel finishes the tile)
    }
    if (p.staged) {                          // (the loop's last __syncthreads / compute has drained every DMA and fragment read)
        __syncthreads();
        nt_epilogue_staged_128<OUT_F32>(p, acc, smem_raw, m0, n0, tid, wm, wn, l15, g);
        return;
    }
    nt_epilogue<OUT_F32, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, l15, g);
}

// sums the K-range partials of a 16-row group (blockIdx.y = i) of one remainder tile (blockIdx.x), same thread <->
// accumulator mapping as the GEMM kernel, and runs the shared epilogue on the totals
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_nt_fixup_kernel(NTArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    const int i = blockIdx.y;
    int tile_m, tile_n;
    tile_coords(p.full + blockIdx.x, tm, tn, tile_m, tile_n, p.group);
    nt_bind_output(p, tile_n * BN);
    f32x4 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w = p.ws + (((long)blockIdx.x * p.split * 16 + i * 4) * 256 + tid) * 4;
#pragma unroll 4
    for (int sidx = 0; sidx < p.split; ++sidx) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] += ld<f32x4>(w + ((long)sidx * 16 + j) * 1024);
    }
    nt_epilogue<OUT_F32, 1>(p, acc, tile_m * BM + wm * 64 + i * 16, tile_n * BN + wn * 64, l15, g);
}

// x * Phi(x) (erf GELU) for the GEGLU epilogue below.  The kernel runs one workgroup per CU, so its epilogue is not hidden
// behind anything: the library erff (two branches, a full-range expf; ~40 VALU instructions per value) would cost more
// there than the separate GEGLU pass it replaces.  Abramowitz & Stegun 7.1.26 instead (|error| <= 1.5e-7 in erf, one
// v_rcp_f32 + one v_exp_f32 + 5 fma):  erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z),
// z = |x| / sqrt 2.  With the halved coefficients y = erfc(z) / 2:  Phi(x) = y for x < 0 (no 1 + erf cancellation) and
// 1 - y otherwise.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
    float y = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    y = fmaf(t, y, 0.5f * 1.421413741f);
    y = fmaf(t, y, 0.5f * -0.284496736f);
    y = fmaf(t, y, 0.5f * 0.254829592f);
    y = y * t * fast_exp2(x * x * (-0.5f * 1.4426950408889634f));
    return x * (x < 0.f ? y : 1.f - y);
}

// GEGLU epilogue of the 256 x 256 kernel (FeedForward(glu=True), e2_tts.py:646,692: `x, gate = proj(x).chunk(2);
// x * gelu(gate)` + Dropout).  The kernel stages W1 rows [n0, n0+128) as its "B low" half and rows [F + n0, F + n0+128)
// as its "B high" half, so a lane's accumulators au / ag hold the value and the gate of the SAME (row, 4 columns).
// The pre-activation is rounded to bf16 first (and stored for the backward when C is set), the product is formed from
// the rounded values: the output is e2k_geglu_fwd of the stored H up to the 1.5e-7 of gelu_erf_fast (below bf16 rounding
// except for the odd last-place flip).
template <int NI>
__device__ __forceinline__ void nt_epilogue_glu(const NTArgs& p, f32x4 (&au)[NI][2], f32x4 (&ag)[NI][2], int mw, int nw, int l15, int g) {
    const int F = p.N >> 1;
    const unsigned seed = p.seed_dev ? *p.seed_dev : p.seed;
    const int nb = nw + 4 * g;
    f32x4 bu[2], bg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bu[j] = bg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { bu[j][r] = p.bias[nb + j * 16 + r]; bg[j][r] = p.bias[F + nb + j * 16 + r]; }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = mw + i * 16 + l15;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb + j * 16;
            const f32x4 xu = au[i][j] + bu[j], xg = ag[i][j] + bg[j];
            float u[4] = {xu[0], xu[1], xu[2], xu[3]}, gt[4] = {xg[0], xg[1], xg[2], xg[3]};
            const u32x2 hu = pack4(u), hg = pack4(gt);
            if (p.C) {
                st<u32x2>((bf16_t*)p.C + (long)m * p.ldc + n, hu);
                st<u32x2>((bf16_t*)p.C + (long)m * p.ldc + F + n, hg);
            }
            unpack4(hu, u);
            unpack4(hg, gt);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ks = p.thresh ? keep_scale(seed, p.stream_id, m, n + r, p.thresh, p.inv_keep) : 1.f;
                o[r] = u[r] * gelu_erf_fast(gt[r]) * ks;
            }
            st<u32x2>(p.glu_out + (long)m * p.ldg + n, pack4(o));
        }
    }
}

// Inference form of the GEGLU epilogue (H not stored) through LDS: the products of half a tile (128 rows x 128 activation
// columns, fp32, row stride 528 B) are staged and leave as one 16-byte store per lane in whole-line row segments, like
// nt_epilogue_staged (the epilogue is bound by the number of store instructions per CU).  Same values as nt_epilogue_glu.
template <int UNUSED = 0>
__device__ __forceinline__ void nt_epilogue_glu_staged(const NTArgs& p, f32x4 (&acc)[2][2][4][2], unsigned char* S, int m0, int n0,
                                                       int tid, int wr, int wc, int l15, int g) {
    const int F = p.N >> 1;
    const unsigned seed = p.seed_dev ? *p.seed_dev : p.seed;
    const int nb = n0 + wc * 32 + 4 * g;
    f32x4 bu[2], bg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bu[j] = bg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { bu[j][r] = p.bias[nb + j * 16 + r]; bg[j][r] = p.bias[F + nb + j * 16 + r]; }
        }
    }
    const int c = tid & 15, rsub = tid >> 4;              // read-back: 16 lanes x 8 columns per row, 32 rows per pass
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wr * 64 + i * 16 + l15, m = m0 + a * 128 + row;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = nb + j * 16;
                const f32x4 xu = acc[a][0][i][j] + bu[j], xg = acc[a][1][i][j] + bg[j];
                float u[4] = {xu[0], xu[1], xu[2], xu[3]}, gt[4] = {xg[0], xg[1], xg[2], xg[3]};
                const u32x2 hu = pack4(u), hg = pack4(gt);           // (the product is formed from the bf16-rounded pre-activation)
                unpack4(hu, u);
                unpack4(hg, gt);
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ks = p.thresh ? keep_scale(seed, p.stream_id, m, n + r, p.thresh, p.inv_keep) : 1.f;
                    o[r] = u[r] * gelu_erf_fast(gt[r]) * ks;
                }
                st<f32x4>(S + row * GSTAGE_ROW + (wc * 32 + j * 16 + 4 * g) * 4, o);
            }
        }
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 32 + rsub, m = m0 + a * 128 + r;
            if (m >= p.M) continue;
            const f32x4 x0 = ld<f32x4>(S + r * GSTAGE_ROW + c * 32), x1 = ld<f32x4>(S + r * GSTAGE_ROW + c * 32 + 16);
            float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            st<u32x4>(p.glu_out + (long)m * p.ldg + n0 + 8 * c, pack8(v));
        }
        if (a == 0) __syncthreads();
    }
}

// 256 x 256 x 64 tile, EIGHT waves (512 threads, one workgroup per CU, two waves per SIMD), 128 KB of LDS, 8 phases per
// pair of K tiles (cdna_hip_programming.md "256^2 8-phase template", rebuilt for this kernel's operand layout and
// epilogue).  Default for shapes that fill the chip with 256 x 256 tiles; on MI355X 790-1011 TFLOP/s on the cfg3 shapes
// against 655-868 of the 128 x 128 kernel (profiles/r02_gemm_t256_first_hw_run.json), bit-repeatable over 50 launches.
//
// Why: with 128 x 128 tiles a K step moves 32 KB through the texture path for 2.1 MFLOP, which takes the load side as
// long as the MFMAs; a 256 x 256 tile moves 64 KB for 8.4 MFLOP, half the bytes per flop.
//
// LDS: two K-tile buffers of four 16-KB half tiles [A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255], rows
// of 128 B with the same source-side XOR swizzle as the other NT kernels.  Wave (wr, wc) = (wave >> 2, wave & 3) owns rows
// wr*64..+63 of BOTH A halves and columns wc*32..+31 of BOTH B halves, i.e. four 64 x 32 quadrants (Alo|Ahi) x (Blo|Bhi),
// so that every wave touches the same half tiles in the same phase:
//
//   phase 1: read Alo (8 x ds_read_b128) + Blo (4)   MFMA Alo x Blo        phase 3: read Ahi (8)   MFMA Ahi x Bhi
//   phase 2: read Bhi (4)                            MFMA Alo x Bhi        phase 4: --             MFMA Ahi x Blo (kept)
//
// Each phase is [ds_reads, counted vmcnt, one half-tile prefetch (2 global_load_lds per lane)] barrier [16 MFMAs]
// barrier.  Waves 4-7 run ONE BARRIER BEHIND waves 0-3 (they take an extra barrier first, waves 0-3 one at the end): a
// SIMD holds wave w and wave w + 4, so while one of them issues its 16 MFMAs (256 cycles of the pipe, priority raised)
// the other does its LDS reads and prefetch -- the barriers are the hand-over.
//
// Prefetch order (one half tile per phase, sequence element e = 4*tile + {0: Alo, 1: Blo, 2: Bhi, 3: Ahi}, element g + 6
// is issued in phase g): phase 1 of tile t stages Bhi(t+1), phase 2 Ahi(t+1), phase 3 Alo(t+2), phase 4 Blo(t+2).
//   * WAR: a half tile is restaged at least two phases after its last ds_read (Alo, Blo: read in phase 1, restaged in
//     phases 3 / 4; Bhi: read in 2, restaged in phase 1 of the next tile; Ahi: read in 3, restaged in phase 2 of the next
//     tile).  Two phases = at least one barrier between the trailing group's lgkmcnt wait and the leading group's issue.
//   * RAW: the vmcnt before the issue of phase g leaves at most three half tiles (6 loads) in flight, i.e. elements
//     <= g + 2 have landed for THIS wave; the barrier that follows publishes that, and the element is first read in
//     phase g + 1 or later (phase 1 of tile t reads elements 4t, 4t+1; phase 2 reads 4t+2; phase 3 reads 4t+3).
//   * In the last five phases nothing is left to issue, and the count is lowered step by step (4, 2, 0).
constexpr int QBM = 256, QBN = 256, QHALF = 128 * BK * 2, QBUF = 4 * QHALF, QTHREADS = 512;

// EPI 1: GEGLU backward epilogue (e2k_gemm_nt_geglu_bwd_bf16); EPI 2: rotary q / k epilogue (e2k_gemm_nt_qkrot_bf16, staged only)
template <bool OUT_F32, bool GLU = false, int EPI = 0>
__global__ __launch_bounds__(QTHREADS, 1) void gemm_nt_256_kernel(NTArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * QBUF > QSTAGE_BYTES) ? 2 * QBUF : QSTAGE_BYTES];
    lds_declare(smem, sizeof(smem));
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + QBM - 1) / QBM, tn = (p.N + QBN - 1) / QBN;      // GLU: N = 2F, F % 128 == 0: tn = F / 128 tiles of 128 value + 128 gate columns
    int tile_m, tile_n;
    const int nk1 = p.K1 / BK, nk = (p.K1 + p.K2) / BK;
    int kb = 0, ke = nk, part = -1;
    // The K-split remainder workgroups come LAST in the grid.  Putting them first (they run 1 / split of a K loop and stagger a quarter of
    // the CUs against the store burst of the full tiles) was landed early in round 5 on a round-4 measurement of -0.8 % and taken out
    // again: same box, NT 37.15 / 37.38 ms first against 37.45 / 37.40 last, the step equal, and 12 % MORE HBM fetch (FETCH_SIZE over the
    // step's launch mix 1.89 against 1.69 G units, twice each; profiles/r05t_rem_first_vs_last_fetch_ab.txt)
    if ((int)blockIdx.x < p.full) {
        tile_coords(xcd_remap((int)blockIdx.x, p.full), tm, tn, tile_m, tile_n, p.group);
    } else {
        part = blockIdx.x - p.full;
        const int r = part / p.split, sidx = part - r * p.split;
        tile_coords(p.full + r, tm, tn, tile_m, tile_n, p.group);
        kb = (int)((long)nk * sidx / p.split);
        ke = (int)((long)nk * (sidx + 1) / p.split);
    }
    const int m0 = tile_m * QBM, n0 = GLU ? tile_n * 128 : tile_n * QBN;
    if (!GLU) nt_bind_output(p, n0);
    const int nhalf = GLU ? (p.N >> 1) : 128;                // B rows between the "low" and the "high" half tile
    const int nt = ke - kb;                                  // K tiles of this workgroup (>= 1)

    // staging: a half tile is 16 wave instructions of 8 rows; wave w issues rows (2w + u)*8 + (lane >> 3), u = 0, 1
    unsigned va[2][2], dv[2][2], vb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (wave * 2 + u) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            const int m = min(m0 + h * 128 + row, p.M - 1), n = min(n0 + h * nhalf + row, p.N - 1);
            va[h][u] = (unsigned)(((long)m * p.lda1 + c * 8) * 2);
            dv[h][u] = p.K2 ? (unsigned)(((long)m * p.lda2 + c * 8) * 2) - va[h][u] : 0u;
            vb[h][u] = (unsigned)(((long)n * p.ldb + c * 8) * 2);
        }
    // fragment read offsets inside a half tile
    int offa[2][4], offb[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ra = wr * 64 + i * 16 + l15;
            offa[kk][i] = ra * 128 + (((kk * 4 + g) ^ (ra & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = wc * 32 + j * 16 + l15;
            offb[kk][j] = rb * 128 + (((kk * 4 + g) ^ (rb & 7)) << 4);
        }
    }
    unsigned char* const S0 = &smem[0];
    // half-tile slots of a buffer: 0 = Alo, 1 = Ahi, 2 = Blo, 3 = Bhi.  `tile` is relative to kb; tiles past the end of
    // the K range are simply not staged (the waits below count what was really issued)
    auto stage_a = [&](int tile, int h) __attribute__((always_inline)) {
        if (tile >= nt) return;
        const int kt = kb + tile;
        const bool first = kt < nk1;                     // wave-uniform
        const char* sa = first ? (const char*)p.A1 + (long)kt * (BK * 2) : (const char*)p.A2 + (long)(kt - nk1) * (BK * 2);
        const unsigned sel = first ? 0u : ~0u;
        unsigned char* dst = S0 + (tile & 1) * QBUF + h * QHALF + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(sa + (va[h][u] + (dv[h][u] & sel)), dst + u * 1024);
    };
    auto stage_b = [&](int tile, int h) __attribute__((always_inline)) {
        if (tile >= nt) return;
        const char* sb = (const char*)p.B + (long)(kb + tile) * (BK * 2);
        unsigned char* dst = S0 + (tile & 1) * QBUF + (2 + h) * QHALF + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(sb + vb[h][u], dst + u * 1024);
    };
    // `left` = half tiles that may stay in flight (issued after the youngest one the next phase reads)
    auto wait_landed = [&](int left) __attribute__((always_inline)) {
        if (left >= 3) wait_vmcnt<6>();
        else if (left == 2) wait_vmcnt<4>();
        else if (left == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };

    f32x4 acc[2][2][4][2];                  // [A half][B half][m16][n16]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ar[2][4], blo[2][2], bhi[2][2];

    auto read_a = [&](const unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) ar[kk][i] = ld<bf16x8>(S + offa[kk][i]);
    };
    auto read_b = [&](bf16x8 (&b)[2][2], const unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[kk][j] = ld<bf16x8>(S + offb[kk][j]);
    };
    auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
        set_prio<1>();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[kk][j], ar[kk][i], c[i][j], 0, 0, 0);
        set_prio<0>();
    };

    // prologue: sequence elements 0..5 = all of tile 0, then Alo, Blo of tile 1; elements 0, 1 must have landed
    stage_a(0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 1); stage_a(1, 0); stage_b(1, 0);
    if (nt >= 2) wait_vmcnt<8>();
    else wait_vmcnt<4>();
    barrier_raw();
    if (wr == 1) barrier_raw();              // waves 4-7 trail by one barrier from here on

    for (int t = 0; t < nt; ++t) {
        const unsigned char* S = S0 + (t & 1) * QBUF;
        const int left = 4 * (nt - t) - 3;   // phase g = 4t + k: 4 nt - g - 3
        // phase 1
        read_b(blo, S + 2 * QHALF);
        sched_fence();
        read_a(S);
        wait_landed(left);
        stage_b(t + 1, 1);
        barrier_raw();
        mma(acc[0][0], blo);
        barrier_raw();
        // phase 2
        read_b(bhi, S + 3 * QHALF);
        wait_landed(left - 1);
        stage_a(t + 1, 1);
        barrier_raw();
        mma(acc[0][1], bhi);
        barrier_raw();
        // phase 3
        read_a(S + QHALF);
        wait_landed(left - 2);
        stage_a(t + 2, 0);
        barrier_raw();
        mma(acc[1][1], bhi);
        barrier_raw();
        // phase 4
        wait_landed(left - 3);
        stage_b(t + 2, 0);
        barrier_raw();
        mma(acc[1][0], blo);
        barrier_raw();
    }
    if (wr == 0) barrier_raw();              // pairs with the extra barrier waves 4-7 took at the start

    if (part >= 0) {        // K-range partial of a remainder tile: [part][(a*2+b)*8 + i*2 + j][tid] x 4 floats
        float* w = p.ws + ((long)part * 32 * QTHREADS + tid) * 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) st<f32x4>(w + (((a * 2 + b) * 4 + i) * 2 + j) * (QTHREADS * 4), acc[a][b][i][j]);
        return;                                               // (the fix-up kernels finish the tile)
    }
    if (GLU) {
        if (p.staged) {                      // inference form (H not stored), every row of the tile's 128 activation columns exists
            __syncthreads();
            nt_epilogue_glu_staged(p, acc, smem, m0, n0, tid, wr, wc, l15, g);
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) nt_epilogue_glu<4>(p, acc[a][0], acc[a][1], m0 + a * 128 + wr * 64, n0 + wc * 32, l15, g);
        return;
    }
    if (p.staged) {                          // (every DMA has landed and every fragment read has been waited for: the ring is free)
        __syncthreads();
        nt_epilogue_staged<OUT_F32, EPI>(p, acc, smem, m0, n0, tid, wr, wc, l15, g);
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            nt_epilogue<OUT_F32, 4, 2, EPI == 1>(p, acc[a][b], m0 + a * 128 + wr * 64, n0 + b * 128 + wc * 32, l15, g);
}

// blockIdx.x = remainder tile, blockIdx.y = (A half * 2 + B half) * 4 + m16 group
template <bool OUT_F32, bool GB = false>
__global__ __launch_bounds__(QTHREADS) void gemm_nt_256_fixup_kernel(NTArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + QBM - 1) / QBM, tn = (p.N + QBN - 1) / QBN;
    const int q = blockIdx.y >> 2, i = blockIdx.y & 3, a = q >> 1, b = q & 1;
    int tile_m, tile_n;
    tile_coords(p.full + blockIdx.x, tm, tn, tile_m, tile_n, p.group);
    nt_bind_output(p, tile_n * QBN);
    f32x4 acc[1][2];
    acc[0][0] = acc[0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w = p.ws + (((long)blockIdx.x * p.split * 32 + (q * 4 + i) * 2) * QTHREADS + tid) * 4;
#pragma unroll 4
    for (int sidx = 0; sidx < p.split; ++sidx) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][j] += ld<f32x4>(w + ((long)sidx * 32 + j) * (QTHREADS * 4));
    }
    nt_epilogue<OUT_F32, 1, 2, GB>(p, acc, tile_m * QBM + a * 128 + wr * 64 + i * 16, tile_n * QBN + b * 128 + wc * 32, l15, g);
}

// GEGLU fix-up: blockIdx.y = A half * 4 + m16 group; both B halves (value | gate) are summed by the same lane
__global__ __launch_bounds__(QTHREADS) void gemm_nt_256_fixup_glu_kernel(NTArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, g = lane >> 4;
    const int tm = (p.M + QBM - 1) / QBM, tn = (p.N + QBN - 1) / QBN;
    const int a = blockIdx.y >> 2, i = blockIdx.y & 3;
    int tile_m, tile_n;
    tile_coords(p.full + blockIdx.x, tm, tn, tile_m, tile_n, p.group);
    f32x4 acc[2][1][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        acc[b][0][0] = acc[b][0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* w = p.ws + (((long)blockIdx.x * p.split * 32 + ((a * 2 + b) * 4 + i) * 2) * QTHREADS + tid) * 4;
#pragma unroll 4
        for (int sidx = 0; sidx < p.split; ++sidx) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[b][0][j] += ld<f32x4>(w + ((long)sidx * 32 + j) * (QTHREADS * 4));
        }
    }
    nt_epilogue_glu<1>(p, acc[0], acc[1], tile_m * QBM + a * 128 + wr * 64 + i * 16, tile_n * 128 + wc * 32, l15, g);
}

// (A BK = 32 variant of this kernel -- three 16-KB LDS stages, loads two K steps ahead with counted s_waitcnt vmcnt,
// three workgroups per CU -- was measured 40-90 % SLOWER on MI355X: with 64-byte LDS rows every global_load_lds
// instruction fetches half cache lines, which doubles the request count on the load side that already takes as long
// as the MFMA side.  Removed; see DESIGN.md.)

// ---------------------------------------------------------------------------------------------- TN (wgrad)

constexpr int TBM = 64;          // reduction (token) rows per step
constexpr int TLD = 288;         // LDS row stride in bytes: 256 B of data + 32 B pad => tr-reads of 8 rows tile all 64 banks

struct TNArgs {
    const bf16_t* A; long lda;   // (M, N)  dY
    const bf16_t* B; long ldb;   // (M, K)  X
    float* C; long ldc;          // (N, K)
    float* ws;                   // [splits][N][K] partial tiles when splits > 1
    int M, N, K, splits, chunk;
    float* colsum; int cs_from;  // optional: colsum[n] += sum_m A[m][n] for n >= cs_from (bias gradient of the same dY)
    // dual-source operands (256 x 256 kernel only): columns n >= N1 of A come from A2 (M, N - N1), columns k >= K1 of B from
    // B2 (M, K - K1); N1 / K1 are multiples of 256, so a tile lies in one source.  A2 / B2 == nullptr: single source
    const bf16_t* A2; long lda2; int N1;
    const bf16_t* B2; long ldb2; int K1;
};

// A GROUP of weight-gradient GEMMs with the same token count M in one launch of the 256 x 256 kernel (e2k_gemm_tn_group_bf16):
// the tiles of all problems form one grid (x = tile over the whole group, y = token split), so that the small outputs of a
// layer (attention out 1024 x 1024 = 16 tiles, the text stream's 4-32 tiles) fill the chip together instead of one by one.
constexpr int TN_GROUP_MAX = 8;
struct TNProb {
    const bf16_t* A; long lda; const bf16_t* B; long ldb; float* C; long ldc;
    int N, K, tile0;             // tile0: index of this problem's first tile in the group's tile list
    float* colsum; int cs_from;
};
struct TNGroupArgs {
    TNProb prob[TN_GROUP_MAX];
    int n, M, splits, chunk, tiles;
    float* ws;
};
// problem that holds tile t of the group (wave-uniform; n <= 8: a linear scan)
__device__ __forceinline__ int tn_group_find(const TNGroupArgs& g, int t) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < TN_GROUP_MAX; ++k)
        if (k < g.n && t >= g.prob[k].tile0) i = k;
    return i;
}
__device__ __forceinline__ TNArgs tn_group_args(const TNGroupArgs& g, int i) {
    TNArgs p;
    const TNProb& q = g.prob[i];
    p.A = q.A; p.lda = q.lda; p.B = q.B; p.ldb = q.ldb; p.C = q.C; p.ldc = q.ldc; p.ws = g.ws;
    p.M = g.M; p.N = q.N; p.K = q.K; p.splits = g.splits; p.chunk = g.chunk;
    p.colsum = q.colsum; p.cs_from = q.cs_from;
    p.A2 = nullptr; p.lda2 = 0; p.N1 = 0; p.B2 = nullptr; p.ldb2 = 0; p.K1 = 0;
    return p;
}

template <bool USE_TR>
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* T, int kk, int col0, int q, int g) {
    bf16x8 f;
    if (USE_TR) {
        const unsigned char* p0 = T + (kk * 32 + 4 * g + (q >> 2)) * TLD + (col0 + (q & 3) * 4) * 2;
        s16x4_ lo = lds_read_tr16_b64(p0);
        s16x4_ hi = lds_read_tr16_b64(p0 + 16 * TLD);
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int row = kk * 32 + (j >> 2) * 16 + 4 * g + (j & 3);
            f[j] = ld<short>(T + row * TLD + (col0 + q) * 2);
        }
    }
    return f;
}

// acc[i][j]: C[n = n0+wn*64+j*16+q][k = k0+wk*64+i*16+4g+r]
// splits == 1: C += acc.  splits > 1: the partial tile goes to the workspace in FRAGMENT order,
//     ws[((split * tiles + tile) * 16 + i*4 + j) * 256 + tid]   (f32x4 units),
// i.e. every wave store is one contiguous 1-KB run (the row-major [split][N][K] layout of rounds 1-2 made it 16 rows x
// 64 bytes: partial cache lines, the store pattern that costs the NT kernel 10 us per 128 KB, profiles/r03_nt_store_burst.json);
// tn_reduce_frag_kernel maps fragments back to (n, k).  (fp32 atomics on C instead of a workspace ran at ~60 G atomics/s.)
__device__ __forceinline__ void tn_store(const TNArgs& p, f32x4 (&acc)[4][4], int tile, int tiles, int n0, int k0, int wn, int wk, int q, int g) {
    if (p.splits > 1) {
        f32x4* w = (f32x4*)p.ws + ((long)blockIdx.y * tiles + tile) * (16 * 256) + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[(i * 4 + j) * 256] = acc[i][j];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + q;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + wk * 64 + i * 16 + 4 * g;
            float* c = p.C + (long)n * p.ldc + k;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k + r < p.K) c[r] += acc[i][j][r];
        }
    }
}

template <bool USE_TR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TNArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * TLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int q = lane & 15, g = lane >> 4;
    const int tn = (p.N + 127) / 128, tk = (p.K + 127) / 128;
    int tile_n, tile_k;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tn, tk, tile_n, tile_k);
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int mbeg = blockIdx.y * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    const int nsteps = (mend - mbeg + TBM - 1) / TBM;

    u32x4 ra[4], rb[4];
    auto gload = [&](int s) {
        const int mb = mbeg + s * TBM;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int chunk = tid + c * 256;
            int row = chunk >> 4, slot = chunk & 15;
            int m = mb + row;
            u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u};
            if (m < mend) {
                int n = n0 + slot * 8, k = k0 + slot * 8;
                if (n < p.N) va = ld<u32x4>(p.A + (long)m * p.lda + n);
                if (k < p.K) vb = ld<u32x4>(p.B + (long)m * p.ldb + k);
            }
            ra[c] = va;
            rb[c] = vb;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int chunk = tid + c * 256;
            int row = chunk >> 4, slot = chunk & 15;
            st<u32x4>(&smem[buf][0][row * TLD + slot * 16], ra[c]);
            st<u32x4>(&smem[buf][1][row * TLD + slot * 16], rb[c]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (nsteps > 0) {
        gload(0);
        sstore(0);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
        const unsigned char* At = smem[buf][0];
        const unsigned char* Bt = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fx[4], fy[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i] = tn_frag<USE_TR>(Bt, kk, wk * 64 + i * 16, q, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) fy[j] = tn_frag<USE_TR>(At, kk, wn * 64 + j * 16, q, g);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < nsteps) sstore(buf ^ 1);
        __syncthreads();
    }
    tn_store(p, acc, tile_n * tk + tile_k, tn * tk, n0, k0, wn, wk, q, g);
}

// Fast TN path (token count a multiple of 64): global_load_lds staging into unpadded 256-B LDS rows whose 16-B chunks
// are XOR-swizzled by 2*(row & 7) on the SOURCE side (conflict-free ds_read_b64_tr_b16: the 8 rows a half-wave reads
// land on 8 distinct chunk pairs of the 256-B bank row), scalar base + hoisted 32-bit lane offsets, two LDS buffers.
template <bool CS>       // CS: also accumulate the column sums of A (bias gradient) in the k-tile-0 workgroups
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(TNArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * 256];
    lds_declare(smem, sizeof(smem));
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int q = lane & 15, g = lane >> 4;
    const int tn = (p.N + 127) / 128, tk = (p.K + 127) / 128;
    int tile_n, tile_k;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tn, tk, tile_n, tile_k);
    const int n0 = tile_n * 128, k0 = tile_k * 128;
    const int mbeg = blockIdx.y * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    const int nsteps = (mend - mbeg) / TBM;

    // source offsets (bytes) of this lane for the 4 + 4 wave instructions of a step, at step 0
    unsigned va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (2 * (row & 7));
        const int cn = min(n0 + c * 8, ((p.N + 7) & ~7) - 8), ck = min(k0 + c * 8, ((p.K + 7) & ~7) - 8);
        va[i] = (unsigned)((((long)(mbeg + row)) * p.lda + cn) * 2);
        vb[i] = (unsigned)((((long)(mbeg + row)) * p.ldb + ck) * 2);
    }
    // fragment read offsets: row (4g + q/4) of a 16-row slab, chunk ((col0/8) ^ 2*(row&7)) + (q&3)/2, byte (q&1)*8
    const int r7 = (4 * g + (q >> 2)) & 7;
    int offx[4], offy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lanepart = (4 * g + (q >> 2)) * 256 + ((q & 3) >> 1) * 16 + (q & 1) * 8;
        offx[i] = lanepart + (((wk * 8 + 2 * i) ^ (2 * r7)) << 4);
        offy[i] = lanepart + (((wn * 8 + 2 * i) ^ (2 * r7)) << 4);
    }
    auto gissue = [&](int s, int buf) {
        const char* sa = (const char*)p.A + (long)s * TBM * p.lda * 2;
        const char* sb = (const char*)p.B + (long)s * TBM * p.ldb * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(sa + va[i], &smem[buf][0][(wave * 4 + i) * 1024]);
            glds16(sb + vb[i], &smem[buf][1][(wave * 4 + i) * 1024]);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Column sums of A (bias gradient) ride along in the k-tile-0 workgroups: one extra MFMA per 16 columns and K step
    // with an all-ones first operand (every row of the result is the column sum); the two waves that share an n range
    // (wk = 0 / 1) take two of its four 16-column groups each.
    const bool do_cs = CS && tile_k == 0 && n0 + 128 > p.cs_from;     // wave-uniform
    f32x4 cs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const short one = 0x3F80;
    const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};
    auto compute = [&](int buf) {
        const unsigned char* At = smem[buf][0];
        const unsigned char* Bt = smem[buf][1];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fx[4], fy[4];
            s16x4_ xl[4], xh[4], yl[4], yh[4];
            // (lds_tr_issue, not the builtin: the builtin makes the compiler drain the LDS-DMA prefetch of the next step
            // in front of every group of reads, see e2k_asm.h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_tr_issue(xl[i], Bt + offx[i], kk * 32 * 256);
                lds_tr_issue(xh[i], Bt + offx[i], (kk * 32 + 16) * 256);
                lds_tr_issue(yl[i], At + offy[i], kk * 32 * 256);
                lds_tr_issue(yh[i], At + offy[i], (kk * 32 + 16) * 256);
            }
            lds_tr_wait(xl[0], xh[0], xl[1], xh[1], xl[2], xh[2], xl[3], xh[3], yl[0], yh[0], yl[1], yh[1], yl[2], yh[2], yl[3], yh[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (concatenated with a shuffle: the two 64-bit results become one 128-bit register tuple without moves)
                fx[i] = __builtin_shufflevector(xl[i], xh[i], 0, 1, 2, 3, 4, 5, 6, 7);
                fy[i] = __builtin_shufflevector(yl[i], yh[i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[i], fy[j], acc[i][j], 0, 0, 0);
            if (do_cs) {
                if (wk == 0) {
                    cs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[0], cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[1], cs[1], 0, 0, 0);
                } else {
                    cs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[2], cs[0], 0, 0, 0);
                    cs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fy[3], cs[1], 0, 0, 0);
                }
            }
        }
    };
    if (nsteps > 0) gissue(0, 0);
    __syncthreads();
    int s = 0;
    for (; s + 1 < nsteps; s += 2) {
        gissue(s + 1, 1);
        compute(0);
        __syncthreads();
        if (s + 2 < nsteps) gissue(s + 2, 0);
        compute(1);
        __syncthreads();
    }
    if (s < nsteps) compute(0);
    tn_store(p, acc, tile_n * tk + tile_k, tn * tk, n0, k0, wn, wk, q, g);
    if (do_cs && g == 0) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int n = n0 + wn * 64 + (2 * wk + jj) * 16 + q;
            if (n < p.N && n >= p.cs_from) atomicAdd(p.colsum + n, cs[jj][0]);
        }
    }
}

// 256 x 256 output tile, EIGHT waves, 8 phases per pair of 64-row reduction steps: the weight-gradient sibling of
// gemm_nt_256_kernel (same half-tile ring, same phase / prefetch / counted-vmcnt schedule -- see the comment there; that
// schedule has run on MI355X).  Per reduction step of 64 token rows the workgroup stages four 16-KB half tiles
//   [A cols 0-127 | A cols 128-255 | B cols 0-127 | B cols 128-255],   A = dY (n columns), B = X (k columns),
// each 64 rows x 256 B with the 16-byte chunks XOR-swizzled by 2 * (row & 7) on the SOURCE side (the layout of
// gemm_tn_glds_kernel: conflict-free ds_read_b64_tr_b16).  Wave (wr, wc) = (wave >> 2, wave & 3) owns columns wr*64..+63 of
// BOTH A halves and columns wc*32..+31 of BOTH B halves, i.e. four 64 (n) x 32 (k) quadrants of the output tile.
// Half the LDS-read bytes per flop and a quarter of the partial-tile traffic per flop of the 128 x 128 kernel.
constexpr int T2 = 256, T2HALF = TBM * 256, T2BUF = 4 * T2HALF, T2THREADS = 512;

template <bool CS, class ARGS = TNArgs>
__global__ __launch_bounds__(T2THREADS, 1) void gemm_tn_256_kernel(ARGS args) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * T2BUF];
    lds_declare(smem, sizeof(smem));
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int q = lane & 15, g = lane >> 4;
    // single problem: the tile list is this problem's; group: locate the problem that owns this workgroup's tile
    constexpr bool GROUP = !std::is_same<ARGS, TNArgs>::value;
    const int gtile = xcd_remap(blockIdx.x, gridDim.x);          // (contiguous runs of the tile list per XCD)
    TNArgs p;
    int ltile, ws_tile0;
    if constexpr (GROUP) {
        const int pi = tn_group_find(args, gtile);
        p = tn_group_args(args, pi);
        ws_tile0 = args.prob[pi].tile0;
        ltile = gtile - ws_tile0;
    } else {
        p = args;
        ws_tile0 = 0;
        ltile = gtile;
    }
    const int ws_tiles = gridDim.x;
    const int tn = (p.N + T2 - 1) / T2, tk = (p.K + T2 - 1) / T2;
    int tile_n, tile_k;
    tile_coords(ltile, tn, tk, tile_n, tile_k);
    const int n0 = tile_n * T2, k0 = tile_k * T2;
    const int mbeg = blockIdx.y * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    const int nt = (mend - mbeg) / TBM;                       // reduction steps of this workgroup
    // operand sources of this tile (dual-source launches: the tile lies in exactly one column block of A and of B)
    const bool a2 = p.A2 != nullptr && n0 >= p.N1, b2 = p.B2 != nullptr && k0 >= p.K1;        // wave-uniform
    const bf16_t* const Asrc = a2 ? p.A2 : p.A;
    const bf16_t* const Bsrc = b2 ? p.B2 : p.B;
    const long lda = a2 ? p.lda2 : p.lda, ldb = b2 ? p.ldb2 : p.ldb;
    const int na0 = a2 ? n0 - p.N1 : n0, kb0 = b2 ? k0 - p.K1 : k0;                           // tile origin inside its source
    const int Na = p.A2 ? (a2 ? p.N - p.N1 : p.N1) : p.N, Kb = p.B2 ? (b2 ? p.K - p.K1 : p.K1) : p.K;   // columns of that source

    // staging: a half tile is 16 wave instructions of 4 rows; wave w issues rows (2w + u) * 4 + (lane >> 4), u = 0, 1
    unsigned va[2][2], vb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = (wave * 2 + u) * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (2 * (row & 7));
            const int cn = min(na0 + h * 128 + c * 8, ((Na + 7) & ~7) - 8), ck = min(kb0 + h * 128 + c * 8, ((Kb + 7) & ~7) - 8);
            va[h][u] = (unsigned)((((long)(mbeg + row)) * lda + cn) * 2);
            vb[h][u] = (unsigned)((((long)(mbeg + row)) * ldb + ck) * 2);
        }
    // fragment read offsets inside a half tile: row (4g + q/4) of a 16-row slab, chunk (col/8 ^ 2*(row&7)) + (q&3)/2, byte (q&1)*8
    const int r7 = (4 * g + (q >> 2)) & 7;
    const int lanepart = (4 * g + (q >> 2)) * 256 + ((q & 3) >> 1) * 16 + (q & 1) * 8;
    int offa[4], offb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offa[i] = lanepart + (((wr * 8 + 2 * i) ^ (2 * r7)) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) offb[j] = lanepart + (((wc * 4 + 2 * j) ^ (2 * r7)) << 4);

    unsigned char* const S0 = &smem[0];
    // half-tile slots of a buffer: 0 = Alo, 1 = Ahi, 2 = Blo, 3 = Bhi; steps past the end are simply not staged
    auto stage_a = [&](int step, int h) __attribute__((always_inline)) {
        if (step >= nt) return;
        const char* sa = (const char*)Asrc + (long)step * TBM * lda * 2;
        unsigned char* dst = S0 + (step & 1) * T2BUF + h * T2HALF + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(sa + va[h][u], dst + u * 1024);
    };
    auto stage_b = [&](int step, int h) __attribute__((always_inline)) {
        if (step >= nt) return;
        const char* sb = (const char*)Bsrc + (long)step * TBM * ldb * 2;
        unsigned char* dst = S0 + (step & 1) * T2BUF + (2 + h) * T2HALF + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) glds16(sb + vb[h][u], dst + u * 1024);
    };
    auto wait_landed = [&](int left) __attribute__((always_inline)) {
        if (left >= 3) wait_vmcnt<6>();
        else if (left == 2) wait_vmcnt<4>();
        else if (left == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };

    f32x4 acc[2][2][4][2];                  // [A half][B half][n16][k16]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments as pairs of 64-bit halves (rows 0-15 / 16-31 of a 32-row reduction slice), joined at the MFMA
    s16x4_ ar[2][4][2], blo[2][2][2], bhi[2][2][2];
    // column sums of A (bias gradient of the same dY) ride along in the k-tile-0 workgroups: wave (wr, wc) takes the
    // 16-column group i = wc of its 64 columns, one extra MFMA per A half and 32 rows with an all-ones operand
    const bool do_cs = CS && p.colsum != nullptr && tile_k == 0 && n0 + T2 > p.cs_from;          // wave-uniform (a group may mix problems with and without a bias gradient)
    f32x4 cs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const short one = 0x3F80;
    const bf16x8 ones = bf16x8{one, one, one, one, one, one, one, one};

    // Transposing reads go through lds_tr_issue (inline assembly), not the builtin: the compiler cannot tell which LDS
    // bytes the builtin reads and put an s_waitcnt vmcnt(0) in front of every read group -- five drains of the LDS-DMA
    // prefetch queue per iteration, next to the counted waits of the schedule (e2k_asm.h).  The reads of a phase are
    // issued before its barrier and waited for (lgkmcnt(0)) right before its MFMAs.
    auto read_a = [&](const unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds_tr_issue(ar[kk][i][0], S + offa[i], kk * 32 * 256);
                lds_tr_issue(ar[kk][i][1], S + offa[i], (kk * 32 + 16) * 256);
            }
    };
    auto read_b = [&](s16x4_ (&b)[2][2][2], const unsigned char* S) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                lds_tr_issue(b[kk][j][0], S + offb[j], kk * 32 * 256);
                lds_tr_issue(b[kk][j][1], S + offb[j], (kk * 32 + 16) * 256);
            }
    };
    auto landed_a = [&]() __attribute__((always_inline)) {
        lds_tr_wait(ar[0][0][0], ar[0][0][1], ar[0][1][0], ar[0][1][1], ar[0][2][0], ar[0][2][1], ar[0][3][0], ar[0][3][1],
                    ar[1][0][0], ar[1][0][1], ar[1][1][0], ar[1][1][1], ar[1][2][0], ar[1][2][1], ar[1][3][0], ar[1][3][1]);
    };
    auto landed_b = [&](s16x4_ (&b)[2][2][2]) __attribute__((always_inline)) {
        lds_tr_wait(b[0][0][0], b[0][0][1], b[0][1][0], b[0][1][1], b[1][0][0], b[1][0][1], b[1][1][0], b[1][1][1]);
    };
    auto join = [](const s16x4_ (&h)[2]) __attribute__((always_inline)) -> bf16x8 {
        return __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // acc[i][j]: C[n = n0 + a*128 + wr*64 + i*16 + q][k = k0 + b*128 + wc*32 + j*16 + 4g + r]
    auto mma = [&](f32x4 (&c)[4][2], const s16x4_ (&b)[2][2][2]) __attribute__((always_inline)) {
        set_prio<1>();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(b[kk][j]), join(ar[kk][i]), c[i][j], 0, 0, 0);
        set_prio<0>();
    };
    auto colsum_mma = [&](int a) __attribute__((always_inline)) {
        if (!do_cs) return;
        bf16x8 f0, f1;                                        // this wave's group i = wc of the A half it has just read
        switch (wc) {
            case 0: f0 = join(ar[0][0]); f1 = join(ar[1][0]); break;
            case 1: f0 = join(ar[0][1]); f1 = join(ar[1][1]); break;
            case 2: f0 = join(ar[0][2]); f1 = join(ar[1][2]); break;
            default: f0 = join(ar[0][3]); f1 = join(ar[1][3]); break;
        }
        cs[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f0, cs[a], 0, 0, 0);
        cs[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f1, cs[a], 0, 0, 0);
    };

    if (nt > 0) {
        // prologue: sequence elements 0..5 = all of step 0, then Alo, Blo of step 1; elements 0, 1 must have landed
        stage_a(0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 1); stage_a(1, 0); stage_b(1, 0);
        if (nt >= 2) wait_vmcnt<8>();
        else wait_vmcnt<4>();
        barrier_raw();
        if (wr == 1) barrier_raw();              // waves 4-7 trail by one barrier from here on

        for (int t = 0; t < nt; ++t) {
            const unsigned char* S = S0 + (t & 1) * T2BUF;
            const int left = 4 * (nt - t) - 3;
            // phase 1
            read_b(blo, S + 2 * T2HALF);
            sched_fence();
            read_a(S);
            wait_landed(left);
            stage_b(t + 1, 1);
            barrier_raw();
            landed_b(blo);
            landed_a();
            mma(acc[0][0], blo);
            colsum_mma(0);
            barrier_raw();
            // phase 2
            read_b(bhi, S + 3 * T2HALF);
            wait_landed(left - 1);
            stage_a(t + 1, 1);
            barrier_raw();
            landed_b(bhi);
            mma(acc[0][1], bhi);
            barrier_raw();
            // phase 3
            read_a(S + T2HALF);
            wait_landed(left - 2);
            stage_a(t + 2, 0);
            barrier_raw();
            landed_a();
            mma(acc[1][1], bhi);
            colsum_mma(1);
            barrier_raw();
            // phase 4
            wait_landed(left - 3);
            stage_b(t + 2, 0);
            barrier_raw();
            mma(acc[1][0], blo);
            barrier_raw();
        }
        if (wr == 0) barrier_raw();              // pairs with the extra barrier waves 4-7 took at the start
    }

    // splits == 1: C += acc.  splits > 1: the partial tile in fragment order (see tn_store),
    //     ws[((split * tiles + tile) * 32 + ((a*2 + b)*4 + i)*2 + j) * 512 + tid]   (f32x4 units)
    if (p.splits > 1) {
        f32x4* w = (f32x4*)p.ws + ((long)blockIdx.y * ws_tiles + ws_tile0 + tile_n * tk + tile_k) * (32 * T2THREADS) + tid;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) w[(((a * 2 + b) * 4 + i) * 2 + j) * T2THREADS] = acc[a][b][i][j];
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + a * 128 + wr * 64 + i * 16 + q;
                if (n >= p.N) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = k0 + b * 128 + wc * 32 + j * 16 + 4 * g;
                        float* c = p.C + (long)n * p.ldc + k;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (k + r < p.K) c[r] += acc[a][b][i][j][r];
                    }
            }
    }
    if (do_cs && g == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int n = n0 + a * 128 + wr * 64 + wc * 16 + q;
            if (n < p.N && n >= p.cs_from) atomicAdd(p.colsum + n, cs[a][0]);
        }
    }
}

// (A pipelined variant of this kernel -- 32 token rows per step, four 16-KB LDS stages, loads three steps ahead with a
// counted s_waitcnt vmcnt -- was measured 20-28 % SLOWER on every cfg3 shape: PMC shows the waves of the kernel above
// parked at s_waitcnt / barriers 58 % of the time, but halving the MFMAs per barrier costs more than the deeper
// prefetch recovers.  Removed; see DESIGN.md.)

// sums the fragment-order partial tiles over the splits and adds the total to C.  One workgroup per 16-ROW SLAB of a tile:
// the fragments that hold those 16 output rows belong to the GEMM waves of one row group -- 2 waves x 4 fragments for the
// 128 x 128 kernels, 4 waves x 4 fragments for the 256 x 256 kernel -- and this kernel's threads stand in for exactly those
// waves (same lane <-> (n, k) mapping, so the 16-byte loads of the partials are contiguous over the lanes).  The slab's
// totals are transposed through LDS and C is read and written in whole rows.  Two earlier forms, both measured on MI355X:
// a fragment-shaped read-modify-write of C (16 rows x 64 bytes per wave instruction) lost 10-15 % on large outputs with few
// splits (FeedForward's second weight, qkv); one workgroup per whole tile left 16-64 workgroups on the chip for the small
// outputs and ran 2-3x slower (profiles/r03_gemm_epilogue_ab_16byte_stores.json, `tn` rows).
//   BIG  (256 x 256 tiles): grid (tiles, 16 slabs), 256 threads;   otherwise (128 x 128): grid (tiles, 8 slabs), 128 threads
struct TNReduceGroup { TNProb prob[TN_GROUP_MAX]; int n; };

template <bool BIG, bool GROUP = false>
__global__ __launch_bounds__(BIG ? 256 : 128) void tn_reduce_frag_kernel(const f32x4* ws, float* C, long ldc, int N, int K, int splits, int tk,
                                                                         TNReduceGroup grp) {
    constexpr int NF = BIG ? 32 : 16, T = BIG ? 512 : 256, TS = BIG ? 256 : 128, ROWB = TS * 4 + 16, RT = BIG ? 256 : 128;
    __shared__ __attribute__((aligned(16))) unsigned char S[16 * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wl = tid >> 6, q = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, tiles = gridDim.x, slab = blockIdx.y;
    const long stride = (long)tiles * NF * T;
    int ltile = tile;
    if (GROUP) {                             // the problem that owns this tile of the group: its C, shape and tile grid
        int pi = 0;
#pragma unroll
        for (int k = 1; k < TN_GROUP_MAX; ++k)
            if (k < grp.n && tile >= grp.prob[k].tile0) pi = k;
        const TNProb& pr = grp.prob[pi];
        C = pr.C; ldc = pr.ldc; N = pr.N; K = pr.K; tk = (pr.K + TS - 1) / TS;
        ltile = tile - pr.tile0;
    }
    int n0 = (ltile / tk) * TS;
    const int k0 = (ltile % tk) * TS;
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
        int f, gtid, col;
        if (BIG) {                          // slab = (a*2 + wr)*4 + i: rows a*128 + wr*64 + i*16 + q; waves (wr, wc = wl), fragments (b, j) = ff
            const int a = slab >> 3, wr = (slab >> 2) & 1, i = slab & 3, b = ff >> 1, j = ff & 1;
            f = ((a * 2 + b) * 4 + i) * 2 + j;
            gtid = (wr * 4 + wl) * 64 + lane;
            col = b * 128 + wl * 32 + j * 16 + 4 * g;
        } else {                            // slab = wn*4 + j: rows wn*64 + j*16 + q; waves (wn, wk = wl), fragments i = ff
            const int wn = slab >> 2, jj = slab & 3;
            f = ff * 4 + jj;
            gtid = (wn * 2 + wl) * 64 + lane;
            col = wl * 64 + ff * 16 + 4 * g;
        }
        const f32x4* w = ws + ((long)tile * NF + f) * T + gtid;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int sp = 0; sp < splits; ++sp) s += w[sp * stride];
        st<f32x4>(S + q * ROWB + col * 4, s);
    }
    __syncthreads();
    n0 += BIG ? ((slab >> 3) * 128 + ((slab >> 2) & 1) * 64 + (slab & 3) * 16) : ((slab >> 2) * 64 + (slab & 3) * 16);
    constexpr int LPR = TS / 4, RPP = RT / LPR;           // lanes per row, rows per pass
    const int c = tid % LPR, rsub = tid / LPR;
    const int k = k0 + 4 * c;
    if (k >= K) return;
    const bool vec = k + 3 < K && (ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0;
#pragma unroll
    for (int pass = 0; pass < 16 / RPP; ++pass) {
        const int r = pass * RPP + rsub, n = n0 + r;
        if (n >= N) continue;
        const f32x4 s = ld<f32x4>(S + r * ROWB + c * 16);
        float* cp = C + (long)n * ldc + k;
        if (vec) {
            st<f32x4>(cp, ld<f32x4>(cp) + s);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + e < K) cp[e] += s[e];
        }
    }
}

int tn_splits(int M, int N, int K, int splits) {
    const int tn = (N + 127) / 128, tk = (K + 127) / 128;
    if (splits <= 0) {
        // 512 workgroups are resident (two per CU).  Pick the split count with the lowest estimated time: rounds of
        // workgroups x reduction steps per split (~0.9 us per 64-row step) + the fp32 partial-tile round trip through
        // the workspace (64 KB written and re-read per workgroup at ~5 TB/s); at least 512 token rows per split.
        const int tiles = tn * tk;
        int maxs = (M + 511) / 512;
        if (maxs > 64) maxs = 64;
        float best = 1e30f;
        splits = 1;
        for (int sp = 1; sp <= maxs; ++sp) {
            const int rounds = (tiles * sp + 511) / 512;
            const float steps = (float)((M + sp - 1) / sp + TBM - 1) / TBM;
            const float t = rounds * steps * 0.9f + (sp > 1 ? tiles * sp * 0.026f + 4.f : 0.f);
            if (t < best * 0.97f) { best = t; splits = sp; }
        }
    }
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TBM - 1) / TBM * TBM;
    return (M + chunk - 1) / chunk;
}

// token splits of the 256 x 256 weight-gradient kernel: fill the 256 workgroup slots (one per CU) once, at least 8
// reduction steps of 64 rows per split
int tn_splits_256(int M, int N, int K, int splits) {
    const int tiles = ((N + T2 - 1) / T2) * ((K + T2 - 1) / T2);
    if (splits <= 0) {
        splits = 256 / tiles;
        const int maxs = M / (TBM * 8);
        if (splits > maxs) splits = maxs;
        if (splits < 1) splits = 1;
    }
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TBM - 1) / TBM * TBM;
    return (M + chunk - 1) / chunk;
}

// which weight-gradient kernel runs for (M, N, K): use_tr 1 = choose, 2 = always 128 x 128, 3 = 256 x 256 wherever it can run
bool tn_use_256(int M, int N, int K, int use_tr) {
    if (use_tr < 1 || (M % TBM) != 0 || N < 8 || K < 8) return false;
    if (use_tr == 3) return true;
    if (use_tr == 2) return false;
    // measured on MI355X (profiles/r02_tn_ab.json, cfg3 shapes, 256 x 256 vs 128 x 128 in TFLOP/s): the large tile wins on
    // very wide outputs (8448 x 8192 x 1024: 965 vs 912) and on long token counts, where it halves the number of partial
    // tiles (33792 x 1024 x 1024: 853 vs 659; 33792 x 1024 x 512: 554 vs 537); it loses in between (8448 x 1024 x 4096:
    // 825 vs 858, 8448 x 3104 x 1024: 700 vs 754) and on narrow outputs (33792 x 512 x 512: 373 vs 383)
    const long nk = (long)N * K;
    return N >= 512 && K >= 512 && (nk >= 6l << 20 || (M >= 16384 && nk >= 512l << 10));
}

}  // namespace

// Remainder split (the last, partial round of tiles cut into K ranges + a fix-up launch): OFF by default since round 6.  Timed alone it
// gains 2-3 % on the NT launch mix of a cfg3 step (736 against 721 TFLOP/s); IN the step the CUs a partial round leaves idle are taken by
// the other launch lanes' kernels, and the 157 fix-up launches and their fp32 partial traffic are pure cost: 84.94 / 85.02 ms with the
// split against 84.56 / 84.41 without, same box, interleaved (profiles/r06f_nt_remainder_split_in_step_ab.txt).  E2K_GEMM_SPLIT turns it
// on (a caller that runs these GEMMs alone on the chip); the 8-slot test hook implies it.
static bool nt_split_on(int flags) { return (flags & (E2K_GEMM_SPLIT | E2K_GEMM_TEST_SLOTS8)) && !(flags & E2K_GEMM_NO_SPLIT); }

// smallest number of 256 x 256 output tiles for which e2k_gemm_nt_bf16 takes the 256 x 256 kernel (E2K_GEMM_T256_MIN overrides, A/B)
static int nt_t256_min() {
    static const int v = getenv("E2K_GEMM_T256_MIN") ? atoi(getenv("E2K_GEMM_T256_MIN")) : 64;
    return v;
}

static int gemm_nt_bf16_impl(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                                const void* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int accumulate,
                                int M, int N, const float* bias, const float* colscale, int64_t lds,
                                int rows_per_batch, const uint8_t* rowmask, const void* resid, int64_t ldr, int flags,
                                float* ws, int64_t ws_bytes, int nsplit, void* C2, int64_t ldc2, const void* resid2, int64_t ldr2,
                                void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if (nsplit < 0 || nsplit >= N || (nsplit > 0 && ((nsplit % QBN) || !C2 || (ldc2 & 7) || (ldr2 & 7) || ((uintptr_t)C2 & 15) ||
                                                    ((uintptr_t)resid2 & 15) || (resid != nullptr) != (resid2 != nullptr))))
        return E2K_ERR_ARG;
    if (K1 <= 0 || (K1 & 7) || (K2 & 7) || K2 < 0) return E2K_ERR_SHAPE;
    if ((lda1 & 7) || (ldb & 7) || (K2 > 0 && ((lda2 & 7) || A2 == nullptr))) return E2K_ERR_ALIGN;
    if (((uintptr_t)A1 | (uintptr_t)B | (uintptr_t)A2) & 15) return E2K_ERR_ALIGN;
    if (colscale && rows_per_batch <= 0) return E2K_ERR_SHAPE;
    if (accumulate && !out_f32) return E2K_ERR_SHAPE;
    NTArgs p{};
    p.A1 = (const bf16_t*)A1; p.lda1 = lda1; p.K1 = K1;
    p.A2 = (const bf16_t*)A2; p.lda2 = lda2; p.K2 = K2;
    p.B = (const bf16_t*)B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.accumulate = accumulate;
    p.M = M; p.N = N;
    p.bias = bias; p.colscale = colscale; p.lds = lds; p.rows_per_batch = rows_per_batch;
    p.rowmask = rowmask; p.resid = (const bf16_t*)resid; p.ldr = ldr;
    p.nsplit = nsplit; p.ldc2 = ldc2; p.ldr2 = ldr2;
    p.C2 = nsplit ? (void*)((char*)C2 - (int64_t)nsplit * (out_f32 ? 4 : 2)) : nullptr;
    p.resid2 = (nsplit && resid2) ? (const bf16_t*)resid2 - nsplit : nullptr;
    const int tn = (N + BN - 1) / BN;
    p.group = 8;
    p.probe = flags & (E2K_GEMM_PROBE_NO_LOADS | E2K_GEMM_PROBE_NO_MATH);
    // C tile through LDS in whole-line row segments (nt_epilogue_staged*) when every epilogue operand allows 16-byte accesses
    p.staged = !(flags & E2K_GEMM_NO_STAGE) && (K1 % BK) == 0 && (K2 % BK) == 0 && !(flags & E2K_GEMM_NO_GLDS) &&
               (N & 7) == 0 && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 &&
               (!bias || ((uintptr_t)bias & 15) == 0) && (!colscale || ((lds & 3) == 0 && ((uintptr_t)colscale & 15) == 0)) &&
               (!resid || ((ldr & 7) == 0 && ((uintptr_t)resid & 15) == 0));      // (the second output's operands were checked above)
    const bool glds = !(flags & E2K_GEMM_NO_GLDS) && (K1 % BK) == 0 && (K2 % BK) == 0;
    const int t256 = ((M + QBM - 1) / QBM) * ((N + QBN - 1) / QBN);
    // default: shapes with at least 64 tiles of 256 x 256.  Timed ALONE the kernel pays only where its tiles fill >= 7/8 of the
    // 256 workgroup slots (+10-22 % there, -10-15 % on half-filled shapes such as 8448 x 1024 x 4096,
    // profiles/r02_gemm_t256_first_hw_run.json) and round 2 drew the line at 224 tiles; IN the step the CUs a half-filling launch
    // leaves free are taken by the other launch lanes' kernels, and the big tile's better K loop wins: cfg3 step 88.9 -> 87.5 ms
    // on one box, 95.5 -> 92.4 on another, the same for thresholds 132 / 66 / 33 / 1 (profiles/r03_t256_threshold_ab.jsonl).
    // E2K_GEMM_T256_MIN overrides (A/B).
    const int t256_min = nt_t256_min();
    const bool q256 = glds && !p.probe && !(flags & E2K_GEMM_NO_T256) &&
                      ((flags & E2K_GEMM_T256) || (t256 >= t256_min && (K1 + K2) >= 4 * BK));
    if (q256) {
        const int T = t256;
        // tile order: workgroups go round-robin over the 8 XCDs and xcd_remap hands each XCD a contiguous run of T / 8 tiles, taken
        // from groups of `group` row tiles x all column tiles.  With the fixed 8 rows a group of a narrow output is the share of
        // TWO or more XCDs (8448 x 1024: 32 tiles against 16.5 per XCD), i.e. every A row panel is fetched by several L2s; the
        // group is sized to one XCD's share instead 
        {
            const int group_env = 0;            // (a fixed group height was an A/B switch in round 4: the per-XCD rule below won)
            const int tn256 = (N + QBN - 1) / QBN;
            const float per_xcd = T / 8.f;
            if (group_env > 0) p.group = group_env;
            else if (tn256 * 8 > 1.5f * per_xcd) {
                int g = (int)(per_xcd / tn256 + 0.5f);
                p.group = g < 1 ? 1 : (g > 8 ? 8 : g);
            }
        }
        p.full = T; p.split = 1; p.ws = ws;
        int rem = 0;
        const int slots = (flags & E2K_GEMM_TEST_SLOTS8) ? 8 : 256;
        if (ws && nt_split_on(flags) && T > slots && (T % slots) != 0) {
            rem = T % slots;
            const int nk = (K1 + K2) / BK;
            int split = 1;
            constexpr int split_cap = 16, split_mink = 4;
            while (split * 2 <= split_cap && split * 2 * rem <= slots && split * 2 * split_mink <= nk) split *= 2;   // >= 4 K tiles per part
            // same trade as below: half a round saved (~1 us per K step of a 256 x 256 tile) against 256 KB of fp32
            // partials per part written and re-read, plus the fix-up launch.  UNMEASURED constants (scaled from the 128 x 128 ones)
            if (!(flags & E2K_GEMM_TEST_SLOTS8))
                while (split > 1 && 1.0f * nk < 1.2f * (rem * split * 0.104f + 4.f)) split >>= 1;
            if (split > 1 && (int64_t)rem * split * QBM * QBN * 4 <= ws_bytes) { p.full = T - rem; p.split = split; }
            else rem = 0;
        }
        dim3 grid(p.full + rem * p.split);
        hipStream_t st = (hipStream_t)stream;
        if (out_f32) hipLaunchKernelGGL(gemm_nt_256_kernel<true>, grid, dim3(QTHREADS), 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_256_kernel<false>, grid, dim3(QTHREADS), 0, st, p);
        E2K_CHECK_LAUNCH();
        if (rem) {
            if (out_f32) hipLaunchKernelGGL(gemm_nt_256_fixup_kernel<true>, dim3(rem, 16), dim3(QTHREADS), 0, st, p);
            else hipLaunchKernelGGL(gemm_nt_256_fixup_kernel<false>, dim3(rem, 16), dim3(QTHREADS), 0, st, p);
            E2K_CHECK_LAUNCH();
        }
        return 0;
    }
    const int tm = (M + BM - 1) / BM, T = tm * tn;
    // Remainder split: `slots` workgroups are resident (256 CUs x 2); a
    // trailing partial round of `rem` tiles would run at rem/slots of the chip (8448 rows x 1024 columns = 528 tiles
    // of 128 x 128: the last 16 cost half a round), so those tiles are cut into `split` K ranges that together fill
    // the chip once more.
    p.full = T; p.split = 1; p.ws = ws;
    int rem = 0;
    const int slots = (flags & E2K_GEMM_TEST_SLOTS8) ? 8 : NT_SLOTS;
    if (glds && ws && nt_split_on(flags) && T > slots && (T % slots) != 0) {
        rem = T % slots;
        const int nk = (K1 + K2) / BK;
        int split = 1;
        while (split * 2 <= 16 && split * 2 * rem <= slots && split * 2 * 2 <= nk) split *= 2;
        // worth it only when the partial round it removes (about half a round: ~0.5 us per K step, measured) costs more
        // than writing + re-reading the fp32 partials (64 KB per 128 x 128 tile at ~5 TB/s) and the fix-up launch (~4 us)
        const float per_part = 0.026f;
        if (!(flags & E2K_GEMM_TEST_SLOTS8))
            while (split > 1 && 0.5f * nk < 1.2f * (rem * split * per_part + 4.f)) split >>= 1;
        if (split > 1 && (int64_t)rem * split * BM * BN * 4 <= ws_bytes) { p.full = T - rem; p.split = split; }
        else rem = 0;
    }
    dim3 grid(p.full + rem * p.split), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_f32) {
        if (glds) hipLaunchKernelGGL(gemm_nt_glds_kernel<true>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_kernel<true>, grid, block, 0, st, p);
    } else {
        if (glds) hipLaunchKernelGGL(gemm_nt_glds_kernel<false>, grid, block, 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_kernel<false>, grid, block, 0, st, p);
    }
    E2K_CHECK_LAUNCH();
    if (rem) {
        if (out_f32) hipLaunchKernelGGL(gemm_nt_fixup_kernel<true>, dim3(rem, 4), block, 0, st, p);
        else hipLaunchKernelGGL(gemm_nt_fixup_kernel<false>, dim3(rem, 4), block, 0, st, p);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

// FeedForward GEMM1 with the GEGLU (+ dropout) as its epilogue: always the 256 x 256 kernel (its two B half tiles are the
// value and the gate columns of the same outputs); shapes it cannot take are refused, the caller then runs
// e2k_gemm_nt_bf16 + e2k_geglu_fwd (e2k_query_gemm_nt_geglu says which)
static bool nt_geglu_ok(int M, int F, int K) { return M > 0 && F >= 128 && (F % 128) == 0 && K >= BK && (K % BK) == 0; }

static int gemm_nt_geglu_bf16_impl(const void* A, int64_t lda, int K, const void* W1, int64_t ldb, const float* bias,
                                      void* H, int64_t ldh, void* out, int64_t ldo, int M, int F, float p_drop, uint32_t seed,
                                      const uint32_t* seed_dev, uint32_t stream_id, int flags, float* ws, int64_t ws_bytes,
                                      void* stream) {
    if (M <= 0 || F <= 0) return 0;
    if (!nt_geglu_ok(M, F, K)) return E2K_ERR_SHAPE;
    if ((lda & 7) || (ldb & 7) || (ldo & 3) || (H && (ldh & 3))) return E2K_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W1) & 15) return E2K_ERR_ALIGN;
    if (out == nullptr || !(p_drop >= 0.f && p_drop < 1.f)) return E2K_ERR_ARG;
    NTArgs p{};
    p.A1 = (const bf16_t*)A; p.lda1 = lda; p.K1 = K;
    p.B = (const bf16_t*)W1; p.ldb = ldb;
    p.C = H; p.ldc = ldh; p.M = M; p.N = 2 * F; p.bias = bias;
    p.glu_out = (bf16_t*)out; p.ldg = ldo;
    p.seed = seed; p.seed_dev = seed_dev; p.stream_id = stream_id;
    p.thresh = (unsigned)(p_drop * 65536.f + 0.5f); p.inv_keep = 1.f / (1.f - p_drop);
    p.group = 8;
    const int T = ((M + QBM - 1) / QBM) * (F / 128);
    p.full = T; p.split = 1; p.ws = ws;
    int rem = 0;
    const int slots = (flags & E2K_GEMM_TEST_SLOTS8) ? 8 : 256;
    if (ws && nt_split_on(flags) && T > slots && (T % slots) != 0) {      // same remainder split as the plain kernel
        rem = T % slots;
        const int nk = K / BK;
        int split = 1;
        while (split * 2 <= 16 && split * 2 * rem <= slots && split * 2 * 4 <= nk) split *= 2;
        if (!(flags & E2K_GEMM_TEST_SLOTS8))
            while (split > 1 && 1.0f * nk < 1.2f * (rem * split * 0.104f + 4.f)) split >>= 1;
        if (split > 1 && (int64_t)rem * split * QBM * QBN * 4 <= ws_bytes) { p.full = T - rem; p.split = split; }
        else rem = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    p.staged = !(flags & E2K_GEMM_NO_STAGE) && H == nullptr && (ldo & 7) == 0 && ((uintptr_t)out & 15) == 0;      // F % 128 == 0 already
    hipLaunchKernelGGL((gemm_nt_256_kernel<false, true>), dim3(p.full + rem * p.split), dim3(QTHREADS), 0, st, p);
    E2K_CHECK_LAUNCH();
    if (rem) {
        hipLaunchKernelGGL(gemm_nt_256_fixup_glu_kernel, dim3(rem, 8), dim3(QTHREADS), 0, st, p);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int e2k_query_gemm_nt_geglu(int M, int F, int K) { return nt_geglu_ok(M, F, K) ? 1 : 0; }

// FeedForward's second Linear dgrad with the GEGLU backward as its epilogue: d(act) = dY W2 never goes to memory.  The 256 x 256
// kernel only (same tile order and remainder split as e2k_gemm_nt_bf16); shapes it cannot take are refused, the caller then runs
// e2k_gemm_nt_bf16 + e2k_geglu_bwd (e2k_query_gemm_nt_geglu_bwd says which)
static bool nt_geglu_bwd_ok(int M, int F, int K) { return M > 0 && F >= QBN && (F % QBN) == 0 && K >= 4 * BK && (K % BK) == 0; }

static int gemm_nt_geglu_bwd_bf16_impl(const void* dY, int64_t ldy, int K, const void* W2T, int64_t ldb, const void* H, int64_t ldh,
                                          void* dH, int64_t lddh, int M, int F, float p_drop, uint32_t seed, const uint32_t* seed_dev,
                                          uint32_t stream_id, int flags, float* ws, int64_t ws_bytes, void* stream) {
    if (M <= 0 || F <= 0) return 0;
    if (!nt_geglu_bwd_ok(M, F, K)) return E2K_ERR_SHAPE;
    if ((ldy & 7) || (ldb & 7) || (ldh & 7) || (lddh & 7)) return E2K_ERR_ALIGN;
    if (((uintptr_t)dY | (uintptr_t)W2T | (uintptr_t)H | (uintptr_t)dH) & 15) return E2K_ERR_ALIGN;
    if (!H || !dH || !(p_drop >= 0.f && p_drop < 1.f)) return E2K_ERR_ARG;
    NTArgs p{};
    p.A1 = (const bf16_t*)dY; p.lda1 = ldy; p.K1 = K;
    p.B = (const bf16_t*)W2T; p.ldb = ldb;
    p.C = dH; p.ldc = lddh; p.M = M; p.N = F;              // (C is never written in this mode)
    p.gb_H = (const bf16_t*)H; p.gb_ldh = ldh; p.gb_dH = (bf16_t*)dH; p.gb_lddh = lddh;
    p.seed = seed; p.seed_dev = seed_dev; p.stream_id = stream_id;
    p.thresh = (unsigned)(p_drop * 65536.f + 0.5f); p.inv_keep = 1.f / (1.f - p_drop);
    p.staged = !(flags & E2K_GEMM_NO_STAGE);
    const int T = ((M + QBM - 1) / QBM) * (F / QBN);
    {
        const int tn256 = F / QBN;
        const float per_xcd = T / 8.f;
        p.group = 8;
        if (tn256 * 8 > 1.5f * per_xcd) {
            int g = (int)(per_xcd / tn256 + 0.5f);
            p.group = g < 1 ? 1 : (g > 8 ? 8 : g);
        }
    }
    p.full = T; p.split = 1; p.ws = ws;
    int rem = 0;
    const int slots = (flags & E2K_GEMM_TEST_SLOTS8) ? 8 : 256;
    if (ws && nt_split_on(flags) && T > slots && (T % slots) != 0) {      // same remainder split as the plain kernel
        rem = T % slots;
        const int nk = K / BK;
        int split = 1;
        while (split * 2 <= 16 && split * 2 * rem <= slots && split * 2 * 4 <= nk) split *= 2;
        if (!(flags & E2K_GEMM_TEST_SLOTS8))
            while (split > 1 && 1.0f * nk < 1.2f * (rem * split * 0.104f + 4.f)) split >>= 1;
        if (split > 1 && (int64_t)rem * split * QBM * QBN * 4 <= ws_bytes) { p.full = T - rem; p.split = split; }
        else rem = 0;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((gemm_nt_256_kernel<false, false, 1>), dim3(p.full + rem * p.split), dim3(QTHREADS), 0, st, p);
    E2K_CHECK_LAUNCH();
    if (rem) {
        hipLaunchKernelGGL((gemm_nt_256_fixup_kernel<false, true>), dim3(rem, 16), dim3(QTHREADS), 0, st, p);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

// 1: the fused launch is what the caller should use; 2: it can run (e2k_gemm_nt_geglu_bwd_bf16 accepts the shape) but the output has
// fewer 256 x 256 tiles than e2k_gemm_nt_bf16 itself asks for before it takes that kernel -- a 256-row tile of a small M is mostly
// padding, and the 128 x 128 kernel + e2k_geglu_bwd is the better pair there; 0: refused
extern "C" int e2k_query_gemm_nt_geglu_bwd(int M, int F, int K) {
    if (!nt_geglu_bwd_ok(M, F, K)) return 0;
    const int t256 = ((M + QBM - 1) / QBM) * (F / QBN);
    return t256 >= nt_t256_min() ? 1 : 2;
}

extern "C" int e2k_gemm_nt_geglu_bwd_bf16(const void* dY, int64_t ldy, int K, const void* W2T, int64_t ldb, const void* H, int64_t ldh,
                                          void* dH, int64_t lddh, int M, int F, float p_drop, uint32_t seed, const uint32_t* seed_dev,
                                          uint32_t stream_id, int flags, float* ws, int64_t ws_bytes, void* stream) {
    return e2k::dispatch("gemm_nt_geglu_bwd_bf16", gemm_nt_geglu_bwd_bf16_impl, dY, ldy, K, W2T, ldb, H, ldh, dH, lddh, M, F, p_drop, seed, seed_dev,
                         stream_id, flags, ws, ws_bytes, stream);
}


// Attention's input projection with the rotary embedding of q and k as its epilogue: columns [0, 2 H 64) of
// x W^T leave the kernel rotated and head-major, (B, H, Ntok, 64) at Q / Kh -- the q / k half of e2k_qkv_post_fwd, which then runs with
// Q = K = NULL and reads only the value columns and the gate columns of C.  The 256 x 256 kernel with the staged epilogue only; shapes
// the kernel cannot take are refused (e2k_query_gemm_nt_qkrot says which, and which it can take but should not), the caller then runs the
// two-launch form.  Bit-identical to it: the product is rounded to bf16 before the rotation, as if it had been stored.  Never splits a remainder.
static bool nt_qkrot_ok(int M, int N, int K, int H) {
    return M > 0 && H > 0 && !(H & 1) && N >= 3 * H * DH_ROT && !(N & 7) && K >= 4 * BK && (K % BK) == 0;
}

static int gemm_nt_qkrot_bf16_impl(const void* A, int64_t lda, int K, const void* W, int64_t ldb, const float* bias, void* C, int64_t ldc,
                                   void* Q, void* Kh, const float* cosb, const float* sinb, int B, int H, int Ntok, int N, void* stream) {
    if (B <= 0 || Ntok <= 0 || N <= 0) return 0;
    const int M = B * Ntok;
    if (!nt_qkrot_ok(M, N, K, H)) return E2K_ERR_SHAPE;
    if (!A || !W || !C || !Q || !Kh || !cosb || !sinb) return E2K_ERR_ARG;
    if ((lda & 7) || (ldb & 7) || (ldc & 7)) return E2K_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)Q | (uintptr_t)Kh | (uintptr_t)cosb | (uintptr_t)sinb | (uintptr_t)bias) & 15)
        return E2K_ERR_ALIGN;
    NTArgs p{};
    p.A1 = (const bf16_t*)A; p.lda1 = lda; p.K1 = K;
    p.B = (const bf16_t*)W; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.bias = bias;
    p.rot_Q = (bf16_t*)Q; p.rot_K = (bf16_t*)Kh; p.rot_cos = cosb; p.rot_sin = sinb;
    p.rot_I = H * DH_ROT; p.rot_2I = 2 * H * DH_ROT; p.rot_N = Ntok; p.rot_H = H;
    p.staged = 1;
    const int tn256 = (N + QBN - 1) / QBN, T = ((M + QBM - 1) / QBM) * tn256;
    {
        const float per_xcd = T / 8.f;          // (tile order: the rule of e2k_gemm_nt_bf16)
        p.group = 8;
        if (tn256 * 8 > 1.5f * per_xcd) {
            int g = (int)(per_xcd / tn256 + 0.5f);
            p.group = g < 1 ? 1 : (g > 8 ? 8 : g);
        }
    }
    p.full = T; p.split = 1;
    hipLaunchKernelGGL((gemm_nt_256_kernel<false, false, 2>), dim3(T), dim3(QTHREADS), 0, (hipStream_t)stream, p);
    E2K_CHECK_LAUNCH();
    return 0;
}

// 1: the fused launch is what the caller should use; 2: it can run, but the output has fewer 256 x 256 tiles than e2k_gemm_nt_bf16 asks
// for before it takes that kernel (the 128 x 128 kernel + the full e2k_qkv_post_fwd is the better pair there); 0: refused
extern "C" int e2k_query_gemm_nt_qkrot(int M, int N, int K, int H) {
    if (!nt_qkrot_ok(M, N, K, H)) return 0;
    const int t256 = ((M + QBM - 1) / QBM) * ((N + QBN - 1) / QBN);
    return t256 >= nt_t256_min() ? 1 : 2;
}

extern "C" int e2k_gemm_nt_qkrot_bf16(const void* A, int64_t lda, int K, const void* W, int64_t ldb, const float* bias, void* C, int64_t ldc,
                                      void* Q, void* Kh, const float* cosb, const float* sinb, int B, int H, int Ntok, int N, void* stream) {
    return e2k::dispatch("gemm_nt_qkrot_bf16", gemm_nt_qkrot_bf16_impl, A, lda, K, W, ldb, bias, C, ldc, Q, Kh, cosb, sinb, B, H, Ntok, N, stream);
}

// 512 partial slots of a 128 x 128 tile or 256 of a 256 x 256 tile (every remainder split fits: rem * split <= slots)
extern "C" int e2k_query_gemm_nt_ws_bytes(void) { return 256 * QBM * QBN * 4; }

// upper bound over both weight-gradient kernels (use_tr = 1 lets the library choose between them): a caller that sizes
// ws = splits * N * K floats from this number is safe whichever kernel e2k_gemm_tn_bf16 selects
extern "C" int e2k_query_gemm_tn_splits(int M, int N, int K, int splits) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const int a = tn_splits(M, N, K, splits);
    const int b = tn_use_256(M, N, K, 3) ? tn_splits_256(M, N, K, splits) : 1;
    return a > b ? a : b;
}

extern "C" int e2k_query_gemm_tn_splits_mode(int M, int N, int K, int splits, int use_tr) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    return tn_use_256(M, N, K, use_tr) ? tn_splits_256(M, N, K, splits) : tn_splits(M, N, K, splits);
}

// floats of workspace e2k_gemm_tn_bf16 needs for (M, N, K, splits, use_tr): splits x whole tiles of the selected kernel
// (partial tiles are stored in fragment order, padded to whole 128 x 128 / 256 x 256 tiles); 0 when nothing is split
extern "C" int64_t e2k_query_gemm_tn_ws_floats(int M, int N, int K, int splits, int use_tr) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const bool big = tn_use_256(M, N, K, use_tr);
    const int ns = big ? tn_splits_256(M, N, K, splits) : tn_splits(M, N, K, splits);
    if (ns <= 1) return 0;
    const int ts = big ? T2 : 128;
    return (int64_t)ns * ((N + ts - 1) / ts) * ((K + ts - 1) / ts) * ts * ts;
}

extern "C" int e2k_colsum_bf16(const void* x, int64_t ldx, float* out, int M, int N, void* stream);

static int gemm_tn_bf16_impl(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                int M, int N, int K, int splits, int use_tr, float* ws, float* colsum, int cs_from,
                                const void* A2, int64_t lda2, int N1, const void* B2, int64_t ldb2, int K1,
                                void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (A2 || B2) {                  // dual-source: 256 x 256 kernel only, block boundaries on tile boundaries
        if ((M % TBM) != 0 || colsum) return E2K_ERR_SHAPE;
        if (A2 && (N1 <= 0 || N1 >= N || (N1 % T2) != 0 || (lda2 & 7) || ((uintptr_t)A2 & 15) || ((N - N1 + 7) & ~7) > lda2)) return E2K_ERR_SHAPE;
        if (B2 && (K1 <= 0 || K1 >= K || (K1 % T2) != 0 || (ldb2 & 7) || ((uintptr_t)B2 & 15) || ((K - K1 + 7) & ~7) > ldb2)) return E2K_ERR_SHAPE;
        if ((lda & 7) || (ldb & 7) || (((uintptr_t)A | (uintptr_t)B) & 15)) return E2K_ERR_ALIGN;
        if (((A2 ? N1 : N) + 7 & ~7) > lda || ((B2 ? K1 : K) + 7 & ~7) > ldb) return E2K_ERR_ALIGN;
        use_tr = 3;
    } else
    {   // 16-B loads may run past N / K up to the next multiple of 8: that must still be inside the row
        if ((lda & 7) || (ldb & 7) || ((N + 7) & ~7) > lda || ((K + 7) & ~7) > ldb) return E2K_ERR_ALIGN;
        if (((uintptr_t)A | (uintptr_t)B) & 15) return E2K_ERR_ALIGN;
    }
    const bool big = tn_use_256(M, N, K, use_tr);
    const int tsz = big ? T2 : 128;
    const int tn = (N + tsz - 1) / tsz, tk = (K + tsz - 1) / tsz;
    splits = big ? tn_splits_256(M, N, K, splits) : tn_splits(M, N, K, splits);
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TBM - 1) / TBM * TBM;
    if (splits > 1 && ws == nullptr) return E2K_ERR_ARG;
    TNArgs p;
    p.ws = ws;
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.splits = splits; p.chunk = chunk;
    p.A2 = (const bf16_t*)A2; p.lda2 = lda2; p.N1 = N1; p.B2 = (const bf16_t*)B2; p.ldb2 = ldb2; p.K1 = K1;
    const bool fast = use_tr && (M % TBM) == 0 && N >= 8 && K >= 8;
    if (colsum && (cs_from < 0 || cs_from >= N || (cs_from & 1))) return E2K_ERR_ARG;
    p.colsum = fast ? colsum : nullptr; p.cs_from = cs_from;
    if (colsum && !fast) {          // the general kernels do not carry the column sums: separate pass over dY
        int rc = e2k_colsum_bf16((const bf16_t*)A + cs_from, lda, colsum + cs_from, M, N - cs_from, stream);
        if (rc) return rc;
    }
    dim3 grid(tn * tk, splits), block(256);
    if (big && p.colsum) hipLaunchKernelGGL(gemm_tn_256_kernel<true>, grid, dim3(T2THREADS), 0, (hipStream_t)stream, p);
    else if (big) hipLaunchKernelGGL(gemm_tn_256_kernel<false>, grid, dim3(T2THREADS), 0, (hipStream_t)stream, p);
    else if (fast && p.colsum) hipLaunchKernelGGL(gemm_tn_glds_kernel<true>, grid, block, 0, (hipStream_t)stream, p);
    else if (fast) hipLaunchKernelGGL(gemm_tn_glds_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
    else if (use_tr) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, block, 0, (hipStream_t)stream, p);
    E2K_CHECK_LAUNCH();
    if (splits > 1) {
        if (big) hipLaunchKernelGGL(tn_reduce_frag_kernel<true>, dim3(tn * tk, 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)ws, C, (long)ldc, N, K, splits, tk, TNReduceGroup{});
        else hipLaunchKernelGGL(tn_reduce_frag_kernel<false>, dim3(tn * tk, 8), dim3(128), 0, (hipStream_t)stream, (const f32x4*)ws, C, (long)ldc, N, K, splits, tk, TNReduceGroup{});
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

// token splits of a grouped launch: all problems share M and the split count; pick the count with the fewest rounds x steps
static int tn_group_splits(int M, int tiles, int splits) {
    if (splits <= 0) {
        const int maxs = M / (TBM * 8) < 1 ? 1 : (M / (TBM * 8) > 16 ? 16 : M / (TBM * 8));
        long best = -1;
        splits = 1;
        for (int sp = 1; sp <= maxs; ++sp) {
            const long rounds = ((long)tiles * sp + 255) / 256;
            const long steps = ((M + sp - 1) / sp + TBM - 1) / TBM;
            const long cost = rounds * (steps + 6) + (sp > 1 ? 2 : 0);        // (+6: prologue + partial-tile store of a workgroup, in steps)
            if (best < 0 || cost < best) { best = cost; splits = sp; }
        }
    }
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TBM - 1) / TBM * TBM;
    return (M + chunk - 1) / chunk;
}

struct TNGroupHost { e2k_tn_problem p[TN_GROUP_MAX]; int n; };

static int tn_group_tiles(const TNGroupHost& h) {
    int t = 0;
    for (int i = 0; i < h.n; ++i) t += ((h.p[i].N + T2 - 1) / T2) * ((h.p[i].K + T2 - 1) / T2);
    return t;
}

static int gemm_tn_group_impl(TNGroupHost h, int M, int splits, float* ws, void* stream) {
    if (h.n <= 0 || M <= 0) return 0;
    if (h.n > TN_GROUP_MAX || (M % TBM) != 0) return E2K_ERR_SHAPE;
    TNGroupArgs g;
    TNReduceGroup r;
    bool any_cs = false;
    int tiles = 0;
    for (int i = 0; i < h.n; ++i) {
        const e2k_tn_problem& q = h.p[i];
        if (q.N < 8 || q.K < 8 || !q.A || !q.B || !q.C) return E2K_ERR_ARG;
        if ((q.lda & 7) || (q.ldb & 7) || ((q.N + 7) & ~7) > q.lda || ((q.K + 7) & ~7) > q.ldb) return E2K_ERR_ALIGN;
        if (((uintptr_t)q.A | (uintptr_t)q.B) & 15) return E2K_ERR_ALIGN;
        if (q.colsum && (q.cs_from < 0 || q.cs_from >= q.N || (q.cs_from & 1))) return E2K_ERR_ARG;
        TNProb& d = g.prob[i];
        d.A = (const bf16_t*)q.A; d.lda = q.lda; d.B = (const bf16_t*)q.B; d.ldb = q.ldb; d.C = q.C; d.ldc = q.ldc;
        d.N = q.N; d.K = q.K; d.tile0 = tiles; d.colsum = q.colsum; d.cs_from = q.cs_from;
        r.prob[i] = d;
        any_cs = any_cs || q.colsum != nullptr;
        tiles += ((q.N + T2 - 1) / T2) * ((q.K + T2 - 1) / T2);
    }
    for (int i = h.n; i < TN_GROUP_MAX; ++i) { g.prob[i] = g.prob[0]; g.prob[i].tile0 = 1 << 30; r.prob[i] = g.prob[i]; }
    splits = tn_group_splits(M, tiles, splits);
    if (splits > 1 && ws == nullptr) return E2K_ERR_ARG;
    int chunk = (M + splits - 1) / splits;
    chunk = (chunk + TBM - 1) / TBM * TBM;
    g.n = h.n; g.M = M; g.splits = splits; g.chunk = chunk; g.tiles = tiles; g.ws = ws;
    r.n = h.n;
    dim3 grid(tiles, splits);
    if (any_cs) hipLaunchKernelGGL((gemm_tn_256_kernel<true, TNGroupArgs>), grid, dim3(T2THREADS), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((gemm_tn_256_kernel<false, TNGroupArgs>), grid, dim3(T2THREADS), 0, (hipStream_t)stream, g);
    E2K_CHECK_LAUNCH();
    if (splits > 1) {
        hipLaunchKernelGGL((tn_reduce_frag_kernel<true, true>), dim3(tiles, 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)ws,
                           (float*)nullptr, 0l, 0, 0, splits, 1, r);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int e2k_gemm_tn_group_bf16(const e2k_tn_problem* problems, int n, int M, int splits, float* ws, void* stream) {
    if (n < 0 || n > TN_GROUP_MAX || (n > 0 && problems == nullptr)) return E2K_ERR_ARG;
    TNGroupHost h;
    h.n = n;
    for (int i = 0; i < n; ++i) h.p[i] = problems[i];
    for (int i = n; i < TN_GROUP_MAX; ++i) h.p[i] = e2k_tn_problem{};
    return e2k::dispatch("gemm_tn_group_bf16", gemm_tn_group_impl, h, M, splits, ws, stream);
}

extern "C" int64_t e2k_query_gemm_tn_group_ws_floats(const e2k_tn_problem* problems, int n, int M, int splits) {
    if (n <= 0 || n > TN_GROUP_MAX || problems == nullptr || M <= 0) return 0;
    TNGroupHost h;
    h.n = n;
    for (int i = 0; i < n; ++i) h.p[i] = problems[i];
    const int tiles = tn_group_tiles(h);
    const int ns = tn_group_splits(M, tiles, splits);
    return ns <= 1 ? 0 : (int64_t)ns * tiles * T2 * T2;
}

// ---- C ABI: every compute entry point goes through e2k::dispatch (plan.h) so that a launch plan can record it

extern "C" int e2k_gemm_nt_bf16(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                                const void* B, int64_t ldb, void* C, int64_t ldc, int out_f32, int accumulate,
                                int M, int N, const float* bias, const float* colscale, int64_t lds,
                                int rows_per_batch, const uint8_t* rowmask, const void* resid, int64_t ldr, int flags,
                                float* ws, int64_t ws_bytes, void* stream) {
    return e2k::dispatch("gemm_nt_bf16", gemm_nt_bf16_impl, A1, lda1, K1, A2, lda2, K2, B, ldb, C, ldc, out_f32, accumulate, M, N, bias, colscale, lds, rows_per_batch, rowmask, resid, ldr, flags, ws, ws_bytes, 0, (void*)nullptr, (int64_t)0, (const void*)nullptr, (int64_t)0, stream);
}

extern "C" int e2k_gemm_nt2_bf16(const void* A1, int64_t lda1, int K1, const void* A2, int64_t lda2, int K2,
                                 const void* B, int64_t ldb, int M, int N, int nsplit, void* C, int64_t ldc, void* C2, int64_t ldc2,
                                 const void* resid, int64_t ldr, const void* resid2, int64_t ldr2, int flags,
                                 float* ws, int64_t ws_bytes, void* stream) {
    if (nsplit <= 0) return E2K_ERR_ARG;
    return e2k::dispatch("gemm_nt_bf16", gemm_nt_bf16_impl, A1, lda1, K1, A2, lda2, K2, B, ldb, C, ldc, 0, 0, M, N, (const float*)nullptr, (const float*)nullptr, (int64_t)0, 0, (const uint8_t*)nullptr, resid, ldr, flags, ws, ws_bytes, nsplit, C2, ldc2, resid2, ldr2, stream);
}

extern "C" int e2k_gemm_nt_geglu_bf16(const void* A, int64_t lda, int K, const void* W1, int64_t ldb, const float* bias,
                                      void* H, int64_t ldh, void* out, int64_t ldo, int M, int F, float p_drop, uint32_t seed,
                                      const uint32_t* seed_dev, uint32_t stream_id, int flags, float* ws, int64_t ws_bytes,
                                      void* stream) {
    return e2k::dispatch("gemm_nt_geglu_bf16", gemm_nt_geglu_bf16_impl, A, lda, K, W1, ldb, bias, H, ldh, out, ldo, M, F, p_drop, seed, seed_dev, stream_id, flags, ws, ws_bytes, stream);
}

extern "C" int e2k_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                                int M, int N, int K, int splits, int use_tr, float* ws, float* colsum, int cs_from,
                                void* stream) {
    return e2k::dispatch("gemm_tn_bf16", gemm_tn_bf16_impl, A, lda, B, ldb, C, ldc, M, N, K, splits, use_tr, ws, colsum, cs_from,
                         (const void*)nullptr, (int64_t)0, 0, (const void*)nullptr, (int64_t)0, 0, stream);
}

extern "C" int e2k_gemm_tn_dual_bf16(const void* A1, int64_t lda1, int N1, const void* A2, int64_t lda2, int N2,
                                     const void* B1, int64_t ldb1, int K1, const void* B2, int64_t ldb2, int K2,
                                     float* C, int64_t ldc, int M, int splits, float* ws, void* stream) {
    if ((A2 == nullptr) != (N2 == 0) || (B2 == nullptr) != (K2 == 0) || N2 < 0 || K2 < 0) return E2K_ERR_ARG;
    return e2k::dispatch("gemm_tn_dual_bf16", gemm_tn_bf16_impl, A1, lda1, B1, ldb1, C, ldc, M, N1 + N2, K1 + K2, splits, 3, ws,
                         (float*)nullptr, 0, A2, lda2, N1, B2, ldb2, K1, stream);
}
