// Hyper-connections (4 residual streams), fused depth + width connection, forward and backward.
//
// Replaces hyper_connections.HyperConnections as the reference uses it
//     x, add_residual = hc(x) ... x = add_residual(branch_out)          (e2_tts.py:870-882, 900-939)
// (arithmetic restated in SURVEY.md Appendix A.5 / oracle HyperConnections).
//
// Data layout: the reference keeps streams in the batch dim '(b s) n d'; here a token's 4 streams are
// contiguous: X[token][s][d] (bf16), one 4*D*2-byte burst per token.
//
// For HC instance i on stream tensor X_i:
//   width :  r = X_i ;  z_s = r_s/|r_s| * sqrt(D) * (gamma+1)
//            a[s][t] = tanh(z_s . Wa[:,t]) * sa + A[s][t]      (t = 0..4)
//            b[s]    = tanh(z_s . wb) * sb + B[s]
//            mix_t   = sum_s a[s][t] r_s ;  branch_in = mix_0 ;  M_i[s] = mix_{s+1}
//   depth :  X_{i+1}[s] = M_i[s] + b[s] * y_i         (y_i = branch output)
// The forward kernel fuses depth_{i-1} with width_i (X_i is never written: HBM traffic 5D in + 5D out per
// token instead of 9D + 9D); the backward kernel fuses width_i-backward with depth_{i-1}-backward and
// recomputes r from (M_{i-1}, y_{i-1}, b_{i-1}).  Per-token coefficients are saved in `coef` (52 floats).
// HBM-bound: algorithmic bytes per token = 2*(5D+5D) forward, 2*(11D+5D) backward.
//
// Work split: NW waves share a token (each owns a contiguous D/NW slice, 8 elements per lane per row at D = 1024
// and 512), so the per-lane state is small enough for 2-3 waves per SIMD; the 24-28 dot products of a token are
// reduced with DPP row operations inside a wave and through a small LDS exchange across the NW waves.  The
// backward accumulates d(Wp) in registers over all tokens of a wave and flushes it once.
#include "e2k_device.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"

using namespace e2k;

namespace {

constexpr int S = 4, NJ = 6, CW = 52;
constexpr int CA = 0, CB = 20, CP = 24, CRN = 48;   // a[s*5+t], b[s], P[s*6+j] (pre-tanh dots), rn[s]
constexpr int NSC = 32;                               // scalar partials: dA[20], dB[4], dsa, dsb, pad
constexpr int NRED = 28;

struct HCParams {
    const float* static_beta; const float* static_alpha; const float* dyn_alpha_fn; const float* dyn_alpha_scale;
    const float* dyn_beta_fn; const float* dyn_beta_scale; const float* gamma;
};

// Wp[j][d] = (gamma[d]+1) * W[d][j]   (j < 5: dynamic_alpha_fn column, j = 5: dynamic_beta_fn)
__device__ __forceinline__ void stage_wp(float* Wp, const HCParams& hp, int D, int tid) {
    for (int d = tid; d < D; d += 256) {
        float g = hp.gamma[d] + 1.f;
#pragma unroll
        for (int t = 0; t < 5; ++t) Wp[t * D + d] = g * hp.dyn_alpha_fn[d * 5 + t];
        Wp[5 * D + d] = g * hp.dyn_beta_fn[d];
    }
}

// sum `vals[0..N)` over the NW waves that share a token: wave-level DPP sums, then LDS exchange (one barrier)
template <int N, int NW>
__device__ __forceinline__ void token_sum(float* vals, float (*red)[4][NRED], int parity, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < N; ++i) vals[i] = wave_sum_fast(vals[i]);
    if (NW == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[parity][wave][i] = vals[i];
    }
    __syncthreads();
    const int w0 = (wave / NW) * NW;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = red[parity][w0][i];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += red[parity][w0 + w][i];
        vals[i] = uniform_f(s);
    }
}

struct HCFwdArgs {
    const bf16_t* Xin; const bf16_t* yprev; const float* coef_prev;
    bf16_t* Mout; bf16_t* bin; float* coef;
    HCParams hp;
    int Mtok;
};

template <int VEC, int NCH, int NW, bool DEPTH, bool WIDTH>
__global__ __launch_bounds__(256) void hc_fwd_kernel(HCFwdArgs p) {
    constexpr int EPL = VEC * NCH, DS = 64 * EPL, D = DS * NW, TPB = 4 / NW;
    __shared__ __attribute__((aligned(16))) float Wp[WIDTH ? NJ * D : 4];
    __shared__ float red[2][4][NRED];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int slot = wave / NW, doff = (wave % NW) * DS;
    float sa = 0.f, sb = 0.f, A[20], B[4];
    if (WIDTH) {
        stage_wp(Wp, p.hp, D, tid);
        sa = p.hp.dyn_alpha_scale[0];
        sb = p.hp.dyn_beta_scale[0];
#pragma unroll
        for (int i = 0; i < 20; ++i) A[i] = p.hp.static_alpha[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = p.hp.static_beta[i];
        __syncthreads();
    }
    const float sqrtD = sqrtf((float)D);
    const int per_iter = gridDim.x * TPB;
    const int niter = (p.Mtok + per_iter - 1) / per_iter;
    for (int it = 0; it < niter; ++it) {
        const int tok_raw = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const bool valid = tok_raw < p.Mtok;
        const long tok = valid ? tok_raw : p.Mtok - 1;
        float r[S][EPL];
#pragma unroll
        for (int s = 0; s < S; ++s) load_row<VEC, NCH>(p.Xin + (tok * S + s) * D + doff, lane, r[s]);
        if (DEPTH) {
            float y[EPL];
            load_row<VEC, NCH>(p.yprev + tok * D + doff, lane, y);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float bp = p.coef_prev[tok * CW + CB + s];
#pragma unroll
                for (int e = 0; e < EPL; ++e) r[s][e] = fmaf(bp, y[e], r[s][e]);
            }
        }
        if (!WIDTH) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < S; ++s) store_row<VEC, NCH>(p.Mout + (tok * S + s) * D + doff, lane, r[s]);
            }
            continue;
        }
        float part[NRED];       // [s*7 + j], j = 6: sum of squares
#pragma unroll
        for (int i = 0; i < NRED; ++i) part[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float w[EPL];
            load_row_f32<VEC, NCH>(Wp + j * D + doff, lane, w);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < EPL; ++e) part[s * 7 + j] = fmaf(r[s][e], w[e], part[s * 7 + j]);
        }
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int e = 0; e < EPL; ++e) part[s * 7 + 6] = fmaf(r[s][e], r[s][e], part[s * 7 + 6]);
        token_sum<NRED, NW>(part, red, it & 1, wave, lane);
        float a[S][5], b[S], P[S][NJ], rn[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rn[s] = 1.f / fmaxf(sqrtf(part[s * 7 + 6]), 1e-12f);
#pragma unroll
            for (int j = 0; j < NJ; ++j) P[s][j] = part[s * 7 + j] * rn[s] * sqrtD;
#pragma unroll
            for (int t = 0; t < 5; ++t) a[s][t] = tanhf_(P[s][t]) * sa + A[s * 5 + t];
            b[s] = tanhf_(P[s][5]) * sb + B[s];
        }
        if (valid) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                float m[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    float v = a[0][t] * r[0][e];
#pragma unroll
                    for (int s = 1; s < S; ++s) v = fmaf(a[s][t], r[s][e], v);
                    m[e] = v;
                }
                if (t == 0) store_row<VEC, NCH>(p.bin + tok * D + doff, lane, m);
                else store_row<VEC, NCH>(p.Mout + (tok * S + (t - 1)) * D + doff, lane, m);
            }
            if (lane == 0 && doff == 0) {
                float* c = p.coef + tok * CW;
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) c[CA + s * 5 + t] = a[s][t];
                    c[CB + s] = b[s];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) c[CP + s * NJ + j] = P[s][j];
                    c[CRN + s] = rn[s];
                }
            }
        }
    }
}

struct HCBwdArgs {
    const bf16_t* Xin; const bf16_t* yprev; const float* coef_prev;
    const bf16_t* G; const bf16_t* dbin; const bf16_t* ycur; const float* coef;
    bf16_t* dR; bf16_t* dyprev;
    HCParams hp;
    float* partial;      // [gridDim.x][NJ*D + 32]
    int Mtok;
};

template <int VEC, int NCH, int NW, bool DEPTH, bool WIDTH>
__global__ __launch_bounds__(256) void hc_bwd_kernel(HCBwdArgs p) {
    constexpr int EPL = VEC * NCH, DS = 64 * EPL, D = DS * NW, TPB = 4 / NW;
    __shared__ __attribute__((aligned(16))) float Wp[WIDTH ? NJ * D : 4];
    __shared__ __attribute__((aligned(16))) float dWp[WIDTH ? NJ * D + NSC : 4];
    __shared__ float red[2][4][NRED];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int slot = wave / NW, doff = (wave % NW) * DS;
    float sa = 0.f, sb = 0.f;
    if (WIDTH) {
        stage_wp(Wp, p.hp, D, tid);
        for (int i = tid; i < NJ * D + NSC; i += 256) dWp[i] = 0.f;
        sa = p.hp.dyn_alpha_scale[0];
        sb = p.hp.dyn_beta_scale[0];
        __syncthreads();
    }
    const float sqrtD = sqrtf((float)D);
    float gw[WIDTH ? NJ : 1][EPL];   // d(Wp) accumulators of this lane's elements, over all tokens of this wave
#pragma unroll
    for (int j = 0; j < (WIDTH ? NJ : 1); ++j)
#pragma unroll
        for (int e = 0; e < EPL; ++e) gw[j][e] = 0.f;

    const int per_iter = gridDim.x * TPB;
    const int niter = (p.Mtok + per_iter - 1) / per_iter;
    for (int it = 0; it < niter; ++it) {
        const int tok_raw = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const bool valid = tok_raw < p.Mtok;
        const long tok = valid ? tok_raw : p.Mtok - 1;
        const float vf = valid ? 1.f : 0.f;
        float bp[S];
        if (DEPTH) {
#pragma unroll
            for (int s = 0; s < S; ++s) bp[s] = p.coef_prev[tok * CW + CB + s];
        }
        if (!WIDTH) {
            // only the depth connection of the previous instance: dy_prev = sum_s b_prev[s] * dX[s]
            float dy[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) dy[e] = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float g[EPL];
                load_row<VEC, NCH>(p.G + (tok * S + s) * D + doff, lane, g);
#pragma unroll
                for (int e = 0; e < EPL; ++e) dy[e] = fmaf(bp[s], g[e], dy[e]);
            }
            if (valid) store_row<VEC, NCH>(p.dyprev + tok * D + doff, lane, dy);
            continue;
        }
        float r[S][EPL], dm[5][EPL], yc[EPL];
#pragma unroll
        for (int s = 0; s < S; ++s) load_row<VEC, NCH>(p.Xin + (tok * S + s) * D + doff, lane, r[s]);
        if (DEPTH) {
            float y[EPL];
            load_row<VEC, NCH>(p.yprev + tok * D + doff, lane, y);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < EPL; ++e) r[s][e] = fmaf(bp[s], y[e], r[s][e]);
        }
        load_row<VEC, NCH>(p.dbin + tok * D + doff, lane, dm[0]);
#pragma unroll
        for (int s = 0; s < S; ++s) load_row<VEC, NCH>(p.G + (tok * S + s) * D + doff, lane, dm[s + 1]);
        load_row<VEC, NCH>(p.ycur + tok * D + doff, lane, yc);

        // 24 dots: da[s][t] = dm_t . r_s (index s*6 + t) ; db[s] = G_s . y_cur (index s*6 + 5)
        float dots[24];
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) v = fmaf(dm[t][e], r[s][e], v);
                dots[s * 6 + t] = v;
            }
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < EPL; ++e) v = fmaf(dm[s + 1][e], yc[e], v);
            dots[s * 6 + 5] = v;
        }
        token_sum<24, NW>(dots, red, it & 1, wave, lane);

        const float* cf = p.coef + tok * CW;
        float a[S][5], c[S][NJ], rn[S], uq[S];
        const bool sc_lane = valid && doff == 0 && lane == 0;      // scalar partials: one lane per token
        float dsa = 0.f, dsb = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rn[s] = cf[CRN + s];
            float acc_uq = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                a[s][t] = cf[CA + s * 5 + t];
                const float P = cf[CP + s * NJ + t];
                const float th = tanhf_(P);
                const float da = dots[s * 6 + t];
                c[s][t] = da * sa * (1.f - th * th);
                acc_uq = fmaf(c[s][t], P, acc_uq);
                if (sc_lane) atomicAdd(&dWp[NJ * D + s * 5 + t], da);
                dsa = fmaf(da, th, dsa);
            }
            const float P = cf[CP + s * NJ + 5];
            const float th = tanhf_(P);
            const float db = dots[s * 6 + 5];
            c[s][5] = db * sb * (1.f - th * th);
            acc_uq = fmaf(c[s][5], P, acc_uq);
            if (sc_lane) atomicAdd(&dWp[NJ * D + 20 + s], db);
            dsb = fmaf(db, th, dsb);
            uq[s] = acc_uq;
        }
        if (sc_lane) { atomicAdd(&dWp[NJ * D + 24], dsa); atomicAdd(&dWp[NJ * D + 25], dsb); }
        // q[s][e] = sqrtD * sum_j c[s][j] Wp[j][d] ; gw[j][e] += sum_s c[s][j] rn[s] sqrtD r[s][e]
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float cs = c[s][j] * sqrtD * rn[s] * vf;
#pragma unroll
                for (int e = 0; e < EPL; ++e) gw[j][e] = fmaf(cs, r[s][e], gw[j][e]);
            }
        }
        float dyp[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) dyp[e] = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // (Wp rows are re-read from LDS per stream: keeps the live register set small -> more waves per SIMD)
            float qs[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) qs[e] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float w[EPL];
                load_row_f32<VEC, NCH>(Wp + j * D + doff, lane, w);
                const float cq = c[s][j] * sqrtD;
#pragma unroll
                for (int e = 0; e < EPL; ++e) qs[e] = fmaf(cq, w[e], qs[e]);
            }
            float dr[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float v = a[s][0] * dm[0][e];
#pragma unroll
                for (int t = 1; t < 5; ++t) v = fmaf(a[s][t], dm[t][e], v);
                float u = r[s][e] * rn[s];
                v = fmaf(rn[s], qs[e] - u * uq[s], v);
                dr[e] = v;
                if (DEPTH) dyp[e] = fmaf(bp[s], v, dyp[e]);
            }
            if (valid) store_row<VEC, NCH>(p.dR + (tok * S + s) * D + doff, lane, dr);
        }
        if (DEPTH && valid) store_row<VEC, NCH>(p.dyprev + tok * D + doff, lane, dyp);
    }
    if (WIDTH) {
        // flush the register accumulators (once per wave): LDS adds, then one row of `partial` per workgroup
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    atomicAdd(&dWp[j * D + doff + ch * 64 * VEC + lane * VEC + v], gw[j][ch * VEC + v]);
        __syncthreads();
        float* out = p.partial + (long)blockIdx.x * (NJ * D + NSC);
        for (int i = tid; i < NJ * D + NSC; i += 256) out[i] = dWp[i];
    }
}

// sum per-block partials and turn d(Wp) into parameter gradients (accumulated into the fp32 grad buffers)
struct HCReduceArgs {
    const float* partial; int nblocks; int D;
    HCParams hp;
    float *g_static_beta, *g_static_alpha, *g_dyn_alpha_fn, *g_dyn_alpha_scale, *g_dyn_beta_fn, *g_dyn_beta_scale, *g_gamma;
};

// grid = D/32 + 1 workgroups.  Workgroup b < D/32: 32 columns d x 8 row-groups, each thread sums nblocks/8 partial
// rows (coalesced over d), LDS tree over the row-groups.  Last workgroup: the 26 scalar gradients.
__global__ __launch_bounds__(256) void hc_reduce_kernel(HCReduceArgs p) {
    __shared__ float red[8][NJ][32];
    const int D = p.D, stride = NJ * D + NSC;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < D / 32) {
        const int dl = tid & 31, part = tid >> 5;
        const int d = blockIdx.x * 32 + dl;
        float acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
        for (int b = part; b < p.nblocks; b += 8) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] += p.partial[(long)b * stride + j * D + d];
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) red[part][j][dl] = acc[j];
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) s += red[q][j][dl];
                acc[j] = s;
            }
            const float g = p.hp.gamma[d] + 1.f;
            float dg = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                p.g_dyn_alpha_fn[d * 5 + t] += g * acc[t];
                dg = fmaf(p.hp.dyn_alpha_fn[d * 5 + t], acc[t], dg);
            }
            p.g_dyn_beta_fn[d] += g * acc[5];
            dg = fmaf(p.hp.dyn_beta_fn[d], acc[5], dg);
            p.g_gamma[d] += dg;
        }
    } else {
        // 26 scalars x 8 row-groups (208 threads)
        const int k = tid & 31, part = tid >> 5;
        float acc = 0.f;
        if (k < 26)
            for (int b = part; b < p.nblocks; b += 8) acc += p.partial[(long)b * stride + NJ * D + k];
        red[part][0][k] = acc;
        __syncthreads();
        if (part == 0 && k < 26) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q][0][k];
            if (k < 20) p.g_static_alpha[k] += s;
            else if (k < 24) p.g_static_beta[k - 20] += s;
            else if (k == 24) p.g_dyn_alpha_scale[0] += s;
            else p.g_dyn_beta_scale[0] += s;
        }
    }
}

template <int VEC, int NCH, int NW>
int launch_fwd(const HCFwdArgs& a, bool depth, bool width, int grid, hipStream_t st) {
    dim3 g(grid), b(256);
    if (depth && width) hipLaunchKernelGGL((hc_fwd_kernel<VEC, NCH, NW, true, true>), g, b, 0, st, a);
    else if (!depth && width) hipLaunchKernelGGL((hc_fwd_kernel<VEC, NCH, NW, false, true>), g, b, 0, st, a);
    else if (depth && !width) hipLaunchKernelGGL((hc_fwd_kernel<VEC, NCH, NW, true, false>), g, b, 0, st, a);
    else return E2K_ERR_ARG;
    return 0;
}
template <int VEC, int NCH, int NW>
int launch_bwd(const HCBwdArgs& a, bool depth, bool width, int grid, hipStream_t st) {
    dim3 g(grid), b(256);
    if (depth && width) hipLaunchKernelGGL((hc_bwd_kernel<VEC, NCH, NW, true, true>), g, b, 0, st, a);
    else if (!depth && width) hipLaunchKernelGGL((hc_bwd_kernel<VEC, NCH, NW, false, true>), g, b, 0, st, a);
    else if (depth && !width) hipLaunchKernelGGL((hc_bwd_kernel<VEC, NCH, NW, true, false>), g, b, 0, st, a);
    else return E2K_ERR_ARG;
    return 0;
}

// D -> (elements per lane per chunk, chunks, waves per token)
#define HC_DISPATCH(D, FN, ...)                                       \
    switch (D) {                                                      \
        case 128: rc = FN<2, 1, 1>(__VA_ARGS__); break;               \
        case 256: rc = FN<4, 1, 1>(__VA_ARGS__); break;               \
        case 512: rc = FN<8, 1, 1>(__VA_ARGS__); break;               \
        case 768: rc = FN<4, 3, 1>(__VA_ARGS__); break;               \
        case 1024: rc = FN<8, 1, 2>(__VA_ARGS__); break;              \
        case 1536: rc = FN<4, 3, 2>(__VA_ARGS__); break;              \
        case 2048: rc = FN<8, 1, 4>(__VA_ARGS__); break;              \
        default: rc = E2K_ERR_SHAPE;                                  \
    }

// the backward keeps more per-lane state (r, dm, d(Wp) accumulators): 4 elements per lane per row at D = 1024 / 512
#define HC_DISPATCH_BWD(D, FN, ...)                                   \
    switch (D) {                                                      \
        case 128: rc = FN<2, 1, 1>(__VA_ARGS__); break;               \
        case 256: rc = FN<4, 1, 1>(__VA_ARGS__); break;               \
        case 512: rc = FN<4, 1, 2>(__VA_ARGS__); break;               \
        case 768: rc = FN<4, 3, 1>(__VA_ARGS__); break;               \
        case 1024: rc = FN<4, 1, 4>(__VA_ARGS__); break;              \
        case 1536: rc = FN<4, 3, 2>(__VA_ARGS__); break;              \
        case 2048: rc = FN<8, 1, 4>(__VA_ARGS__); break;              \
        default: rc = E2K_ERR_SHAPE;                                  \
    }

int tokens_per_block(int D, bool bwd) {
    if (bwd) return D == 1024 || D == 2048 ? 1 : (D == 512 || D == 1536 ? 2 : 4);
    return D == 1024 || D == 1536 ? 2 : (D == 2048 ? 1 : 4);
}

int grid_for(int Mtok, int D, int max_blocks, bool bwd) {
    const int tpb = tokens_per_block(D, bwd);
    int g = (Mtok + tpb - 1) / tpb;
    return g < max_blocks ? g : max_blocks;
}

}  // namespace

extern "C" int e2k_query_hc_coef_width(void) { return CW; }
extern "C" int e2k_query_hc_bwd_blocks(int Mtok, int D) { return grid_for(Mtok, D, 512, true); }
extern "C" int e2k_query_hc_partial_stride(int D) { return NJ * D + NSC; }

extern "C" int e2k_hc_fwd(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
                          float* coef, const float* static_beta, const float* static_alpha,
                          const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
                          const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
                          int has_width, void* stream) {
    if (Mtok <= 0) return 0;
    HCFwdArgs a;
    a.Xin = (const bf16_t*)Xin; a.yprev = (const bf16_t*)yprev; a.coef_prev = coef_prev;
    a.Mout = (bf16_t*)Mout; a.bin = (bf16_t*)bin; a.coef = coef;
    a.hp = HCParams{static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma};
    a.Mtok = Mtok;
    if (!Xin || !Mout || (has_depth && (!yprev || !coef_prev)) || (has_width && (!bin || !coef || !gamma))) return E2K_ERR_ARG;
    int rc = 0;
    HC_DISPATCH(D, launch_fwd, a, has_depth != 0, has_width != 0, grid_for(Mtok, D, 1024, false), (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

extern "C" int e2k_hc_bwd(const void* Xin, const void* yprev, const float* coef_prev, const void* G,
                          const void* dbin, const void* ycur, const float* coef, void* dR, void* dyprev,
                          const float* static_beta, const float* static_alpha, const float* dyn_alpha_fn,
                          const float* dyn_alpha_scale, const float* dyn_beta_fn, const float* dyn_beta_scale,
                          const float* gamma, float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn,
                          float* g_dyn_alpha_scale, float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma,
                          float* partial, int Mtok, int D, int has_depth, int has_width, void* stream) {
    if (Mtok <= 0) return 0;
    HCBwdArgs a;
    a.Xin = (const bf16_t*)Xin; a.yprev = (const bf16_t*)yprev; a.coef_prev = coef_prev;
    a.G = (const bf16_t*)G; a.dbin = (const bf16_t*)dbin; a.ycur = (const bf16_t*)ycur; a.coef = coef;
    a.dR = (bf16_t*)dR; a.dyprev = (bf16_t*)dyprev;
    a.hp = HCParams{static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma};
    a.partial = partial; a.Mtok = Mtok;
    if (!G || (has_depth && (!yprev || !coef_prev || !dyprev))) return E2K_ERR_ARG;
    if (has_width && (!Xin || !dbin || !ycur || !coef || !dR || !partial || !gamma || !g_gamma)) return E2K_ERR_ARG;
    const int grid = grid_for(Mtok, D, 512, true);      // two workgroups per CU (the kernel fits 2 waves per SIMD)
    int rc = 0;
    HC_DISPATCH_BWD(D, launch_bwd, a, has_depth != 0, has_width != 0, grid, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    if (has_width) {
        HCReduceArgs r;
        r.partial = partial; r.nblocks = grid; r.D = D; r.hp = a.hp;
        r.g_static_beta = g_static_beta; r.g_static_alpha = g_static_alpha; r.g_dyn_alpha_fn = g_dyn_alpha_fn;
        r.g_dyn_alpha_scale = g_dyn_alpha_scale; r.g_dyn_beta_fn = g_dyn_beta_fn; r.g_dyn_beta_scale = g_dyn_beta_scale;
        r.g_gamma = g_gamma;
        hipLaunchKernelGGL(hc_reduce_kernel, dim3(D / 32 + 1), dim3(256), 0, (hipStream_t)stream, r);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}
