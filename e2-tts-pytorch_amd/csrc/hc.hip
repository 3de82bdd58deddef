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
#include <type_traits>
#include "e2k_device.h"
#include "plan.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"

using namespace e2k;

namespace {

constexpr int S = 4, NJ = 6, CW = 52;
constexpr int CA = 0, CB = 20, CP = 24, CRN = 48;   // a[s*5+t], b[s], P[s*6+j] (pre-tanh dots), rn[s]
constexpr int NSC = 32;                               // scalar partials: dA[20], dB[4], dsa, dsb, pad

struct HCParams {
    const float* static_beta; const float* static_alpha; const float* dyn_alpha_fn; const float* dyn_alpha_scale;
    const float* dyn_beta_fn; const float* dyn_beta_scale; const float* gamma;
};

// Wp[j][d] = (gamma[d]+1) * W[d][j]   (j < 5: dynamic_alpha_fn column, j = 5: dynamic_beta_fn), fp32 in LDS.
// A lane that owns 8 consecutive elements (VEC = 8) would read them as two 16-byte LDS accesses 32 bytes apart from
// its neighbour's (bank conflicts); such rows are stored with the two halves of every 8-element group split into
// separate 256-float planes, so that each 16-byte access of a wave is contiguous over the lanes.
template <int VEC> __device__ __forceinline__ int wp_pos(int d) {
    if (VEC != 8) return d;
    return (d & ~511) | ((d & 4) << 6) | ((d & 504) >> 1) | (d & 3);
}
template <int VEC>
__device__ __forceinline__ void stage_wp(float* Wp, const HCParams& hp, int D, int tid) {
    for (int d = tid; d < D; d += 256) {
        float g = hp.gamma[d] + 1.f;
        const int q = wp_pos<VEC>(d);
#pragma unroll
        for (int t = 0; t < 5; ++t) Wp[t * D + q] = g * hp.dyn_alpha_fn[d * 5 + t];
        Wp[5 * D + q] = g * hp.dyn_beta_fn[d];
    }
}
// the lane's elements of one Wp row (same element order as load_row<VEC, NCH>)
template <int VEC, int NCH> __device__ __forceinline__ void load_wp(const float* row, int lane, float* f) {
    if (VEC == 8) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v = *reinterpret_cast<const f32x4*>(row + c * 512 + h * 256 + lane * 4);
                f[c * 8 + h * 4] = v[0]; f[c * 8 + h * 4 + 1] = v[1]; f[c * 8 + h * 4 + 2] = v[2]; f[c * 8 + h * 4 + 3] = v[3];
            }
    } else {
        load_row_f32<VEC, NCH>(row, lane, f);
    }
}

// packed (raw bf16 pairs) row loads: the next token's rows are fetched into these while the current token is being
// processed, half the registers of the unpacked floats
template <int VEC, int NCH> __device__ __forceinline__ void load_raw(const bf16_t* row, int lane, unsigned* w) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bf16_t* q = row + c * 64 * VEC + lane * VEC;
        if (VEC == 8) { u32x4 v = ld<u32x4>(q); w[c * 4] = v[0]; w[c * 4 + 1] = v[1]; w[c * 4 + 2] = v[2]; w[c * 4 + 3] = v[3]; }
        else if (VEC == 4) { u32x2 v = ld<u32x2>(q); w[c * 2] = v[0]; w[c * 2 + 1] = v[1]; }
        else w[c] = ld<unsigned>(q);
    }
}
template <int N> __device__ __forceinline__ void unpack_raw(const unsigned* w, float* f) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { f[2 * i] = bflo(w[i]); f[2 * i + 1] = bfhi(w[i]); }
}
// dot product of two per-lane rows with packed fp32 FMAs (v_pk_fma_f32)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ float dot_pk(const float* a, const float* b) {
    f32x2_ acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        f32x2_ x = {a[i], a[i + 1]}, y = {b[i], b[i + 1]};
        acc = __builtin_elementwise_fma(x, y, acc);
    }
    return acc[0] + acc[1];
}
// tanh on the hardware exp2 / rcp units
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = fast_exp2(-2.885390081777927f * fabsf(x));
    const float t = (1.0f - e) * fast_rcp(1.0f + e);
    return x < 0.f ? -t : t;
}

constexpr int NRED = 40;                             // LDS exchange row (28 forward -- 34 with the branch-input norm -- / 24 backward values used)

// Reduce N per-lane partials (N a multiple of 8) over the NW waves that share a token, leaving the totals spread over LANES instead of
// broadcast: wave_sum_rows leaves the total of value i + (N / 4) (r & 1) + (N / 2) (r >> 1) in register i of row r (two swap levels + four
// in-row DPP levels on N / 4 registers), the first lane of each row stores its N / 4 totals to LDS at their value indices, and after one
// barrier (the caller's) lane l (< 32) picks up the total with index `pick` (summed over the NW waves).  The per-token scalar math that
// follows then runs once per lane-slot instead of N times on wave-uniform values.
template <int N>
__device__ __forceinline__ void token_scatter_nosync(float (&v)[N], float (*red)[4][NRED], int parity, int wave, int lane, int base = 0) {
    static_assert(N % 8 == 0 && N <= NRED, "");
    wave_sum_rows<N>(v);
    if ((lane & 15) == 0) {
        float* dst = &red[parity][wave][base + wave_sum_rows_index<N>(0, lane >> 4)];        // (N / 4 consecutive totals, 8-byte aligned)
#pragma unroll
        for (int i = 0; i < N / 4; i += 2) *reinterpret_cast<f32x2_*>(dst + i) = f32x2_{v[i], v[i + 1]};
    }
}
template <int N, int NW>
__device__ __forceinline__ void token_scatter(float* vals, float (*red)[4][NRED], int parity, int wave, int lane) {
    constexpr int NP = (N + 7) & ~7;
    float v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) v[i] = i < N ? vals[i] : 0.f;
    token_scatter_nosync<NP>(v, red, parity, wave, lane);
    __syncthreads();
}
template <int NW>
__device__ __forceinline__ float token_pick(float (*red)[4][NRED], int parity, int wave, int pick) {
    const int w0 = (wave / NW) * NW;
    float s = red[parity][w0][pick];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[parity][w0 + w][pick];
    return s;
}

struct HCFwdArgs {
    const bf16_t* Xin; const bf16_t* yprev; const float* coef_prev;
    bf16_t* Mout; bf16_t* bin; float* coef;
    HCParams hp;
    int Mtok;
    // NORM (e2k_hc_fwd_norm): the (Adaptive)RMSNorm that follows every width connection in the backbone (e2_tts.py:875,881,908-914,926,937:
    // `x, add_residual = hc(x); x = norm(x[, cond])`), applied to the branch input while it is in registers: xn = bin / |bin| sqrt(D)
    // (ngamma[row / rows_per_batch] + ngam_off); nrn[tok] = 1 / |bin| (what e2k_rmsnorm_bwd wants).  bin may be NULL then (no-grad forward)
    const float* ngamma; long ngam_ld; float ngam_off; int rows_per_batch; bf16_t* xn; float* nrn;
};

template <int VEC, int NCH, int NW, bool DEPTH, bool WIDTH, bool NORM = false>
__global__ __launch_bounds__(256) void hc_fwd_kernel(HCFwdArgs p) {
    static_assert(WIDTH || !NORM, "");
    constexpr int EPL = VEC * NCH, DS = 64 * EPL, D = DS * NW, TPB = 4 / NW;
    __shared__ __attribute__((aligned(16))) float Wp[WIDTH ? NJ * D : 4];
    __shared__ __attribute__((aligned(16))) float red[2][4][NRED];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int slot = wave / NW, doff = (wave % NW) * DS;
    // lane slot (ls, lj) = (stream, column) of the per-token scalar math: lj < 5 alpha column, 5 beta, 6 sum of squares
    const int ls = (lane >> 3) & 3, lj = lane & 7;
    float scale_l = 0.f, stat_l = 0.f;
    if (WIDTH) {
        stage_wp<VEC>(Wp, p.hp, D, tid);
        if (lane < 32 && lj < 5) { scale_l = p.hp.dyn_alpha_scale[0]; stat_l = p.hp.static_alpha[ls * 5 + lj]; }
        if (lane < 32 && lj == 5) { scale_l = p.hp.dyn_beta_scale[0]; stat_l = p.hp.static_beta[ls]; }
        __syncthreads();
    }
    const float sqrtD = sqrtf((float)D);
    const int per_iter = gridDim.x * TPB;
    const int niter = (p.Mtok + per_iter - 1) / per_iter;
    unsigned rawX[S][EPL / 2], rawY[EPL / 2];
    float bpn[S];
    auto prefetch = [&](int it) {
        const int tr = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const long tk = tr < p.Mtok ? tr : p.Mtok - 1;
#pragma unroll
        for (int s = 0; s < S; ++s) load_raw<VEC, NCH>(p.Xin + (tk * S + s) * D + doff, lane, rawX[s]);
        if (DEPTH) {
            load_raw<VEC, NCH>(p.yprev + tk * D + doff, lane, rawY);
#pragma unroll
            for (int s = 0; s < S; ++s) bpn[s] = p.coef_prev[tk * CW + CB + s];
        }
    };
    prefetch(0);
    // all-valid iterations run without any validity branch (lets the compiler count outstanding stores instead of
    // draining them before it may touch the prefetched rows); at most one trailing iteration is checked
    auto body = [&](int it, auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
        const int tok_raw = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const bool valid = !CHECK || tok_raw < p.Mtok;
        const long tok = valid ? tok_raw : p.Mtok - 1;
        float r[S][EPL];
#pragma unroll
        for (int s = 0; s < S; ++s) unpack_raw<EPL>(rawX[s], r[s]);
        if (DEPTH) {
            float y[EPL];
            unpack_raw<EPL>(rawY, y);
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) r[s][e] = fmaf(bpn[s], y[e], r[s][e]);
            }
        }
        if (!CHECK) prefetch(it + 1);       // (clamped to the last token past the end)
        if (!WIDTH) {
            if (valid) {
#pragma unroll
                for (int s = 0; s < S; ++s) store_row<VEC, NCH>(p.Mout + (tok * S + s) * D + doff, lane, r[s]);
            }
            return;
        }
        constexpr int NPART = NORM ? 34 : 28;
        float part[NPART];    // [s*7 + j], j = 6: sum of squares; NORM: [28 + pair] the six cross products <r_s, r_s'>, s < s'
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float w[EPL];
            load_wp<VEC, NCH>(Wp + j * D + doff, lane, w);
#pragma unroll
            for (int s = 0; s < S; ++s) part[s * 7 + j] = dot_pk<EPL>(r[s], w);
        }
#pragma unroll
        for (int s = 0; s < S; ++s) part[s * 7 + 6] = dot_pk<EPL>(r[s], r[s]);
        if (NORM) {
            // |bin|^2 = sum_{s,s'} a_s a_s' <r_s, r_s'> (bin = sum_s a[s][0] r_s): the Gram matrix of the four streams rides in the ONE
            // reduction round the coefficients need anyway, so the norm of the branch input costs no second exchange between the
            // token's waves
            int q = 28;
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int s2 = s + 1; s2 < S; ++s2) part[q++] = dot_pk<EPL>(r[s], r[s2]);
        }
        token_scatter<NPART, NW>(part, red, it & 1, wave, lane);
        // lane-parallel coefficients: lane (ls, lj) owns a[ls][lj] (lj < 5) / b[ls] (lj == 5)
        const float dotl = token_pick<NW>(red, it & 1, wave, ls * 7 + (lj < 7 ? lj : 6));
        const float ssl = token_pick<NW>(red, it & 1, wave, ls * 7 + 6);
        const float rnl = fast_rsq(fmaxf(ssl, 1e-24f));
        const float Pl = dotl * rnl * sqrtD;
        const float coefl = fmaf(tanh_fast(Pl), scale_l, stat_l);
        float a[S][5];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < 5; ++t) a[s][t] = lane_bcast(coefl, s * 8 + t);
        if (valid && doff == 0 && lane < 32) {
            float* c = p.coef + tok * CW;
            if (lj < 5) c[CA + ls * 5 + lj] = coefl;
            if (lj == 5) c[CB + ls] = coefl;
            if (lj < 6) c[CP + ls * NJ + lj] = Pl;
            if (lj == 6) c[CRN + ls] = rnl;
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
                if (t == 0) {
                    if (!NORM || p.bin) store_row<VEC, NCH>(p.bin + tok * D + doff, lane, m);
                    if (NORM) {
                        float ssb = 0.f;
                        int q = 28;
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            ssb = fmaf(a[s][0] * a[s][0], token_pick<NW>(red, it & 1, wave, s * 7 + 6), ssb);
#pragma unroll
                            for (int s2 = s + 1; s2 < S; ++s2) ssb = fmaf(2.f * a[s][0] * a[s2][0], token_pick<NW>(red, it & 1, wave, q++), ssb);
                        }
                        const float rnb = 1.f / fmaxf(fast_sqrt(fmaxf(ssb, 0.f)), 1e-12f);
                        const float sc = rnb * sqrtD;
                        float gn[EPL];
                        load_row_f32<VEC, NCH>(p.ngamma + (tok / p.rows_per_batch) * p.ngam_ld + doff, lane, gn);
#pragma unroll
                        for (int e = 0; e < EPL; ++e) m[e] = m[e] * sc * (gn[e] + p.ngam_off);
                        store_row<VEC, NCH>(p.xn + tok * D + doff, lane, m);
                        if (p.nrn && doff == 0 && lane == 0) p.nrn[tok] = rnb;
                    }
                } else store_row<VEC, NCH>(p.Mout + (tok * S + (t - 1)) * D + doff, lane, m);
            }
        }
    };
    const int nfull = p.Mtok / per_iter;
    for (int it = 0; it < nfull; ++it) body(it, std::false_type{});
    if (nfull < niter) body(nfull, std::true_type{});
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
__global__ __launch_bounds__(256, 3) void hc_bwd_kernel(HCBwdArgs p) {
    constexpr int EPL = VEC * NCH, DS = 64 * EPL, D = DS * NW, TPB = 4 / NW;
    __shared__ __attribute__((aligned(16))) float Wp[WIDTH ? NJ * D : 4];
    __shared__ __attribute__((aligned(16))) float dWp[WIDTH ? TPB * (NJ * D + NSC) : 4];      // one d(Wp) + scalars copy per token slot
    __shared__ __attribute__((aligned(16))) float red[2][4][NRED];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int slot = wave / NW, doff = (wave % NW) * DS;
    // lane slot (ls, lj) of the per-token scalar math: pair (stream ls, column lj); lj < 5 alpha, 5 beta
    const int ls = (lane >> 3) & 3, lj = lane & 7;
    const bool lact = lane < 32 && lj < 6;
    float scale_l = 0.f;
    float accD = 0.f, accT = 0.f;        // per-lane sums over this wave's tokens: d(static), d(scale) contributions
    if (WIDTH) {
        stage_wp<VEC>(Wp, p.hp, D, tid);
        if (lact) scale_l = lj < 5 ? p.hp.dyn_alpha_scale[0] : p.hp.dyn_beta_scale[0];
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
    // next-token rows, packed: G[4], and with WIDTH also Xin[4], yprev, dbin, ycur
    unsigned rawG[S][EPL / 2], rawX[WIDTH ? S : 1][EPL / 2], rawY[EPL / 2], rawB[EPL / 2], rawC[EPL / 2];
    float bpn[S], Pln = 0.f;
    auto prefetch = [&](int it) {
        const int tr = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const long tk = tr < p.Mtok ? tr : p.Mtok - 1;
        if (WIDTH) {
            if (lact) Pln = p.coef[tk * CW + CP + ls * NJ + lj];     // this lane's pre-tanh dot (used after the reduction)
#pragma unroll
            for (int s = 0; s < S; ++s) load_raw<VEC, NCH>(p.Xin + (tk * S + s) * D + doff, lane, rawX[s]);
            if (DEPTH) load_raw<VEC, NCH>(p.yprev + tk * D + doff, lane, rawY);
            load_raw<VEC, NCH>(p.dbin + tk * D + doff, lane, rawB);
            load_raw<VEC, NCH>(p.ycur + tk * D + doff, lane, rawC);
        }
#pragma unroll
        for (int s = 0; s < S; ++s) load_raw<VEC, NCH>(p.G + (tk * S + s) * D + doff, lane, rawG[s]);
        if (DEPTH) {
#pragma unroll
            for (int s = 0; s < S; ++s) bpn[s] = p.coef_prev[tk * CW + CB + s];
        }
    };
    prefetch(0);
    // all-valid iterations run without any validity branch (lets the compiler count outstanding stores instead of
    // draining them before it may touch the prefetched rows); at most one trailing iteration is checked
    auto body = [&](int it, auto checked) {
        constexpr bool CHECK = decltype(checked)::value;
        const int tok_raw = (it * gridDim.x + blockIdx.x) * TPB + slot;
        const bool valid = !CHECK || tok_raw < p.Mtok;
        const long tok = valid ? tok_raw : p.Mtok - 1;
        const float vf = valid ? 1.f : 0.f;
        float bp[S];
        if (DEPTH) {
#pragma unroll
            for (int s = 0; s < S; ++s) bp[s] = bpn[s];
        }
        if (!WIDTH) {
            // only the depth connection of the previous instance: dy_prev = sum_s b_prev[s] * dX[s]
            float dy[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) dy[e] = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                float g[EPL];
                unpack_raw<EPL>(rawG[s], g);
#pragma unroll
                for (int e = 0; e < EPL; ++e) dy[e] = fmaf(bp[s], g[e], dy[e]);
            }
            if (!CHECK) prefetch(it + 1);       // (clamped to the last token past the end)
            if (valid) store_row<VEC, NCH>(p.dyprev + tok * D + doff, lane, dy);
            return;
        }
        float r[S][EPL], dm[5][EPL], yc[EPL];
#pragma unroll
        for (int s = 0; s < S; ++s) unpack_raw<EPL>(rawX[s], r[s]);
        if (DEPTH) {
            float y[EPL];
            unpack_raw<EPL>(rawY, y);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < EPL; ++e) r[s][e] = fmaf(bp[s], y[e], r[s][e]);
        }
        unpack_raw<EPL>(rawB, dm[0]);
#pragma unroll
        for (int s = 0; s < S; ++s) unpack_raw<EPL>(rawG[s], dm[s + 1]);
        unpack_raw<EPL>(rawC, yc);
        const float Pl = Pln;
        if (!CHECK) prefetch(it + 1);       // (clamped to the last token past the end)

        const float* cf = p.coef + tok * CW;
        // 24 dots: da[s][t] = dm_t . r_s (index s*6 + t) ; db[s] = G_s . y_cur (index s*6 + 5); reduced eight at a time (wave_sum_rows: its
        // cost is linear in the count, and few of them are live at once) and handed to LDS by the first lane of each row
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float d8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int idx = g * 8 + q, s = idx / 6, t = idx % 6;
                d8[q] = t < 5 ? dot_pk<EPL>(dm[t], r[s]) : dot_pk<EPL>(dm[s + 1], yc);
            }
            token_scatter_nosync<8>(d8, red, it & 1, wave, lane, g * 8);
        }
        // independent of the reduction, done while the other waves arrive: pre[s] = sum_t a[s][t] dm_t
        float a[S][5], rn[S], pre[S][EPL];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rn[s] = sload(cf + CRN + s);
#pragma unroll
            for (int t = 0; t < 5; ++t) a[s][t] = sload(cf + CA + s * 5 + t);
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float v = a[s][0] * dm[0][e];
#pragma unroll
                for (int t = 1; t < 5; ++t) v = fmaf(a[s][t], dm[t][e], v);
                pre[s][e] = v;
            }
        }
        float w6[NJ][EPL];       // this lane's Wp elements (LDS reads in flight across the barrier)
#pragma unroll
        for (int j = 0; j < NJ; ++j) load_wp<VEC, NCH>(Wp + j * D + doff, lane, w6[j]);
        __syncthreads();
        // lane-parallel: lane (ls, lj) turns its dot into c = d(pre-tanh dot); group-of-8 sums give uq[ls]
        const float dotl = lact ? token_pick<NW>(red, it & 1, wave, ls * 6 + (lj < 6 ? lj : 0)) * vf : 0.f;
        const float th = tanh_fast(Pl);
        const float cl = dotl * scale_l * (1.f - th * th);
        accD += dotl;
        accT = fmaf(dotl, th, accT);
        const float uql = group8_sum(cl * Pl);
        const float cql = cl * sqrtD;
        float cq[S][NJ], uq[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            uq[s] = lane_bcast(uql, s * 8);
#pragma unroll
            for (int j = 0; j < NJ; ++j) cq[s][j] = lane_bcast(cql, s * 8 + j);
        }
        // u = r * rn (in place) ; q[s][e] = sum_j cq[s][j] Wp[j][d] ; gw[j][e] += sum_s cq[s][j] u[s][e]
        // dr[s] = sum_t a[s][t] dm_t + rn[s] (q[s] - u[s] uq[s]).  Each Wp row is read from LDS once per token.
        float qs[S][EPL];
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int e = 0; e < EPL; ++e) { r[s][e] *= rn[s]; qs[s][e] = 0.f; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    qs[s][e] = fmaf(cq[s][j], w6[j][e], qs[s][e]);
                    gw[j][e] = fmaf(cq[s][j], r[s][e], gw[j][e]);
                }
        }
        float dyp[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) dyp[e] = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            float dr[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float v = fmaf(rn[s], qs[s][e] - r[s][e] * uq[s], pre[s][e]);
                dr[e] = v;
                if (DEPTH) dyp[e] = fmaf(bp[s], v, dyp[e]);
            }
            if (valid) store_row<VEC, NCH>(p.dR + (tok * S + s) * D + doff, lane, dr);
        }
        if (DEPTH && valid) store_row<VEC, NCH>(p.dyprev + tok * D + doff, lane, dyp);
    };
    const int nfull = p.Mtok / per_iter;
    for (int it = 0; it < nfull; ++it) body(it, std::false_type{});
    if (nfull < niter) body(nfull, std::true_type{});
    if (WIDTH) {
        // flush the register accumulators (once per wave): plain LDS stores into the wave's token slot (every element of a
        // slot has exactly one owner; LDS float atomics cost ~0.6 us per instruction and workgroup round on this part, see
        // dwconv_bwd_kernel), the slots are added on the way out: one row of `partial` per workgroup
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    dWp[(slot * NJ + j) * D + doff + ch * 64 * VEC + lane * VEC + v] = gw[j][ch * VEC + v];
        if (doff == 0) {             // scalar gradients: d(static_alpha/beta) = sum of dots, d(scale) = sum of dot * tanh
            float* sc = dWp + TPB * NJ * D + slot * NSC;
            const float ta = wave_sum_fast(lact && lj < 5 ? accT : 0.f), tb = wave_sum_fast(lact && lj == 5 ? accT : 0.f);
            if (lact) sc[lj < 5 ? ls * 5 + lj : 20 + ls] = accD;        // (one lane per entry)
            if (lane == 0) { sc[24] = ta; sc[25] = tb; }
            if (lane >= 26 && lane < NSC) sc[lane] = 0.f;
        }
        __syncthreads();
        float* out = p.partial + (long)blockIdx.x * (NJ * D + NSC);
        for (int i = tid; i < NJ * D + NSC; i += 256) {
            float v;
            if (i < NJ * D) {
                v = dWp[i];
#pragma unroll
                for (int t = 1; t < TPB; ++t) v += dWp[t * NJ * D + i];
            } else {
                v = dWp[TPB * NJ * D + (i - NJ * D)];
#pragma unroll
                for (int t = 1; t < TPB; ++t) v += dWp[TPB * NJ * D + t * NSC + (i - NJ * D)];
            }
            out[i] = v;
        }
    }
}

// sum per-block partials and turn d(Wp) into parameter gradients (accumulated into the fp32 grad buffers)
struct HCReduceArgs {
    const float* partial; int nblocks; int D;
    HCParams hp;
    float *g_static_beta, *g_static_alpha, *g_dyn_alpha_fn, *g_dyn_alpha_scale, *g_dyn_beta_fn, *g_dyn_beta_scale, *g_gamma;
};

// grid = (D/32 + 1, RSPLIT).  Workgroup (x < D/32, y): 32 columns d x 8 row-groups over the partial rows
// b = y*8 + part (mod 8*RSPLIT), coalesced over d, LDS tree over the row-groups, then one atomic add per gradient
// element (RSPLIT adds per address).  x = D/32: the 26 scalar gradients.
constexpr int RSPLIT = 8;
__device__ __forceinline__ void hc_reduce_body(const HCReduceArgs& p, int bx, int by) {
    __shared__ float red[8][NJ][32];
    const int D = p.D, stride = NJ * D + NSC;
    const int tid = threadIdx.x;
    const int part = tid >> 5, row0 = by * 8 + part;
    if (bx < D / 32) {
        const int dl = tid & 31;
        const int d = bx * 32 + dl;
        float acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
        for (int b = row0; b < p.nblocks; b += 8 * RSPLIT) {
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
                atomicAdd(&p.g_dyn_alpha_fn[d * 5 + t], g * acc[t]);
                dg = fmaf(p.hp.dyn_alpha_fn[d * 5 + t], acc[t], dg);
            }
            atomicAdd(&p.g_dyn_beta_fn[d], g * acc[5]);
            dg = fmaf(p.hp.dyn_beta_fn[d], acc[5], dg);
            atomicAdd(&p.g_gamma[d], dg);
        }
    } else {
        // 26 scalars x 8 row-groups (208 threads)
        const int k = tid & 31;
        float acc = 0.f;
        if (k < 26)
            for (int b = row0; b < p.nblocks; b += 8 * RSPLIT) acc += p.partial[(long)b * stride + NJ * D + k];
        red[part][0][k] = acc;
        __syncthreads();
        if (part == 0 && k < 26) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[q][0][k];
            if (k < 20) atomicAdd(&p.g_static_alpha[k], s);
            else if (k < 24) atomicAdd(&p.g_static_beta[k - 20], s);
            else if (k == 24) atomicAdd(&p.g_dyn_alpha_scale[0], s);
            else atomicAdd(&p.g_dyn_beta_scale[0], s);
        }
    }
}
__global__ __launch_bounds__(256) void hc_reduce_kernel(HCReduceArgs p) { hc_reduce_body(p, blockIdx.x, blockIdx.y); }

// the reductions of up to HC_BATCH_MAX hyper-connections in ONE launch (grid.z = item): a layer's six (audio 3-4, text 3) are
// launched together at the end of the layer instead of one 264-workgroup launch each (144 launches per cfg3 step, each
// reading its 19 MB of partials at a quarter of the HBM rate)
constexpr int HC_BATCH_MAX = 8;
struct HCReduceBatch { HCReduceArgs it[HC_BATCH_MAX]; int n; };
__global__ __launch_bounds__(256) void hc_reduce_batch_kernel(HCReduceBatch b) {
    const HCReduceArgs& p = b.it[blockIdx.z];
    if ((int)blockIdx.x > p.D / 32) return;          // (the grid is sized for the widest item)
    hc_reduce_body(p, blockIdx.x, blockIdx.y);
}

template <int VEC, int NCH, int NW>
int launch_fwd(const HCFwdArgs& a, bool depth, bool width, int grid, hipStream_t st) {
    dim3 g(grid), b(256);
    if (a.xn) {
        if (depth && width) hipLaunchKernelGGL((hc_fwd_kernel<VEC, NCH, NW, true, true, true>), g, b, 0, st, a);
        else if (!depth && width) hipLaunchKernelGGL((hc_fwd_kernel<VEC, NCH, NW, false, true, true>), g, b, 0, st, a);
        else return E2K_ERR_ARG;
        return 0;
    }
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
        case 384: rc = FN<2, 3, 1>(__VA_ARGS__); break;               \
        case 512: rc = FN<8, 1, 1>(__VA_ARGS__); break;               \
        case 640: rc = FN<2, 5, 1>(__VA_ARGS__); break;               \
        case 768: rc = FN<4, 3, 1>(__VA_ARGS__); break;               \
        case 896: rc = FN<2, 7, 1>(__VA_ARGS__); break;               \
        case 1024: rc = FN<8, 1, 2>(__VA_ARGS__); break;              \
        case 1280: rc = FN<4, 5, 1>(__VA_ARGS__); break;              \
        case 1792: rc = FN<2, 7, 2>(__VA_ARGS__); break;              \
        case 1536: rc = FN<4, 3, 2>(__VA_ARGS__); break;              \
        case 2048: rc = FN<8, 1, 4>(__VA_ARGS__); break;              \
        default: rc = E2K_ERR_SHAPE;                                  \
    }

// the backward keeps more per-lane state (r, dm, d(Wp) accumulators): 4 elements per lane per row at D = 1024 / 512
#define HC_DISPATCH_BWD(D, FN, ...)                                   \
    switch (D) {                                                      \
        case 128: rc = FN<2, 1, 1>(__VA_ARGS__); break;               \
        case 256: rc = FN<4, 1, 1>(__VA_ARGS__); break;               \
        case 384: rc = FN<2, 3, 1>(__VA_ARGS__); break;               \
        case 512: rc = FN<4, 1, 2>(__VA_ARGS__); break;               \
        case 640: rc = FN<2, 5, 1>(__VA_ARGS__); break;               \
        case 768: rc = FN<4, 3, 1>(__VA_ARGS__); break;               \
        case 896: rc = FN<2, 7, 1>(__VA_ARGS__); break;               \
        case 1024: rc = FN<4, 1, 4>(__VA_ARGS__); break;              \
        case 1280: rc = FN<2, 5, 2>(__VA_ARGS__); break;              \
        case 1792: rc = FN<2, 7, 2>(__VA_ARGS__); break;              \
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
extern "C" int e2k_query_hc_bwd_blocks(int Mtok, int D) { return grid_for(Mtok, D, 768, true); }
extern "C" int e2k_query_hc_partial_stride(int D) { return NJ * D + NSC; }

static int hc_fwd_impl(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
                          float* coef, const float* static_beta, const float* static_alpha,
                          const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
                          const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
                          int has_width, const float* ngamma, int64_t ngam_ld, float ngam_off, int rows_per_batch, void* xn, float* nrn,
                          void* stream) {
    if (Mtok <= 0) return 0;
    HCFwdArgs a;
    a.Xin = (const bf16_t*)Xin; a.yprev = (const bf16_t*)yprev; a.coef_prev = coef_prev;
    a.Mout = (bf16_t*)Mout; a.bin = (bf16_t*)bin; a.coef = coef;
    a.hp = HCParams{static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma};
    a.Mtok = Mtok;
    a.ngamma = ngamma; a.ngam_ld = ngam_ld; a.ngam_off = ngam_off; a.rows_per_batch = rows_per_batch; a.xn = (bf16_t*)xn; a.nrn = nrn;
    if (xn && (!has_width || !ngamma || rows_per_batch <= 0)) return E2K_ERR_ARG;
    if (!Xin || !Mout || (has_depth && (!yprev || !coef_prev)) || (has_width && ((!bin && !xn) || !coef || !gamma))) return E2K_ERR_ARG;
    int rc = 0;
    HC_DISPATCH(D, launch_fwd, a, has_depth != 0, has_width != 0, grid_for(Mtok, D, 1024, false), (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

static int hc_bwd_impl(const void* Xin, const void* yprev, const float* coef_prev, const void* G,
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
    const int grid = grid_for(Mtok, D, 768, true);      // three workgroups per CU (3 waves per SIMD, 50 KB LDS each)
    int rc = 0;
    HC_DISPATCH_BWD(D, launch_bwd, a, has_depth != 0, has_width != 0, grid, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    if (has_width == 1) {            // (has_width == 2: the caller runs e2k_hc_bwd_reduce later, e.g. on another stream)
        HCReduceArgs r;
        r.partial = partial; r.nblocks = grid; r.D = D; r.hp = a.hp;
        r.g_static_beta = g_static_beta; r.g_static_alpha = g_static_alpha; r.g_dyn_alpha_fn = g_dyn_alpha_fn;
        r.g_dyn_alpha_scale = g_dyn_alpha_scale; r.g_dyn_beta_fn = g_dyn_beta_fn; r.g_dyn_beta_scale = g_dyn_beta_scale;
        r.g_gamma = g_gamma;
        hipLaunchKernelGGL(hc_reduce_kernel, dim3(D / 32 + 1, RSPLIT), dim3(256), 0, (hipStream_t)stream, r);
        E2K_CHECK_LAUNCH();
    }
    return 0;
}

// second half of e2k_hc_bwd(has_width = 2): the per-workgroup partials -> parameter gradients.  Nothing on the backward chain
// reads these gradients, so the schedule runs it with the weight-gradient GEMMs on the WGRAD lane
static int hc_bwd_reduce_impl(const float* partial, const float* dyn_alpha_fn, const float* dyn_beta_fn, const float* gamma,
                              float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn, float* g_dyn_alpha_scale,
                              float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma, int Mtok, int D, void* stream) {
    if (Mtok <= 0) return 0;
    if (!partial || !gamma || !g_gamma || !dyn_alpha_fn || !dyn_beta_fn) return E2K_ERR_ARG;
    if (D % 32) return E2K_ERR_SHAPE;
    HCReduceArgs r;
    r.partial = partial; r.nblocks = grid_for(Mtok, D, 768, true); r.D = D;
    r.hp = HCParams{nullptr, nullptr, dyn_alpha_fn, nullptr, dyn_beta_fn, nullptr, gamma};
    r.g_static_beta = g_static_beta; r.g_static_alpha = g_static_alpha; r.g_dyn_alpha_fn = g_dyn_alpha_fn;
    r.g_dyn_alpha_scale = g_dyn_alpha_scale; r.g_dyn_beta_fn = g_dyn_beta_fn; r.g_dyn_beta_scale = g_dyn_beta_scale;
    r.g_gamma = g_gamma;
    hipLaunchKernelGGL(hc_reduce_kernel, dim3(D / 32 + 1, RSPLIT), dim3(256), 0, (hipStream_t)stream, r);
    E2K_CHECK_LAUNCH();
    return 0;
}

struct HCReduceBatchHost { e2k_hc_reduce_item it[HC_BATCH_MAX]; int n; };
static int hc_bwd_reduce_batch_impl(HCReduceBatchHost h, void* stream) {
    HCReduceBatch b;
    b.n = 0;
    int dmax = 0;
    for (int i = 0; i < h.n; ++i) {
        const e2k_hc_reduce_item& q = h.it[i];
        if (q.Mtok <= 0) continue;
        if (!q.partial || !q.gamma || !q.g_gamma || !q.dyn_alpha_fn || !q.dyn_beta_fn) return E2K_ERR_ARG;
        if (q.D % 32) return E2K_ERR_SHAPE;
        HCReduceArgs& r = b.it[b.n++];
        r.partial = q.partial; r.nblocks = grid_for(q.Mtok, q.D, 768, true); r.D = q.D;
        r.hp = HCParams{nullptr, nullptr, q.dyn_alpha_fn, nullptr, q.dyn_beta_fn, nullptr, q.gamma};
        r.g_static_beta = q.g_static_beta; r.g_static_alpha = q.g_static_alpha; r.g_dyn_alpha_fn = q.g_dyn_alpha_fn;
        r.g_dyn_alpha_scale = q.g_dyn_alpha_scale; r.g_dyn_beta_fn = q.g_dyn_beta_fn; r.g_dyn_beta_scale = q.g_dyn_beta_scale;
        r.g_gamma = q.g_gamma;
        dmax = q.D > dmax ? q.D : dmax;
    }
    if (b.n == 0) return 0;
    hipLaunchKernelGGL(hc_reduce_batch_kernel, dim3(dmax / 32 + 1, RSPLIT, b.n), dim3(256), 0, (hipStream_t)stream, b);
    E2K_CHECK_LAUNCH();
    return 0;
}

// ---- C ABI: every compute entry point goes through e2k::dispatch (plan.h) so that a launch plan can record it

extern "C" int e2k_hc_bwd_reduce_batch(const e2k_hc_reduce_item* items, int n, void* stream) {
    if (n < 0 || n > HC_BATCH_MAX || (n > 0 && !items)) return E2K_ERR_ARG;
    HCReduceBatchHost h;
    h.n = n;
    for (int i = 0; i < n; ++i) h.it[i] = items[i];
    for (int i = n; i < HC_BATCH_MAX; ++i) h.it[i] = e2k_hc_reduce_item{};
    return e2k::dispatch("hc_bwd_reduce_batch", hc_bwd_reduce_batch_impl, h, stream);
}

extern "C" int e2k_hc_fwd(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
                          float* coef, const float* static_beta, const float* static_alpha,
                          const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
                          const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
                          int has_width, void* stream) {
    return e2k::dispatch("hc_fwd", hc_fwd_impl, Xin, yprev, coef_prev, Mout, bin, coef, static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma, Mtok, D, has_depth, has_width,
                         (const float*)nullptr, (int64_t)0, 0.f, 0, (void*)nullptr, (float*)nullptr, stream);
}

extern "C" int e2k_hc_fwd_norm(const void* Xin, const void* yprev, const float* coef_prev, void* Mout, void* bin,
                               float* coef, const float* static_beta, const float* static_alpha,
                               const float* dyn_alpha_fn, const float* dyn_alpha_scale, const float* dyn_beta_fn,
                               const float* dyn_beta_scale, const float* gamma, int Mtok, int D, int has_depth,
                               const float* norm_gamma, int64_t ldg, float gamma_off, int rows_per_batch, void* xn, float* rn, void* stream) {
    if (!xn) return E2K_ERR_ARG;
    return e2k::dispatch("hc_fwd_norm", hc_fwd_impl, Xin, yprev, coef_prev, Mout, bin, coef, static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma, Mtok, D, has_depth, 1,
                         norm_gamma, ldg, gamma_off, rows_per_batch, xn, rn, stream);
}

extern "C" int e2k_hc_bwd(const void* Xin, const void* yprev, const float* coef_prev, const void* G,
                          const void* dbin, const void* ycur, const float* coef, void* dR, void* dyprev,
                          const float* static_beta, const float* static_alpha, const float* dyn_alpha_fn,
                          const float* dyn_alpha_scale, const float* dyn_beta_fn, const float* dyn_beta_scale,
                          const float* gamma, float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn,
                          float* g_dyn_alpha_scale, float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma,
                          float* partial, int Mtok, int D, int has_depth, int has_width, void* stream) {
    return e2k::dispatch("hc_bwd", hc_bwd_impl, Xin, yprev, coef_prev, G, dbin, ycur, coef, dR, dyprev, static_beta, static_alpha, dyn_alpha_fn, dyn_alpha_scale, dyn_beta_fn, dyn_beta_scale, gamma, g_static_beta, g_static_alpha, g_dyn_alpha_fn, g_dyn_alpha_scale, g_dyn_beta_fn, g_dyn_beta_scale, g_gamma, partial, Mtok, D, has_depth, has_width, stream);
}

extern "C" int e2k_hc_bwd_reduce(const float* partial, const float* dyn_alpha_fn, const float* dyn_beta_fn, const float* gamma,
                                 float* g_static_beta, float* g_static_alpha, float* g_dyn_alpha_fn, float* g_dyn_alpha_scale,
                                 float* g_dyn_beta_fn, float* g_dyn_beta_scale, float* g_gamma, int Mtok, int D, void* stream) {
    return e2k::dispatch("hc_bwd_reduce", hc_bwd_reduce_impl, partial, dyn_alpha_fn, dyn_beta_fn, gamma, g_static_beta, g_static_alpha, g_dyn_alpha_fn, g_dyn_alpha_scale, g_dyn_beta_fn, g_dyn_beta_scale, g_gamma, Mtok, D, stream);
}
