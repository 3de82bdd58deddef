// Attention across the frequency tokens of one frame (Transformer(has_freq_axis = True), e2_tts.py:653-656,920-932): the
// reference rearranges '(b f) n d -> (b n) f d' and runs a default-keyword x-transformers Attention over the f (2 .. 8)
// tokens -- no mask, no dropout, no soft-clamp, no head gates, rotary over the token index, value residual mixed at 0.5.
// The sequence is far too short for the MFMA ring kernels of attn.hip and the rearrangement is pure data movement, so:
// one wave per (batch row, frame, head), lane = channel (dim_head = 64), the f x f scores by wave reductions, and the f
// tokens of a frame are addressed where they lie (row (b f + j) N + n of the token-major projection output): no transposes.
// HBM-bound: 3 I read + I written per token forward, (I + 3 I) read + 3 I written backward (bf16), + the fp32 value-residual
// accumulator on the layers after the first.
#include "e2k_device.h"
#include "plan.h"
#include "../../include/e2k.h"

using namespace e2k;

namespace {

constexpr int FDH = 64;

struct FreqArgs {
    const bf16_t* qkv; long ld;          // (B F N, ld): [q (H 64) | k | v]
    const bf16_t* vfirst; long ldv;      // first layer's v columns (same token order), or null
    const float* cosb; const float* sinb;    // (F, 32)
    bf16_t* out;                         // (B F N, H 64)
    int B, N, H;
    // backward
    const bf16_t* dout; bf16_t* dqkv; long lddq; float* dvfirst; int first_layer;
};

template <int F>
__device__ __forceinline__ void freq_load(const FreqArgs& p, int b, int n, int h, int lane, float (&q)[F], float (&k)[F], float (&v)[F],
                                          float (&cs)[F], float (&sn)[F]) {
    const int I = p.H * FDH;
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const long row = ((long)b * F + j) * p.N + n;
        const bf16_t* r = p.qkv + row * p.ld + h * FDH + lane;
        float qv = bf2f(r[0]), kv = bf2f(r[I]);
        v[j] = bf2f(r[2 * I]);
        if (p.vfirst) v[j] = 0.5f * (v[j] + bf2f(p.vfirst[row * p.ldv + h * FDH + lane]));       // value_residual.lerp(v, 0.5)
        // rotary over the token index j: interleaved pairs (2i, 2i + 1) share theta_i
        cs[j] = p.cosb[j * 32 + (lane >> 1)];
        sn[j] = p.sinb[j * 32 + (lane >> 1)];
        const float qp = __shfl_xor(qv, 1), kp = __shfl_xor(kv, 1);
        const float sg = (lane & 1) ? sn[j] : -sn[j];
        q[j] = fmaf(qp, sg, qv * cs[j]);
        k[j] = fmaf(kp, sg, kv * cs[j]);
    }
}

// P[i][j] = softmax_j(q_i . k_j / 8), identical in every lane
template <int F>
__device__ __forceinline__ void freq_probs(const float (&q)[F], const float (&k)[F], float (&P)[F][F]) {
#pragma unroll
    for (int i = 0; i < F; ++i) {
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            P[i][j] = wave_sum(q[i] * k[j]) * 0.125f;
            m = fmaxf(m, P[i][j]);
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) { P[i][j] = __expf(P[i][j] - m); s += P[i][j]; }
        const float inv = 1.f / s;
#pragma unroll
        for (int j = 0; j < F; ++j) P[i][j] *= inv;
    }
}

template <int F>
__global__ __launch_bounds__(256) void freq_attn_fwd_kernel(FreqArgs p) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)p.B * p.N * p.H;
    if (w >= total) return;
    const int h = (int)(w % p.H);
    const long bn = w / p.H;
    const int n = (int)(bn % p.N), b = (int)(bn / p.N);
    float q[F], k[F], v[F], cs[F], sn[F], P[F][F];
    freq_load<F>(p, b, n, h, lane, q, k, v, cs, sn);
    freq_probs<F>(q, k, P);
#pragma unroll
    for (int i = 0; i < F; ++i) {
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) o = fmaf(P[i][j], v[j], o);
        p.out[(((long)b * F + i) * p.N + n) * ((long)p.H * FDH) + h * FDH + lane] = f2bf(o);
    }
}

template <int F>
__global__ __launch_bounds__(256) void freq_attn_bwd_kernel(FreqArgs p) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)p.B * p.N * p.H;
    if (w >= total) return;
    const int h = (int)(w % p.H);
    const long bn = w / p.H;
    const int n = (int)(bn % p.N), b = (int)(bn / p.N);
    const int I = p.H * FDH;
    float q[F], k[F], v[F], cs[F], sn[F], P[F][F];
    freq_load<F>(p, b, n, h, lane, q, k, v, cs, sn);
    freq_probs<F>(q, k, P);
    float dO[F], dq[F], dk[F], dv[F];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        dO[i] = bf2f(p.dout[(((long)b * F + i) * p.N + n) * (long)I + h * FDH + lane]);
        dq[i] = 0.f; dk[i] = 0.f; dv[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < F; ++i) {
        float dP[F], dot = 0.f;
#pragma unroll
        for (int j = 0; j < F; ++j) {
            dv[j] = fmaf(P[i][j], dO[i], dv[j]);
            dP[j] = wave_sum(dO[i] * v[j]);
            dot = fmaf(P[i][j], dP[j], dot);
        }
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const float ds = P[i][j] * (dP[j] - dot) * 0.125f;
            dq[i] = fmaf(ds, k[j], dq[i]);
            dk[j] = fmaf(ds, q[i], dk[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < F; ++j) {
        const long row = ((long)b * F + j) * p.N + n;
        // inverse rotation (transpose of the forward's): d x[2i] = d y[2i] c + d y[2i+1] s ;  d x[2i+1] = d y[2i+1] c - d y[2i] s
        const float qp = __shfl_xor(dq[j], 1), kp = __shfl_xor(dk[j], 1);
        const float sg = (lane & 1) ? -sn[j] : sn[j];
        const float dqr = fmaf(qp, sg, dq[j] * cs[j]), dkr = fmaf(kp, sg, dk[j] * cs[j]);
        float dvr = dv[j];
        const long vo = row * (long)I + h * FDH + lane;
        if (p.vfirst) {                       // v' = (v + v_first) / 2: half to this layer's projection, half to the first layer's values
            dvr *= 0.5f;
            p.dvfirst[vo] += dvr;
        } else if (p.first_layer && p.dvfirst) {
            dvr += p.dvfirst[vo];
        }
        bf16_t* d = p.dqkv + row * p.lddq + h * FDH + lane;
        d[0] = f2bf(dqr);
        d[I] = f2bf(dkr);
        d[2 * I] = f2bf(dvr);
    }
}

template <int F> void launch_fwd(const FreqArgs& a, hipStream_t st) {
    const long waves = (long)a.B * a.N * a.H;
    hipLaunchKernelGGL(freq_attn_fwd_kernel<F>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}
template <int F> void launch_bwd(const FreqArgs& a, hipStream_t st) {
    const long waves = (long)a.B * a.N * a.H;
    hipLaunchKernelGGL(freq_attn_bwd_kernel<F>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}

}  // namespace

#define FREQ_DISPATCH(fn, F, a, st)                                                                          \
    switch (F) {                                                                                             \
        case 1: fn<1>(a, st); break; case 2: fn<2>(a, st); break; case 3: fn<3>(a, st); break;              \
        case 4: fn<4>(a, st); break; case 5: fn<5>(a, st); break; case 6: fn<6>(a, st); break;              \
        case 7: fn<7>(a, st); break; case 8: fn<8>(a, st); break; default: return E2K_ERR_SHAPE;            \
    }

static int freq_attn_fwd_impl(const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb, const float* sinb, void* out,
                              int B, int F, int N, int H, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    if (F < 1 || F > 8) return E2K_ERR_SHAPE;
    if (!qkv || !cosb || !sinb || !out) return E2K_ERR_ARG;
    FreqArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.vfirst = (const bf16_t*)vfirst; a.ldv = ldv; a.cosb = cosb; a.sinb = sinb;
    a.out = (bf16_t*)out; a.B = B; a.N = N; a.H = H;
    FREQ_DISPATCH(launch_fwd, F, a, (hipStream_t)stream)
    E2K_CHECK_LAUNCH();
    return 0;
}

static int freq_attn_bwd_impl(const void* dout, const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb,
                              const float* sinb, float* dvfirst, int first_layer, void* dqkv, int64_t lddq, int B, int F, int N, int H,
                              void* stream) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    if (F < 1 || F > 8) return E2K_ERR_SHAPE;
    if (!dout || !qkv || !cosb || !sinb || !dqkv || (vfirst && !dvfirst)) return E2K_ERR_ARG;
    FreqArgs a{};
    a.qkv = (const bf16_t*)qkv; a.ld = ld; a.vfirst = (const bf16_t*)vfirst; a.ldv = ldv; a.cosb = cosb; a.sinb = sinb;
    a.B = B; a.N = N; a.H = H;
    a.dout = (const bf16_t*)dout; a.dqkv = (bf16_t*)dqkv; a.lddq = lddq; a.dvfirst = dvfirst; a.first_layer = first_layer;
    FREQ_DISPATCH(launch_bwd, F, a, (hipStream_t)stream)
    E2K_CHECK_LAUNCH();
    return 0;
}

extern "C" int e2k_freq_attn_fwd(const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb, const float* sinb,
                                 void* out, int B, int F, int N, int H, void* stream) {
    return e2k::dispatch("freq_attn_fwd", freq_attn_fwd_impl, qkv, ld, vfirst, ldv, cosb, sinb, out, B, F, N, H, stream);
}

extern "C" int e2k_freq_attn_bwd(const void* dout, const void* qkv, int64_t ld, const void* vfirst, int64_t ldv, const float* cosb,
                                 const float* sinb, float* dvfirst, int first_layer, void* dqkv, int64_t lddq, int B, int F, int N,
                                 int H, void* stream) {
    return e2k::dispatch("freq_attn_bwd", freq_attn_bwd_impl, dout, qkv, ld, vfirst, ldv, cosb, sinb, dvfirst, first_layer, dqkv, lddq,
                         B, F, N, H, stream);
}
