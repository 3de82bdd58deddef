// HBM-bound row / element kernels of the transformer block (bf16 activations, fp32 statistics and parameters):
//   RMSNorm / AdaptiveRMSNorm        x_transformers.RMSNorm, AdaptiveRMSNorm  (e2_tts.py:615,637,645,688,691,729)
//   AdaLN-Zero gate backward         AdaLNZero                                (e2_tts.py:332-351)
//   GEGLU                            x_transformers.FeedForward(glu=True)     (e2_tts.py:646,692)
//   depthwise conv k=31 + SiLU       DepthwiseConv                            (e2_tts.py:295-328)
//   column sums (bias gradients), fp32 -> bf16 parameter shadow casts.
#include "e2k_device.h"
#include "plan.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"

using namespace e2k;

namespace {

// ------------------------------------------------------------------------------------------------ RMSNorm

struct NormArgs {
    const bf16_t* x; const float* gamma; long ldg; float gamma_off; int rows_per_batch;
    bf16_t* y; float* rn; int M;
    // backward
    const bf16_t* dy; bf16_t* dx; float* dgamma;
};

template <int VEC, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(NormArgs p) {
    constexpr int EPL = VEC * NCH, D = 64 * EPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sqrtD = sqrtf((float)D);
    for (int row = blockIdx.x * 4 + wave; row < p.M; row += gridDim.x * 4) {
        float x[EPL], g[EPL];
        load_row<VEC, NCH>(p.x + (long)row * D, lane, x);
        load_row_f32<VEC, NCH>(p.gamma + (long)(row / p.rows_per_batch) * p.ldg, lane, g);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss = fmaf(x[e], x[e], ss);
        ss = wave_sum(ss);
        const float rn = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        const float sc = rn * sqrtD;
#pragma unroll
        for (int e = 0; e < EPL; ++e) x[e] = x[e] * sc * (g[e] + p.gamma_off);
        store_row<VEC, NCH>(p.y + (long)row * D, lane, x);
        if (lane == 0) p.rn[row] = rn;
    }
}

// grid (blocks_per_batch, nb): every block stays inside one batch row-group so d(gamma) reduces in registers
// RB_WAVES waves per workgroup, each with one row in flight ahead of the one it works on: the bytes in flight are what bounds these
// kernels (4 waves x 256 workgroups x 4 KB = 4 MB, about 2 us of latency: 2-3 TB/s), and more workgroups mean more tail atomics
constexpr int RB_WAVES = 8;
constexpr int RMS_BWD_WGS = 256;
template <int VEC, int NCH>
__global__ __launch_bounds__(64 * RB_WAVES) void rmsnorm_bwd_kernel(NormArgs p) {
    constexpr int EPL = VEC * NCH, D = 64 * EPL;
    __shared__ float red[RB_WAVES][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int row0 = b * p.rows_per_batch;
    const int nrows = min(p.rows_per_batch, p.M - row0);
    const float sqrtD = sqrtf((float)D);
    float g[EPL], dg[EPL];
    load_row_f32<VEC, NCH>(p.gamma + (long)b * p.ldg, lane, g);
#pragma unroll
    for (int e = 0; e < EPL; ++e) { g[e] += p.gamma_off; dg[e] = 0.f; }
    // the next row of this wave is fetched (packed bf16 pairs) while the current one is processed
    unsigned rx[EPL / 2], rdy[EPL / 2];
    float rnn = 0.f;
    const int step = gridDim.x * RB_WAVES;
    int i = blockIdx.x * RB_WAVES + wave;
    auto prefetch = [&](int ii) {
        const long row = row0 + min(ii, nrows - 1);
        load_raw_row<VEC, NCH>(p.x + row * D, lane, rx);
        load_raw_row<VEC, NCH>(p.dy + row * D, lane, rdy);
        rnn = p.rn[row];
    };
    if (i < nrows) prefetch(i);
    for (; i < nrows; i += step) {
        const long row = row0 + i;
        float x[EPL], dy[EPL];
        unpack_raw_row<EPL>(rx, x);
        unpack_raw_row<EPL>(rdy, dy);
        const float rn = rnn;
        prefetch(i + step);
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            x[e] *= rn;                       // u
            dg[e] = fmaf(dy[e] * sqrtD, x[e], dg[e]);
            dy[e] *= g[e];                    // t
            dot = fmaf(dy[e], x[e], dot);
        }
        dot = wave_sum_fast(dot);
        const float sc = rn * sqrtD;
#pragma unroll
        for (int e = 0; e < EPL; ++e) dy[e] = sc * (dy[e] - x[e] * dot);
        store_row<VEC, NCH>(p.dx + row * D, lane, dy);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[wave][c * 64 * VEC + lane * VEC + v] = dg[c * VEC + v];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 64 * RB_WAVES) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < RB_WAVES; ++w) sum += red[w][d];
        atomicAdd(p.dgamma + (long)b * p.ldg + d, sum);
    }
}

template <int VEC, int NCH> int launch_norm_fwd(const NormArgs& a, hipStream_t st) {
    int grid = (a.M + 3) / 4; if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL((rmsnorm_fwd_kernel<VEC, NCH>), dim3(grid), dim3(256), 0, st, a);
    return 0;
}
template <int VEC, int NCH> int launch_norm_bwd(const NormArgs& a, hipStream_t st) {
    const int nb = (a.M + a.rows_per_batch - 1) / a.rows_per_batch;
    // ~256 workgroups in all: every workgroup ends with D atomic adds into d(gamma), and with one gamma row (plain
    // RMSNorm) all of them hit the same D addresses
    int per = (min(a.rows_per_batch, a.M) + RB_WAVES - 1) / RB_WAVES;
    int cap = (RMS_BWD_WGS + nb - 1) / nb;  // (256 against 512 / 1024 workgroups: 1.64 / 1.78 / 2.63 ms per cfg3 step, profiles/r04_row_grid_ab.txt)
    if (per > cap) per = cap;
    hipLaunchKernelGGL((rmsnorm_bwd_kernel<VEC, NCH>), dim3(per, nb), dim3(64 * RB_WAVES), 0, st, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------ AdaLN-Zero gate backward
//   forward (fused in the GEMM epilogue):  y = ao * g[b]            (g = sigmoid(to_gamma(c)), per batch row)
//   backward: dao = dy * g[b] ;  gsum[b][d] += sum_rows dy * y      (so that dg = gsum / g, d(pre-sigmoid) = gsum * (1 - g))

struct GateArgs { const bf16_t* dy; const bf16_t* y; const float* g; bf16_t* dao; float* gsum; long ldg; int M, rows_per_batch; };

template <int VEC, int NCH>
__global__ __launch_bounds__(64 * RB_WAVES) void gate_bwd_kernel(GateArgs p) {
    constexpr int EPL = VEC * NCH, D = 64 * EPL;
    __shared__ float red[RB_WAVES][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int row0 = b * p.rows_per_batch;
    const int nrows = min(p.rows_per_batch, p.M - row0);
    float g[EPL], acc[EPL];
    load_row_f32<VEC, NCH>(p.g + (long)b * p.ldg, lane, g);
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    // the next row of this wave is fetched (packed bf16 pairs) while the current one is processed
    unsigned ry[EPL / 2], rdy[EPL / 2];
    const int step = gridDim.x * RB_WAVES;
    int i = blockIdx.x * RB_WAVES + wave;
    auto prefetch = [&](int ii) {
        const long row = row0 + min(ii, nrows - 1);
        load_raw_row<VEC, NCH>(p.y + row * D, lane, ry);
        load_raw_row<VEC, NCH>(p.dy + row * D, lane, rdy);
    };
    if (i < nrows) prefetch(i);
    for (; i < nrows; i += step) {
        const long row = row0 + i;
        float y[EPL], dy[EPL];
        unpack_raw_row<EPL>(ry, y);
        unpack_raw_row<EPL>(rdy, dy);
        prefetch(i + step);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { acc[e] = fmaf(dy[e], y[e], acc[e]); dy[e] *= g[e]; }
        store_row<VEC, NCH>(p.dao + row * D, lane, dy);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[wave][c * 64 * VEC + lane * VEC + v] = acc[c * VEC + v];
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 64 * RB_WAVES) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < RB_WAVES; ++w) sum += red[w][d];
        atomicAdd(p.gsum + (long)b * p.ldg + d, sum);
    }
}
constexpr int GATE_WGS = 2048 / RB_WAVES;
template <int VEC, int NCH> int launch_gate_bwd(const GateArgs& a, hipStream_t st) {
    const int nb = (a.M + a.rows_per_batch - 1) / a.rows_per_batch;
    int per = (min(a.rows_per_batch, a.M) + RB_WAVES - 1) / RB_WAVES;
    int cap = (GATE_WGS + nb - 1) / nb;     // (512 workgroups of 4 waves + the row prefetch: 0.72 ms per cfg3 step against 0.97 with 1024 and none)
    if (per > cap) per = cap;
    hipLaunchKernelGGL((gate_bwd_kernel<VEC, NCH>), dim3(per, nb), dim3(64 * RB_WAVES), 0, st, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------ GEGLU

// (gelu_erf, gelu_erf_grad, keep_scale: e2k_device.h -- shared with the GEGLU epilogue of the NT GEMM)

struct GegluArgs {
    const bf16_t* H; long ldh; bf16_t* out; const bf16_t* dout; bf16_t* dH; int M, F;
    unsigned seed, stream, thresh; float inv_keep;     // dropout (thresh = 0: off)
    const unsigned* seed_dev;                          // if set, the seed is read from device memory (graph replay)
};

template <bool BWD>
__global__ __launch_bounds__(256) void geglu_kernel(GegluArgs p) {
    const int fv = p.F / 8;
    const long total = (long)p.M * fv;
    const unsigned seed = p.seed_dev ? *p.seed_dev : p.seed;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / fv), c = (int)(i % fv) * 8;
        float u[8], gt[8];
        unpack8(ld<u32x4>(p.H + (long)m * p.ldh + c), u);
        unpack8(ld<u32x4>(p.H + (long)m * p.ldh + p.F + c), gt);
        float ks[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ks[e] = p.thresh ? keep_scale(seed, p.stream, m, c + e, p.thresh, p.inv_keep) : 1.f;
        if (!BWD) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = u[e] * gelu_erf(gt[e]) * ks[e];
            st<u32x4>(p.out + (long)m * p.F + c, pack8(o));
        } else {
            float d[8], du[8], dg[8];
            unpack8(ld<u32x4>(p.dout + (long)m * p.F + c), d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float dd = d[e] * ks[e];
                du[e] = dd * gelu_erf(gt[e]);
                dg[e] = dd * u[e] * gelu_erf_grad(gt[e]);
            }
            st<u32x4>(p.dH + (long)m * p.ldh + c, pack8(du));
            st<u32x4>(p.dH + (long)m * p.ldh + p.F + c, pack8(dg));
        }
    }
}

// ------------------------------------------------------------------------------------------------ column sums

__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* x, long ldx, float* out, int M, int N) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (col >= N) return;
    float a0 = 0.f, a1 = 0.f;
    for (int m = blockIdx.y; m < M; m += gridDim.y) {
        unsigned v = ld<unsigned>(x + (long)m * ldx + col);
        a0 += bflo(v);
        a1 += bfhi(v);
    }
    atomicAdd(out + col, a0);
    if (col + 1 < N) atomicAdd(out + col + 1, a1);
}

// ------------------------------------------------------------------------------------------------ LinearFourierEmbed

// LinearFourierEmbed's activation (e2_tts.py:368-386, `attn_fourier_embed_input`): h (M, nf + nrest) = linear(x) ->
// y (M, 2 nf + nrest) = [sin h[:nf] | cos h[:nf] | h[nf:]].  One thread per 8 output columns (nf, nrest multiples of 8).
// h stays fp32 (the projection GEMM's fp32 output): rounding an angle of a few radians to bf16 moves its sine by percents.
__global__ __launch_bounds__(256) void fourier_cat_fwd_kernel(const float* h, long ldh, bf16_t* y, long ldy, long M, int nf, int nrest) {
    const int cv = (2 * nf + nrest) / 8;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cv;
        const int c = (int)(i - m * cv) * 8;
        const int src = c < nf ? c : (c < 2 * nf ? c - nf : c - nf);
        float f[8];
        {
            const f32x4 a = ld<f32x4>(h + m * ldh + src), b = ld<f32x4>(h + m * ldh + src + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { f[k] = a[k]; f[4 + k] = b[k]; }
        }
        if (c < nf) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = sinf(f[k]);
        } else if (c < 2 * nf) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = cosf(f[k]);
        }
        st<u32x4>(y + m * ldy + c, pack8(f));
    }
}

// dh[:nf] = dy_sin cos h - dy_cos sin h ;  dh[nf:] = dy[2 nf:]
__global__ __launch_bounds__(256) void fourier_cat_bwd_kernel(const bf16_t* dy, long ldy, const float* h, long ldh, bf16_t* dh, long lddh,
                                                              long M, int nf, int nrest) {
    const int cv = (nf + nrest) / 8;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cv;
        const int c = (int)(i - m * cv) * 8;
        float f[8];
        if (c < nf) {
            float hv[8], ds[8], dc[8];
            {
                const f32x4 a = ld<f32x4>(h + m * ldh + c), b = ld<f32x4>(h + m * ldh + c + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { hv[k] = a[k]; hv[4 + k] = b[k]; }
            }
            unpack8(ld<u32x4>(dy + m * ldy + c), ds);
            unpack8(ld<u32x4>(dy + m * ldy + nf + c), dc);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = ds[k] * cosf(hv[k]) - dc[k] * sinf(hv[k]);
        } else {
            unpack8(ld<u32x4>(dy + m * ldy + nf + c), f);
        }
        st<u32x4>(dh + m * lddh + c, pack8(f));
    }
}

// ------------------------------------------------------------------------------------------------ casts

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* src, bf16_t* dst, long n) {
    const long nv = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        f32x4 a = ld<f32x4>(src + i * 8), b = ld<f32x4>(src + i * 8 + 4);
        float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st<u32x4>(dst + i * 8, pack8(f));
    }
    if (blockIdx.x == 0) {
        for (long i = nv * 8 + threadIdx.x; i < n; i += 256) dst[i] = f2bf(src[i]);
    }
}

// src (R, C) fp32 row-major -> dst (C, R) bf16 row-major (ldd elements per dst row)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* src, bf16_t* dst, int R, int C, long ldd) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * ldd + r] = f2bf(tile[tx][i]);
    }
}

// All transposed bf16 shadows of a module in ONE launch (the per-matrix kernel above: 228 launches, 4.3 ms per dim-1024 /
// depth-24 step -- 2-byte stores and a tail of tiny grids).  desc[i] = {src offset, dst offset, R, C, ldd, first block}
// (int64, element offsets into the flat fp32 parameter buffer / the flat transposed-shadow buffer); a block finds its
// matrix by binary search over `first block`, transposes one 64 x 64 tile through LDS and writes 32 contiguous bytes
// per thread.
__global__ __launch_bounds__(256) void cast_transpose_batch_kernel(const float* flat, bf16_t* flatT, const long* desc, int n) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n - 1;
    const long blk = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid * 6 + 5] <= blk) lo = mid;
        else hi = mid - 1;
    }
    const long* d = desc + lo * 6;
    const float* src = flat + d[0];
    bf16_t* dst = flatT + d[1];
    const int R = (int)d[2], C = (int)d[3];
    const long ldd = d[4];
    const int local = (int)(blk - d[5]), tiles_c = (C + 63) / 64;
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
    const int tid = threadIdx.x;
    {
        const int tx = tid & 15, ty = tid >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < R) {
                const float* q = src + (long)r * C + c;
                if (c + 3 < C && ((C & 3) == 0) && (((uintptr_t)q & 15) == 0)) v = ld<f32x4>(q);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < C) v[e] = q[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[ty + 16 * k][4 * tx + e] = v[e];
        }
    }
    __syncthreads();
    const int c = c0 + (tid >> 2), rb = (tid & 3) * 16;
    if (c >= C) return;
    bf16_t* o = dst + (long)c * ldd + r0 + rb;
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = tile[rb + j][tid >> 2];
    if (r0 + rb + 15 < R && (((uintptr_t)o & 15) == 0)) {
        st<u32x4>(o, pack8(f));
        st<u32x4>(o + 8, pack8(f + 8));
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r0 + rb + j < R) o[j] = f2bf(f[j]);
    }
}

// ------------------------------------------------------------------------------------------------ depthwise conv + SiLU
// channels-last: x (B, N, C); forward block = 128 frames x 64 channels of one batch row, thread = 2 channels x 16 frames;
// backward block = 64 frames x 64 channels per tile, several tiles per workgroup, thread = 2 channels x 8 frames.

typedef float f32x2_ __attribute__((ext_vector_type(2)));
constexpr int CTN = 64, CTC = 64;

struct ConvArgs {
    const bf16_t* x; const uint8_t* mask; const float* w; const float* bias;
    bf16_t* pre; bf16_t* y;
    const bf16_t* dy; bf16_t* dx; float* dw; float* dbias;
    float* ws;          // backward: per-workgroup (dw, dbias) partials [b][channel tile][x][64][ks + 1], or NULL (atomics)
    int B, N, C, tiles_per_block;
    int tune;           // backward tuning / ablation bits (e2k_dwconv_bwd `split` argument >> 1): 1 = no gradient flush, 2 = no arithmetic; >> 7: workgroups per (channel tile, batch)
    int defer_reduce;      // backward: leave the (dw, dbias) partials in ws, e2k_dwconv_bwd_reduce sums them later
};

// Forward: 128 frames x 64 channels per workgroup (the 30-row halo is 23 % of the tile), thread = 2 channels x 16 frames; the
// 64 x KS weights and the frame mask go through LDS once per workgroup.  (Round 3; the first kernel used 64-frame tiles -- halo
// 47 % -- and had every thread fetch its 2 x KS weights itself, 62 loads per thread = 5 x the tile's bytes through the L1, and the
// mask byte of each of its frames from global memory: 31.4 -> 24.4 us at the cfg3 audio shape with a mask, 17.6 -> 15.4 us at the
// text shape, profiles/r03_dwconv_fwd_ab.jsonl.  Still 2.1 TB/s of its algorithmic bytes: 82 VGPRs, five workgroups per CU -- which is
// ALL 4.5 workgroups per CU of a cfg3 call at once, so their load / compute / store phases run in lockstep and do not overlap.  It is not
// the 4-byte stores: the outputs through LDS in 16-byte pieces, a quarter of the store instructions, measured +3 % here and -1 % in
// the backward (round 5, profiles/r05w_dwconv_16byte_stores_ab.txt) and were taken out again.)
template <int KS>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(ConvArgs p) {
    constexpr int FR = 128, PAD = KS / 2, ROWS = FR + KS - 1;
    __shared__ __attribute__((aligned(16))) unsigned xt[ROWS][CTC / 2];
    __shared__ __attribute__((aligned(16))) float wl[KS][CTC];
    __shared__ float bl[CTC];
    __shared__ unsigned char mk[FR];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * FR, c0 = blockIdx.y * CTC, b = blockIdx.z;
    {
        constexpr int ITEMS = ROWS * (CTC / 8), NIT = (ITEMS + 255) / 256;
        u32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int j = i / (CTC / 8), c8 = (i % (CTC / 8)) * 8;
            const int n = n0 - PAD + j;
            const int nc = min(max(n, 0), p.N - 1);
            const u32x4 raw = ld<u32x4>(p.x + ((long)b * p.N + nc) * p.C + c0 + c8);
            const unsigned char m = p.mask ? p.mask[(long)b * p.N + nc] : (unsigned char)1;
            v[it] = (n >= 0 && n < p.N && m != 0 && i < ITEMS) ? raw : u32x4{0u, 0u, 0u, 0u};
        }
        // weights of this channel tile: (64, KS) contiguous in global memory -> wl[k][c]
        for (int i = tid; i < CTC * KS; i += 256) wl[i % KS][i / KS] = p.w[(long)c0 * KS + i];
        if (tid < CTC) bl[tid] = p.bias[c0 + tid];
        if (tid < FR) {
            const int n = n0 + tid;
            mk[tid] = (n < p.N && (p.mask == nullptr || p.mask[(long)b * p.N + n])) ? 1 : 0;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < ITEMS) st<u32x4>(&xt[i / (CTC / 8)][(i % (CTC / 8)) * 4], v[it]);
        }
    }
    __syncthreads();
    const int cp = tid & 31, fg = tid >> 5;
    // input row i meets the taps k = i - o of the 16 outputs: a window of 16 (+ 4) taps, one new tap per row out of LDS (all KS taps in
    // registers were 62 of this kernel's 154 VGPRs)
    constexpr int WIN = 16 + 4;         // + the rows of a chunk whose taps are fetched before its first row is used
    f32x2_ w[WIN];
    const f32x2_ bb = *reinterpret_cast<const f32x2_*>(&bl[cp * 2]);
    f32x2_ a[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) a[o] = bb;
#pragma unroll
    for (int i0 = 0; i0 < 16 + KS - 1; i0 += 4) {       // four rows at a time, pinned: left alone the compiler fetches all taps and rows first and runs one output's chain after the other
        unsigned v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i < KS) w[i % WIN] = *reinterpret_cast<const f32x2_*>(&wl[i][cp * 2]);
            if (i < 16 + KS - 1) v[u] = xt[fg * 16 + i][cp];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u;
            if (i >= 16 + KS - 1) continue;
            const f32x2_ xx = {bflo(v[u]), bfhi(v[u])};
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                const int k = i - o;
                if (k >= 0 && k < KS) a[o] = __builtin_elementwise_fma(w[k % WIN], xx, a[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) pin_vgpr(a[o]);
        order_memory();
    }
    bf16_t* pre = p.pre + ((long)b * p.N + n0 + fg * 16) * p.C + c0 + cp * 2;
    bf16_t* y = p.y + ((long)b * p.N + n0 + fg * 16) * p.C + c0 + cp * 2;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        if (n0 + fg * 16 + o < p.N) {
            if (p.pre) st<unsigned>(pre + (long)o * p.C, pack2bf(a[o][0], a[o][1]));       // (null: a no-grad forward, nothing reads the pre-activation)
            st<unsigned>(y + (long)o * p.C, mk[fg * 16 + o] ? pack2bf(siluf_(a[o][0]), siluf_(a[o][1])) : 0u);
        }
    }
}

__device__ __forceinline__ float silu_grad(float x) { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }

// Backward, round 4: d(pre-activation) tile in fp32, the masked input as the bf16 pairs it arrives in (row stride 36 words: the two
// frame groups of a wave land in different bank halves), the flipped taps in LDS with a window of 8 in registers (as the forward).
// All KS taps + all KS tap gradients in registers were 226 VGPRs = two workgroups per CU, whose load and arithmetic phases did not
// overlap (1.0 TB/s of the kernel's bytes); now the tap gradients are the only long-lived registers.
constexpr int XTS = CTC / 2 + 4;
template <int KS>
__global__ __launch_bounds__(256, 3) void dwconv_bwd_kernel(ConvArgs p) {
    constexpr int PAD = KS / 2, ROWS = CTN + KS - 1;
    __shared__ __attribute__((aligned(16))) float smem[ROWS * CTC + ROWS * XTS];
    __shared__ __attribute__((aligned(16))) float wl[KS][CTC];      // flipped: wl[k][c] = w[c][KS - 1 - k]
    float (*dpt)[CTC] = reinterpret_cast<float (*)[CTC]>(smem);                         // d(pre-activation), frame n0 - PAD + j
    unsigned (*xt)[XTS] = reinterpret_cast<unsigned (*)[XTS]>(smem + ROWS * CTC);      // masked input,      frame n0 - PAD + j
    static_assert(4 * (KS + 1) * CTC <= ROWS * CTC + ROWS * XTS, "the gradient staging reuses the tile buffers");
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * CTC, b = blockIdx.z;
    const int cp = tid & 31, fg = tid >> 5;
    const int ch = c0 + cp * 2;
    for (int i = tid; i < CTC * KS; i += 256) wl[KS - 1 - i % KS][i / KS] = p.w[(long)c0 * KS + i];
    // channel pair (ch, ch+1) lives in the two halves of 64-bit register pairs: every FMA below is one v_pk_fma_f32
    // weight / bias gradient partials stay in registers over all frame tiles of this workgroup (p.tiles_per_block):
    // one LDS + global flush per workgroup instead of per tile
    f32x2_ gw[KS], sb = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) gw[k] = f32x2_{0.f, 0.f};
    const int ntiles = (p.N + CTN - 1) / CTN;
    const int t_beg = blockIdx.x * p.tiles_per_block, t_end = min(ntiles, t_beg + p.tiles_per_block);
    for (int tile = t_beg; tile < t_end; ++tile) {
        const int n0 = tile * CTN;
        __syncthreads();                  // previous tile fully consumed
        // 8 channels (16 B) per item; the loads of two items (6 x 16 B) are in flight together
        constexpr int ITEMS = ROWS * (CTC / 8), NIT = (ITEMS + 255) / 256;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += 2) {
            u32x4 rd[2], rp[2], rx[2];
            bool live[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + (it0 + u) * 256;
                const int j = i / (CTC / 8), c8 = (i % (CTC / 8)) * 8;
                const int n = n0 - PAD + j;
                // unconditional loads from a clamped row, mask byte fetched alongside (a load that depends on the mask
                // byte costs a second memory latency per item)
                const int nc = min(max(n, 0), p.N - 1);
                const long off = ((long)b * p.N + nc) * p.C + c0 + (i < ITEMS ? c8 : 0);
                rd[u] = ld<u32x4>(p.dy + off);
                rp[u] = ld<u32x4>(p.pre + off);
                rx[u] = ld<u32x4>(p.x + off);
                const unsigned char mk = p.mask ? p.mask[(long)b * p.N + nc] : (unsigned char)1;
                live[u] = it0 + u < NIT && i < ITEMS && n >= 0 && n < p.N && mk != 0;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = tid + (it0 + u) * 256;
                if (it0 + u >= NIT || i >= ITEMS) continue;
                const int j = i / (CTC / 8), c8 = (i % (CTC / 8)) * 8;
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] = 0.f;
                if (live[u]) {
                    float pr[8];
                    unpack8(rd[u], d);
                    unpack8(rp[u], pr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] *= silu_grad(pr[e]);
                }
                st<f32x4>(&dpt[j][c8], f32x4{d[0], d[1], d[2], d[3]});
                st<f32x4>(&dpt[j][c8 + 4], f32x4{d[4], d[5], d[6], d[7]});
                st<u32x4>(&xt[j][c8 / 2], live[u] ? rx[u] : u32x4{0u, 0u, 0u, 0u});
            }
        }
        __syncthreads();
        if (p.tune & 2) continue;
        // dx[o] = sum_k' wflip[k'] dp_tile[o + k']: row i meets the taps k' = i - o of the 8 outputs
        constexpr int WIN = 8 + 4;
        f32x2_ a[8], ww[WIN];
#pragma unroll
        for (int o = 0; o < 8; ++o) a[o] = f32x2_{0.f, 0.f};
#pragma unroll
        for (int i0 = 0; i0 < 8 + KS - 1; i0 += 4) {        // four rows at a time, pinned (as the forward)
            f32x2_ dd[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i < KS) ww[i % WIN] = ld<f32x2_>(&wl[i][cp * 2]);
                if (i < 8 + KS - 1) dd[u] = ld<f32x2_>(&dpt[fg * 8 + i][cp * 2]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u;
                if (i >= 8 + KS - 1) continue;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const int k = i - o;
                    if (k >= 0 && k < KS) a[o] = __builtin_elementwise_fma(ww[k % WIN], dd[u], a[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < 8; ++o) pin_vgpr(a[o]);
            order_memory();
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int n = n0 + fg * 8 + o;
            if (n < p.N) {
                const bool keep = p.mask == nullptr || p.mask[(long)b * p.N + n];
                st<unsigned>(p.dx + ((long)b * p.N + n) * p.C + ch, keep ? pack2bf(a[o][0], a[o][1]) : 0u);
            }
        }
        // dw[k] += sum_o dp_tile[o + PAD] * x_tile[o + k]
        f32x2_ dq[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            dq[o] = ld<f32x2_>(&dpt[fg * 8 + o + PAD][cp * 2]);
            sb += dq[o];
        }
#pragma unroll
        for (int i = 0; i < 8 + KS - 1; ++i) {
            const unsigned xv = xt[fg * 8 + i][cp];
            const f32x2_ xx = {bflo(xv), bfhi(xv)};
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const int k = i - o;
                if (k >= 0 && k < KS) gw[k] = __builtin_elementwise_fma(dq[o], xx, gw[k]);
            }
        }
    }
    __syncthreads();
    if (p.tune & 1) return;
    // (dw, dbias) of the workgroup: the two frame groups of a wave (lanes l, l + 32: same channel pair) are added with a
    // shuffle, the four waves through plain LDS stores into the (now free) tile buffers.  LDS float atomics here
    // (8 adds per address) cost 42 of the kernel's 84 us at the cfg3 audio shape (tools/probes/conv_ablate.py).
    float (*stage)[KS + 1][CTC] = reinterpret_cast<float (*)[KS + 1][CTC]>(smem);
    const int wv = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k <= KS; ++k) {
        const f32x2_ g = k < KS ? gw[k < KS ? k : 0] : sb;
        const float s0 = g[0] + __shfl_xor(g[0], 32), s1 = g[1] + __shfl_xor(g[1], 32);
        if (lane < 32) st<f32x2_>(&stage[wv][k][cp * 2], f32x2_{s0, s1});
    }
    __syncthreads();
    // 512 workgroups x 2080 fp32 atomics on 32-fold shared addresses were more than half of this kernel's time (45 of 84 us
    // at the cfg3 audio shape, tools/probes/conv_ablate.py): the partials go to a workspace, conv_reduce_kernel adds them up
    float* wsb = p.ws ? p.ws + (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (CTC * (KS + 1)) : nullptr;
    for (int i = tid; i < CTC * (KS + 1); i += 256) {
        int k = i / CTC, c = i % CTC;
        float v = (stage[0][k][c] + stage[1][k][c]) + (stage[2][k][c] + stage[3][k][c]);
        if (wsb) wsb[i] = v;
        else if (k < KS) atomicAdd(p.dw + (long)(c0 + c) * KS + k, v);
        else atomicAdd(p.dbias + c0 + c, v);
    }
}

// dw[c][k] += sum over (batch, x) of the workgroup partials; grid (C / 64, 64 (ks + 1) / 64) of ONE wave each: 512 single-wave
// workgroups spread over the chip (128 workgroups of 256 threads with one running sum: 12-17 us for 4-6 MB of partials), four
// independent sums per thread in a fixed order
__global__ __launch_bounds__(64) void conv_reduce_kernel(const float* ws, float* dw, float* dbias, int B, int nct, int gx, int KS) {
    const int E = CTC * (KS + 1);
    const int ct = blockIdx.x, i = blockIdx.y * 64 + threadIdx.x;
    if (i >= E) return;
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < B; ++b) {
        const float* pb = ws + ((long)b * nct + ct) * gx * E + i;
        int x = 0;
        for (; x + 4 <= gx; x += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += pb[(long)(x + u) * E];
        }
        for (; x < gx; ++x) s4[x & 3] += pb[(long)x * E];
    }
    const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    const int k = i / CTC, c = i % CTC;
    if (k < KS) dw[(long)(ct * CTC + c) * KS + k] += s;
    else dbias[ct * CTC + c] += s;
}

constexpr int CONV_BWD_WGS = 768;      // three workgroups per CU
int conv_bwd_gx(int B, int N, int C, int tune) {
    // about 512 workgroups: fewer, longer workgroups cut the traffic of the (dw, dbias) partials
    const int ntiles = (N + CTN - 1) / CTN, cb = (C / CTC) * B;
    int gx = (CONV_BWD_WGS + cb - 1) / cb;
    if (tune >> 7) gx = tune >> 7;
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    const int tpb = (ntiles + gx - 1) / gx;
    return (ntiles + tpb - 1) / tpb;          // workgroups along x actually launched
}

template <int KS> int launch_conv(ConvArgs a, bool bwd, hipStream_t st) {
    const int ntiles = (a.N + CTN - 1) / CTN;
    dim3 grid(ntiles, a.C / CTC, a.B), block(256);
    if (bwd) {
        const int gxl = conv_bwd_gx(a.B, a.N, a.C, a.tune);
        a.tiles_per_block = (ntiles + gxl - 1) / gxl;
        {
            grid.x = (ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
            hipLaunchKernelGGL(dwconv_bwd_kernel<KS>, grid, block, 0, st, a);
            if (a.ws && !(a.tune & 1) && !a.defer_reduce)
                hipLaunchKernelGGL(conv_reduce_kernel, dim3(a.C / CTC, KS + 1), dim3(64), 0, st,
                                   (const float*)a.ws, a.dw, a.dbias, a.B, a.C / CTC, (int)grid.x, KS);
        }
    } else {
        grid.x = (a.N + 127) / 128;
        hipLaunchKernelGGL(dwconv_fwd_kernel<KS>, grid, block, 0, st, a);
    }
    return 0;
}
int dispatch_conv(const ConvArgs& a, int ks, bool bwd, hipStream_t st) {
    switch (ks) {
        case 31: return launch_conv<31>(a, bwd, st);
        case 15: return launch_conv<15>(a, bwd, st);
        case 7: return launch_conv<7>(a, bwd, st);
        case 3: return launch_conv<3>(a, bwd, st);
        default: return E2K_ERR_SHAPE;
    }
}

}  // namespace

static int rmsnorm_fwd_impl(const void* x, const float* gamma, int64_t ldg, float gamma_off, int rows_per_batch,
                               void* y, float* rn, int M, int D, void* stream) {
    if (M <= 0) return 0;
    if (!x || !gamma || !y || !rn || rows_per_batch <= 0) return E2K_ERR_ARG;
    NormArgs a{};
    a.x = (const bf16_t*)x; a.gamma = gamma; a.ldg = ldg; a.gamma_off = gamma_off; a.rows_per_batch = rows_per_batch;
    a.y = (bf16_t*)y; a.rn = rn; a.M = M;
    int rc = 0;
    E2K_ROW_DISPATCH(D, launch_norm_fwd, a, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

static int rmsnorm_bwd_impl(const void* dy, const void* x, const float* rn, const float* gamma, int64_t ldg,
                               float gamma_off, int rows_per_batch, void* dx, float* dgamma, int M, int D,
                               void* stream) {
    if (M <= 0) return 0;
    if (!dy || !x || !rn || !gamma || !dx || !dgamma || rows_per_batch <= 0) return E2K_ERR_ARG;
    NormArgs a{};
    a.x = (const bf16_t*)x; a.gamma = gamma; a.ldg = ldg; a.gamma_off = gamma_off; a.rows_per_batch = rows_per_batch;
    a.rn = const_cast<float*>(rn); a.M = M; a.dy = (const bf16_t*)dy; a.dx = (bf16_t*)dx; a.dgamma = dgamma;
    int rc = 0;
    E2K_ROW_DISPATCH(D, launch_norm_bwd, a, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

static int gate_bwd_impl(const void* dy, const void* y, const float* g, void* dao, float* gsum, int64_t ldg,
                            int M, int D, int rows_per_batch, void* stream) {
    if (M <= 0) return 0;
    if (!dy || !y || !g || !dao || !gsum || rows_per_batch <= 0) return E2K_ERR_ARG;
    GateArgs a{(const bf16_t*)dy, (const bf16_t*)y, g, (bf16_t*)dao, gsum, (long)ldg, M, rows_per_batch};
    int rc = 0;
    E2K_ROW_DISPATCH(D, launch_gate_bwd, a, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

static int geglu_grid(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

static int geglu_fwd_impl(const void* H, int64_t ldh, void* out, int M, int F, float p_drop, uint32_t seed,
                             const uint32_t* seed_dev, uint32_t stream_id, void* stream) {
    if (M <= 0 || F <= 0) return 0;
    if ((F & 7) || (ldh & 7)) return E2K_ERR_ALIGN;
    GegluArgs a{};
    a.H = (const bf16_t*)H; a.ldh = ldh; a.out = (bf16_t*)out; a.M = M; a.F = F;
    a.seed = seed; a.seed_dev = seed_dev; a.stream = stream_id; a.thresh = (unsigned)(p_drop * 65536.f + 0.5f); a.inv_keep = 1.f / (1.f - p_drop);
    hipLaunchKernelGGL(geglu_kernel<false>, dim3(geglu_grid((long)M * F / 8)), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int geglu_bwd_impl(const void* dout, const void* H, int64_t ldh, void* dH, int M, int F, float p_drop,
                             uint32_t seed, const uint32_t* seed_dev, uint32_t stream_id, void* stream) {
    if (M <= 0 || F <= 0) return 0;
    if ((F & 7) || (ldh & 7)) return E2K_ERR_ALIGN;
    GegluArgs a{};
    a.H = (const bf16_t*)H; a.ldh = ldh; a.dout = (const bf16_t*)dout; a.dH = (bf16_t*)dH; a.M = M; a.F = F;
    a.seed = seed; a.seed_dev = seed_dev; a.stream = stream_id; a.thresh = (unsigned)(p_drop * 65536.f + 0.5f); a.inv_keep = 1.f / (1.f - p_drop);
    hipLaunchKernelGGL(geglu_kernel<true>, dim3(geglu_grid((long)M * F / 8)), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int colsum_bf16_impl(const void* x, int64_t ldx, float* out, int M, int N, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    if ((N & 1) || (ldx & 1)) return E2K_ERR_ALIGN;
    int splits = (M + 63) / 64; if (splits > 128) splits = 128;
    hipLaunchKernelGGL(colsum_kernel, dim3((N / 2 + 255) / 256, splits), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (long)ldx, out, M, N);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int fourier_cat_fwd_impl(const float* h, int64_t ldh, void* y, int64_t ldy, int64_t M, int nf, int nrest, void* stream) {
    if (M <= 0) return 0;
    if (nf < 0 || nrest < 0 || (nf & 7) || (nrest & 7) || nf + nrest == 0 || (ldh & 3) || (ldy & 7)) return E2K_ERR_SHAPE;
    if (!h || !y) return E2K_ERR_ARG;
    if (((uintptr_t)h | (uintptr_t)y) & 15) return E2K_ERR_ALIGN;
    long g = (M * ((2 * nf + nrest) / 8) + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(fourier_cat_fwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const float*)h, (long)ldh, (bf16_t*)y,
                       (long)ldy, (long)M, nf, nrest);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int fourier_cat_bwd_impl(const void* dy, int64_t ldy, const float* h, int64_t ldh, void* dh, int64_t lddh, int64_t M, int nf,
                                int nrest, void* stream) {
    if (M <= 0) return 0;
    if (nf < 0 || nrest < 0 || (nf & 7) || (nrest & 7) || nf + nrest == 0 || (ldh & 3) || (ldy & 7) || (lddh & 7)) return E2K_ERR_SHAPE;
    if (!dy || !h || !dh) return E2K_ERR_ARG;
    if (((uintptr_t)dy | (uintptr_t)h | (uintptr_t)dh) & 15) return E2K_ERR_ALIGN;
    long g = (M * ((nf + nrest) / 8) + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(fourier_cat_bwd_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (long)ldy,
                       (const float*)h, (long)ldh, (bf16_t*)dh, (long)lddh, (long)M, nf, nrest);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cast_bf16_impl(const float* src, void* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return E2K_ERR_ALIGN;
    long g = (n / 8 + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cast_transpose_bf16_impl(const float* src, void* dst, int R, int C, int64_t ldd, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       src, (bf16_t*)dst, R, C, (long)ldd);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cast_transpose_batch_impl(const float* flat, void* flatT, const int64_t* desc, int n, int total_blocks, void* stream) {
    if (n <= 0 || total_blocks <= 0) return 0;
    if (!flat || !flatT || !desc) return E2K_ERR_ARG;
    hipLaunchKernelGGL(cast_transpose_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, flat, (bf16_t*)flatT,
                       (const long*)desc, n);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int dwconv_fwd_impl(const void* x, const uint8_t* mask, const float* w, const float* bias, void* pre,
                              void* y, int B, int N, int C, int ks, void* stream) {
    if (B <= 0 || N <= 0) return 0;
    if (C % CTC) return E2K_ERR_SHAPE;
    ConvArgs a{};
    a.x = (const bf16_t*)x; a.mask = mask; a.w = w; a.bias = bias; a.pre = (bf16_t*)pre; a.y = (bf16_t*)y;
    a.B = B; a.N = N; a.C = C;
    int rc = dispatch_conv(a, ks, false, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

static int dwconv_bwd_impl(const void* dy, const void* pre, const void* x, const uint8_t* mask, const float* w,
                              void* dx, float* dw, float* dbias, float* ws, int B, int N, int C, int ks, int split, void* stream) {
    if (B <= 0 || N <= 0) return 0;
    if (C % CTC) return E2K_ERR_SHAPE;
    if ((split & 1) && !ws) return E2K_ERR_ARG;     // bit 0: the caller sums the partials later (e2k_dwconv_bwd_reduce): needs the workspace
    ConvArgs a{};
    a.x = (const bf16_t*)x; a.mask = mask; a.w = w; a.pre = (bf16_t*)pre; a.dy = (const bf16_t*)dy;
    a.dx = (bf16_t*)dx; a.dw = dw; a.dbias = dbias; a.ws = ws; a.B = B; a.N = N; a.C = C; a.tune = split >> 1; a.defer_reduce = split & 1;
    int rc = dispatch_conv(a, ks, true, (hipStream_t)stream);
    if (rc) return rc;
    E2K_CHECK_LAUNCH();
    return 0;
}

// second half of e2k_dwconv_bwd(split bit 0): sums the per-workgroup (dw, dbias) partials of ws into dw / dbias.  Nothing on
// the backward chain reads them: the schedule runs it with the weight-gradient GEMMs on the WGRAD lane
static int dwconv_bwd_reduce_impl(const float* ws, float* dw, float* dbias, int B, int N, int C, int ks, int split, void* stream) {
    if (B <= 0 || N <= 0) return 0;
    if (!ws || !dw || !dbias || (C % CTC) || !(ks == 31 || ks == 15 || ks == 7 || ks == 3)) return E2K_ERR_ARG;
    const int gx = conv_bwd_gx(B, N, C, split >> 1);
    hipLaunchKernelGGL(conv_reduce_kernel, dim3(C / CTC, ks + 1), dim3(64), 0, (hipStream_t)stream,
                       ws, dw, dbias, B, C / CTC, gx, ks);
    E2K_CHECK_LAUNCH();
    return 0;
}

// ---- C ABI: every compute entry point goes through e2k::dispatch (plan.h) so that a launch plan can record it

extern "C" int e2k_dwconv_bwd_reduce(const float* ws, float* dw, float* dbias, int B, int N, int C, int ks, int split, void* stream) {
    return e2k::dispatch("dwconv_bwd_reduce", dwconv_bwd_reduce_impl, ws, dw, dbias, B, N, C, ks, split, stream);
}

extern "C" int e2k_rmsnorm_fwd(const void* x, const float* gamma, int64_t ldg, float gamma_off, int rows_per_batch,
                               void* y, float* rn, int M, int D, void* stream) {
    return e2k::dispatch("rmsnorm_fwd", rmsnorm_fwd_impl, x, gamma, ldg, gamma_off, rows_per_batch, y, rn, M, D, stream);
}

extern "C" int e2k_rmsnorm_bwd(const void* dy, const void* x, const float* rn, const float* gamma, int64_t ldg,
                               float gamma_off, int rows_per_batch, void* dx, float* dgamma, int M, int D,
                               void* stream) {
    return e2k::dispatch("rmsnorm_bwd", rmsnorm_bwd_impl, dy, x, rn, gamma, ldg, gamma_off, rows_per_batch, dx, dgamma, M, D, stream);
}

extern "C" int e2k_gate_bwd(const void* dy, const void* y, const float* g, void* dao, float* gsum, int64_t ldg,
                            int M, int D, int rows_per_batch, void* stream) {
    return e2k::dispatch("gate_bwd", gate_bwd_impl, dy, y, g, dao, gsum, ldg, M, D, rows_per_batch, stream);
}

extern "C" int e2k_geglu_fwd(const void* H, int64_t ldh, void* out, int M, int F, float p_drop, uint32_t seed,
                             const uint32_t* seed_dev, uint32_t stream_id, void* stream) {
    return e2k::dispatch("geglu_fwd", geglu_fwd_impl, H, ldh, out, M, F, p_drop, seed, seed_dev, stream_id, stream);
}

extern "C" int e2k_geglu_bwd(const void* dout, const void* H, int64_t ldh, void* dH, int M, int F, float p_drop,
                             uint32_t seed, const uint32_t* seed_dev, uint32_t stream_id, void* stream) {
    return e2k::dispatch("geglu_bwd", geglu_bwd_impl, dout, H, ldh, dH, M, F, p_drop, seed, seed_dev, stream_id, stream);
}

extern "C" int e2k_colsum_bf16(const void* x, int64_t ldx, float* out, int M, int N, void* stream) {
    return e2k::dispatch("colsum_bf16", colsum_bf16_impl, x, ldx, out, M, N, stream);
}

extern "C" int e2k_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
    return e2k::dispatch("cast_bf16", cast_bf16_impl, src, dst, n, stream);
}

extern "C" int e2k_cast_transpose_bf16(const float* src, void* dst, int R, int C, int64_t ldd, void* stream) {
    return e2k::dispatch("cast_transpose_bf16", cast_transpose_bf16_impl, src, dst, R, C, ldd, stream);
}

extern "C" int e2k_cast_transpose_batch(const float* flat, void* flatT, const int64_t* desc, int n, int total_blocks, void* stream) {
    return e2k::dispatch("cast_transpose_batch", cast_transpose_batch_impl, flat, flatT, desc, n, total_blocks, stream);
}

extern "C" int e2k_dwconv_fwd(const void* x, const uint8_t* mask, const float* w, const float* bias, void* pre,
                              void* y, int B, int N, int C, int ks, void* stream) {
    return e2k::dispatch("dwconv_fwd", dwconv_fwd_impl, x, mask, w, bias, pre, y, B, N, C, ks, stream);
}

extern "C" int e2k_query_dwconv_bwd_ws_floats(int B, int N, int C, int ks) {
    if (B <= 0 || N <= 0 || C <= 0 || (C % CTC)) return 0;
    return conv_bwd_gx(B, N, C, 0) * B * (C / CTC) * CTC * (ks + 1);
}

extern "C" int e2k_dwconv_bwd(const void* dy, const void* pre, const void* x, const uint8_t* mask, const float* w,
                              void* dx, float* dw, float* dbias, float* ws, int B, int N, int C, int ks, int split, void* stream) {
    return e2k::dispatch("dwconv_bwd", dwconv_bwd_impl, dy, pre, x, mask, w, dx, dw, dbias, ws, B, N, C, ks, split, stream);
}

extern "C" int e2k_fourier_cat_fwd(const float* h, int64_t ldh, void* y, int64_t ldy, int64_t M, int nf, int nrest, void* stream) {
    return e2k::dispatch("fourier_cat_fwd", fourier_cat_fwd_impl, h, ldh, y, ldy, M, nf, nrest, stream);
}

extern "C" int e2k_fourier_cat_bwd(const void* dy, int64_t ldy, const float* h, int64_t ldh, void* dh, int64_t lddh, int64_t M, int nf,
                                   int nrest, void* stream) {
    return e2k::dispatch("fourier_cat_bwd", fourier_cat_bwd_impl, dy, ldy, h, ldh, dh, lddh, M, nf, nrest, stream);
}
