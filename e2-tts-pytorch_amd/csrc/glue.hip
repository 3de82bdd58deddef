// Glue kernels around the depth loop: stream pack / unpack (abs-pos + registers + expansion to the 4 hyper-connection
// streams and its reduction), key masks, the time-conditioning MLP and the backward of the hoisted conditioning block.
// All HBM-bound element-wise work, written so that a whole forward / backward of the backbone is a sequence of e2k_*
// calls a launch plan can record (plan.h) -- no tensor-library ops in between.
//
// Reference: Transformer.forward e2_tts.py:760-771 (abs_pos, registers, mask padding), :818 / :947-949 (expand / reduce
// streams, drop registers), RandomFourierEmbed + time_cond_mlp :355-364,621-625,782.
#include "e2k_device.h"
#include "plan.h"
#include "../../include/e2k.h"

using namespace e2k;

namespace {

inline int grid_1d(long work, int per_block = 256, int cap = 4096) {
    long g = (work + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* src, float* dst, long n) {
    const long nv = n / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float f[8];
        unpack8(ld<u32x4>(src + i * 8), f);
        st<f32x4>(dst + i * 8, f32x4{f[0], f[1], f[2], f[3]});
        st<f32x4>(dst + i * 8 + 4, f32x4{f[4], f[5], f[6], f[7]});
    }
    if (blockIdx.x == 0)
        for (long i = nv * 8 + threadIdx.x; i < n; i += 256) dst[i] = bf2f(src[i]);
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* src, float* dst, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = sigmoidf_(src[i]);
}

__global__ __launch_bounds__(256) void fill2d_kernel(unsigned char* dst, long pitch, int value, long width, long rows) {
    const long total = width * rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / width, c = i - r * width;
        dst[r * pitch + c] = (unsigned char)value;
    }
}

__global__ __launch_bounds__(256) void build_masks_kernel(const uint8_t* mask, uint8_t* kmask, uint8_t* mask_n, int B, int T,
                                                           int R, int Npad) {
    const int N = T + R;
    const long total = (long)B * Npad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / Npad), n = (int)(i - (long)b * Npad);
        uint8_t v = 0;
        if (n < R) v = 1;
        else if (n < N) v = mask ? (mask[(long)b * T + (n - R)] ? 1 : 0) : 1;
        kmask[i] = v;
        if (mask_n && n < N) mask_n[(long)b * N + n] = v;
    }
}

// one thread = 8 consecutive channels of one (b, n): writes the same 16 bytes to the 4 streams
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* x, const float* abs_pos, const float* regs, bf16_t* X,
                                                        int B, int T, int R, int D) {
    const int N = T + R, D8 = D / 8;
    const long total = (long)B * N * D8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D8);
        const long bn = i / D8;
        const int n = (int)(bn % N), b = (int)(bn / N);
        float f[8];
        if (n < R) {
            const float* s = regs + (long)n * D + c * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = s[k];
        } else {
            const float* s = x + ((long)b * T + (n - R)) * D + c * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = s[k];
            if (abs_pos) {
                const float* a = abs_pos + (long)(n - R) * D + c * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] += a[k];
            }
        }
        const u32x4 v = pack8(f);
        bf16_t* o = X + bn * 4 * D + c * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) st<u32x4>(o + (long)s * D, v);
    }
}

// one thread = 8 channels of one position n, loops over the batch: dx rows and the batch-summed register / abs-pos rows
__global__ __launch_bounds__(256) void pack_bwd_kernel(const bf16_t* dX, float* dx, float* dregs, float* dabs, int B, int T,
                                                        int R, int D) {
    const int N = T + R, D8 = D / 8;
    const long total = (long)N * D8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D8), n = (int)(i / D8);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            const bf16_t* g = dX + ((long)b * N + n) * 4 * D + c * 8;
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float f[8];
                unpack8(ld<u32x4>(g + (long)s * D), f);
#pragma unroll
                for (int k = 0; k < 8; ++k) s8[k] += f[k];
            }
            if (n >= R) {
                float* o = dx + ((long)b * T + (n - R)) * D + c * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = s8[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += s8[k];
        }
        float* o = n < R ? dregs + (long)n * D + c * 8 : (dabs ? dabs + (long)(n - R) * D + c * 8 : nullptr);
        if (o) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] += acc[k];
        }
    }
}

__global__ __launch_bounds__(256) void unpack_fwd_kernel(const bf16_t* X, bf16_t* xsum, int B, int T, int R, int D) {
    const int N = T + R, D8 = D / 8;
    const long total = (long)B * T * D8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D8);
        const long bt = i / D8;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const bf16_t* g = X + ((long)b * N + R + t) * 4 * D + c * 8;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float f[8];
            unpack8(ld<u32x4>(g + (long)s * D), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) s8[k] += f[k];
        }
        st<u32x4>(xsum + bt * D + c * 8, pack8(s8));
    }
}

__global__ __launch_bounds__(256) void unpack_bwd_kernel(const bf16_t* dxs, bf16_t* dX, int B, int T, int R, int D) {
    const int N = T + R, D8 = D / 8;
    const long total = (long)B * N * D8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % D8);
        const long bn = i / D8;
        const int n = (int)(bn % N), b = (int)(bn / N);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (n >= R) v = ld<u32x4>(dxs + ((long)b * T + (n - R)) * D + c * 8);
        bf16_t* o = dX + bn * 4 * D + c * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) st<u32x4>(o + (long)s * D, v);
    }
}

// ---- time conditioning: B is the batch (a handful of rows), D up to a few thousand -> one wave per output channel

__global__ __launch_bounds__(256) void fourier_kernel(const float* times, const float* fw, float* four, int B, int D) {
    const int H = D / 2, K = D + 1;
    const long total = (long)B * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / K), k = (int)(i - (long)b * K);
        const float t = times[b];
        float v;
        if (k == 0) v = t;
        else {
            const int j = (k - 1) % H;
            const float fr = t * fw[j] * 2.f * 3.14159265358979323846f;       // x * weights * 2 * pi, in that order (e2_tts.py:362)
            v = (k - 1) < H ? sinf(fr) : cosf(fr);
        }
        four[i] = v;
    }
}

__global__ __launch_bounds__(256) void time_mlp_fwd_kernel(const float* four, const float* W, const float* bias, float* pre,
                                                            float* out, int B, int D) {
    const int K = D + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= D) return;
    const float* w = W + (long)j * K;
    for (int b = 0; b < B; ++b) {
        const float* f = four + (long)b * K;
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += w[k] * f[k];
        s = wave_sum(s);
        if (lane == 0) {
            const float p = s + bias[j];
            pre[(long)b * D + j] = p;
            out[(long)b * D + j] = siluf_(p);
        }
    }
}

// dW[j][k] += sum_b dpre[b][j] four[b][k];  dbias[j] += sum_b dpre[b][j];   dpre = dout * silu'(pre)
__global__ __launch_bounds__(256) void time_mlp_bwd_kernel(const float* dout, const float* four, const float* pre, float* dW,
                                                            float* dbias, int B, int D) {
    const int K = D + 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= D) return;
    float db = 0.f;
    for (int b = 0; b < B; ++b) {
        const float p = pre[(long)b * D + j], sg = sigmoidf_(p);
        const float dp = dout[(long)b * D + j] * (sg * (1.f + p * (1.f - sg)));
        db += dp;
        const float* f = four + (long)b * K;
        float* w = dW + (long)j * K;
        for (int k = lane; k < K; k += 64) w[k] += dp * f[k];
    }
    if (lane == 0) dbias[j] += db;
}

// backward of the hoisted conditioning block; one thread per (l, slot, d) column, loops over the batch
__global__ __launch_bounds__(256) void cond_bwd_prep_kernel(float* dcond, const float* gates, bf16_t* dcb, bf16_t* dct, float* gbias,
                                                             int B, int L, int D, int KB) {
    const long C = (long)4 * L * D;
    for (long col = (long)blockIdx.x * 256 + threadIdx.x; col < C; col += (long)gridDim.x * 256) {
        const int slot = (int)((col / D) & 3);
        const bool is_gate = slot & 1;
        float sum = 0.f;
        for (int b = 0; b < KB; ++b) {
            float v = 0.f;
            if (b < B) {
                v = dcond[(long)b * C + col];
                if (is_gate) {
                    v *= 1.f - gates[(long)b * C + col];
                    dcond[(long)b * C + col] = v;
                }
                dcb[(long)b * C + col] = f2bf(v);
                sum += v;
            }
            dct[col * KB + b] = f2bf(v);
        }
        if (is_gate) gbias[col] += sum;
    }
}

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* in, long ld, float* out, int R, int C) {
    const long total = (long)R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i / R), r = (int)(i - (long)c * R);
        out[i] = in[(long)r * ld + c];
    }
}

// fp32 (R, C) -> bf16 (R, ldd) with the columns C..Cpad-1 of every row zeroed (K padding of the 100-channel input /
// output projections: the GEMM kernels want K in multiples of 8)
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* src, long lds_, bf16_t* dst, long ldd, int R, int C, int Cpad) {
    const long total = (long)R * Cpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / Cpad), c = (int)(i - (long)r * Cpad);
        dst[(long)r * ldd + c] = c < C ? f2bf(src[(long)r * lds_ + c]) : (bf16_t)0;
    }
}

// The prologue of E2TTS.forward (e2_tts.py:1519-1543) in one pass over the (B T, C) mel frames: w = (1 - t) x0 + t x1 (the point on the
// probability path), flow = x1 - x0 (the regression target), cond = x1 outside the masked span and 0 inside it -- and the two operands
// of the input projection GEMM, w and cond as bf16 rows zero-padded to Cpad columns (what e2k_cast_pad_bf16 made of them in two more
// launches).  The products and the sum of w are rounded one by one (no contraction into an fma): bit for bit the tensor library's
// (1 - t) * x0 + t * x1.
__global__ __launch_bounds__(256) void flow_pack_kernel(const float* x0, const float* x1, const float* t, const uint8_t* span, bf16_t* wb, bf16_t* cb,
                                                        long ldp, float* flow, float* cond, int M, int T, int C, int Cpad) {
    const long total = (long)M * Cpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / Cpad), c = (int)(i - (long)r * Cpad);
        bf16_t wv = 0, cv = 0;
        if (c < C) {
            const long j = (long)r * C + c;
            const float a = x0[j], b = x1[j], tt = t[r / T];
            const float w = __fadd_rn(__fmul_rn(1.f - tt, a), __fmul_rn(tt, b));
            const float cd = span[r] ? 0.f : b;
            flow[j] = b - a;
            cond[j] = cd;
            wv = f2bf(w);
            cv = f2bf(cd);
        }
        wb[(long)r * ldp + c] = wv;
        cb[(long)r * ldp + c] = cv;
    }
}

// masked mean squared error over the masked span (e2_tts.py:1580-1582: F.mse_loss(pred, flow)[mask].mean()):
// acc[0] += sum_m mask[m] * sum_c (pred - flow)^2,  acc[1] += sum_m mask[m]      (fp32 atomics, one per block)
__global__ __launch_bounds__(256) void masked_mse_fwd_kernel(const float* pred, const float* flow, const uint8_t* mask, float* acc,
                                                              int M, int C) {
    __shared__ float red[2][4];
    float s = 0.f, cnt = 0.f;
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / C);
        if (mask[m]) {
            const float d = pred[i] - flow[i];
            s = fmaf(d, d, s);
            if (i - (long)m * C == 0) cnt += 1.f;
        }
    }
    s = wave_sum(s);
    cnt = wave_sum(cnt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(acc + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}
// loss = acc[0] / (acc[1] * C);   dpred = dloss * 2 (pred - flow) mask / (acc[1] * C)
__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(const float* pred, const float* flow, const uint8_t* mask, const float* acc,
                                                              const float* dloss, float* dpred, int M, int C) {
    const float k = 2.f * dloss[0] / (acc[1] * (float)C);
    const long total = (long)M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / C);
        dpred[i] = mask[m] ? k * (pred[i] - flow[i]) : 0.f;
    }
}

// ---- CharacterEmbed (e2_tts.py:390-412): out[b][n] = W[n < nt ? tok[b][n] + 1 : 0]  (the byte tokens are
// shifted by one so that the -1 padding and the frames past the text both land on row 0); fp32 rows of D.
__global__ __launch_bounds__(256) void char_embed_fwd_kernel(const long* tok, const float* W, float* out, int B, int nt, int T, int D, int V) {
    const long total = (long)B * T * (D / 4);
    const int dv = D / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long bn = i / dv;
        const int c = (int)(i - bn * dv) * 4, n = (int)(bn % T), b = (int)(bn / T);
        long idx = n < nt ? tok[(long)b * nt + n] + 1 : 0;
        idx = idx < 0 ? 0 : (idx >= V ? V - 1 : idx);               // (nn.Embedding raises on out-of-range ids: _CharEmbedFn checks host-side token tensors, device ones under E2K_CHECK_TOKEN_IDS=1; this only keeps the read in bounds)
        st<f32x4>(out + bn * D + c, ld<f32x4>(W + idx * D + c));
    }
}
// dW[idx] += dout[b][n]  (fp32 atomics: rows of the 257-entry table collect thousands of tokens each)
__global__ __launch_bounds__(256) void char_embed_bwd_kernel(const long* tok, const float* dout, float* dW, int B, int nt, int T, int D, int V) {
    const long total = (long)B * T * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long bn = i / D;
        const int c = (int)(i - bn * D), n = (int)(bn % T), b = (int)(bn / T);
        long idx = n < nt ? tok[(long)b * nt + n] + 1 : 0;
        idx = idx < 0 ? 0 : (idx >= V ? V - 1 : idx);
        atomicAdd(dW + idx * D + c, dout[i]);
    }
}

// ---- duration head (e2_tts.py:1098-1111 with maybe_masked_mean :212-224 and HLGaussLayer's regression mode):
//   pooled[b] = sum_n mask[b][n] embed[b][n] / max(sum_n mask[b][n], 1)      (mask = NULL: plain mean over n)
//   pred[b]   = softplus(w . pooled[b])                                      (nn.Softplus: x for x > 20)
// One workgroup per batch row; pooled and the pre-activation z are kept for the backward.
__global__ __launch_bounds__(256) void duration_head_fwd_kernel(const float* embed, const uint8_t* mask, const float* w, float* pooled,
                                                                 float* z, float* pred, int T, int D) {
    __shared__ float red[4];
    __shared__ float cnt_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* e = embed + (long)b * T * D;
    const uint8_t* m = mask ? mask + (long)b * T : nullptr;
    if (tid == 0) {
        float c = 0.f;
        for (int n = 0; n < T; ++n) c += (!m || m[n]) ? 1.f : 0.f;
        cnt_s = m ? fmaxf(c, 1.f) : (float)T;
    }
    __syncthreads();
    const float inv = 1.f / cnt_s;
    float dot = 0.f;
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int n = 0; n < T; ++n)
            if (!m || m[n]) s += e[(long)n * D + d];
        s *= inv;
        pooled[(long)b * D + d] = s;
        dot = fmaf(s, w[d], dot);
    }
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) {
        const float zz = red[0] + red[1] + red[2] + red[3];
        z[b] = zz;
        pred[b] = zz > 20.f ? zz : log1pf(__expf(zz));
    }
}
// dpred (B) -> dembed[b][n] = mask[b][n] / cnt_b * dz_b w ;  dw += sum_b dz_b pooled[b] ;  dz = dpred sigmoid(z)
__global__ __launch_bounds__(256) void duration_head_bwd_kernel(const float* dpred, const float* z, const float* pooled, const uint8_t* mask,
                                                                 const float* w, float* dembed, float* dw, int T, int D) {
    __shared__ float cnt_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* m = mask ? mask + (long)b * T : nullptr;
    if (tid == 0) {
        float c = 0.f;
        for (int n = 0; n < T; ++n) c += (!m || m[n]) ? 1.f : 0.f;
        cnt_s = m ? fmaxf(c, 1.f) : (float)T;
    }
    __syncthreads();
    const float zz = z[b];
    const float dz = dpred[b] * (zz > 20.f ? 1.f : sigmoidf_(zz));
    const float k = dz / cnt_s;
    float* de = dembed + (long)b * T * D;
    for (int d = tid; d < D; d += 256) {
        const float g = k * w[d];
        for (int n = 0; n < T; ++n) de[(long)n * D + d] = (!m || m[n]) ? g : 0.f;
        atomicAdd(dw + d, dz * pooled[(long)b * D + d]);
    }
}

}  // namespace

static int flow_pack_impl(const float* x0, const float* x1, const float* t, const uint8_t* span_mask, void* w_bf16, void* cond_bf16, int64_t ldp,
                          float* flow, float* cond, int B, int T, int C, int Cpad, void* stream) {
    if (B <= 0 || T <= 0 || Cpad <= 0) return 0;
    if (!x0 || !x1 || !t || !span_mask || !w_bf16 || !cond_bf16 || !flow || !cond) return E2K_ERR_ARG;
    if (C <= 0 || C > Cpad || ldp < Cpad) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(flow_pack_kernel, dim3(grid_1d((long)B * T * Cpad)), dim3(256), 0, (hipStream_t)stream, x0, x1, t, span_mask, (bf16_t*)w_bf16,
                       (bf16_t*)cond_bf16, (long)ldp, flow, cond, B * T, T, C, Cpad);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cast_pad_bf16_impl(const float* src, int64_t lds_, void* dst, int64_t ldd, int R, int C, int Cpad, void* stream) {
    if (R <= 0 || Cpad <= 0) return 0;
    if (C > Cpad || ldd < Cpad || lds_ < C) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_1d((long)R * Cpad)), dim3(256), 0, (hipStream_t)stream, src, (long)lds_, (bf16_t*)dst,
                       (long)ldd, R, C, Cpad);
    E2K_CHECK_LAUNCH();
    return 0;
}

// Classifier-free-guidance combine of sample() (e2_tts.py:1303-1330: cfg_transformer_with_pred_head + project):
//   u = pred - null ;  unit = pred / max(|pred|, 1e-12)  (fp64, per sample over all of its (n, d) elements)
//   par = (u . unit) unit ;  orth = u - par   (fp64, each rounded to fp32 as `project` returns them)
//   out = pred + (orth + par * keep_parallel_frac) * cfg_strength          (remove_parallel = 0: out = pred + u * cfg_strength)
// One workgroup per sample: reduce (fp64), then a second pass over the sample (L2-resident) -- six tensor-library passes and
// two fp64 copies of the predictions in the reference.
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* pred, const float* nul, float* out, long L, float strength,
                                                          float keep_frac, int remove_parallel) {
    __shared__ double red[2][4];
    const long b = blockIdx.x;
    const float* p = pred + b * L;
    const float* q = nul + b * L;
    float* o = out + b * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double coef = 0.0, inv = 0.0;
    if (remove_parallel) {
        double up = 0.0, pp = 0.0;
        for (long i = tid; i < L; i += 256) {
            const double pv = (double)p[i], uv = (double)(p[i] - q[i]);          // (pred - null is formed in fp32, then widened)
            up = fma(uv, pv, up);
            pp = fma(pv, pv, pp);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { up += __shfl_xor(up, m); pp += __shfl_xor(pp, m); }
        if (lane == 0) { red[0][wave] = up; red[1][wave] = pp; }
        __syncthreads();
        up = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        pp = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        inv = 1.0 / fmax(sqrt(pp), 1e-12);               // F.normalize(y, dim=-1): y / max(|y|, eps)
        coef = up * inv;                                   // u . unit
    }
    for (long i = tid; i < L; i += 256) {
        const float pv = p[i], uv = pv - q[i];
        float upd = uv;
        if (remove_parallel) {
            const double par = coef * ((double)pv * inv);
            const float parf = (float)par, orthf = (float)((double)uv - par);
            upd = orthf + parf * keep_frac;
        }
        o[i] = pv + upd * strength;
    }
}

static int cfg_combine_impl(const float* pred, const float* null_pred, float* out, int B, int64_t L, float cfg_strength,
                            float keep_parallel_frac, int remove_parallel, void* stream) {
    if (B <= 0 || L <= 0) return 0;
    if (!pred || !null_pred || !out) return E2K_ERR_ARG;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pred, null_pred, out, (long)L, cfg_strength,
                       keep_parallel_frac, remove_parallel);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int masked_mse_fwd_impl(const float* pred, const float* flow, const uint8_t* mask, float* acc, int M, int C, void* stream) {
    if (M <= 0 || C <= 0) return E2K_ERR_SHAPE;
    hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return 1000 + (int)e;
    hipLaunchKernelGGL(masked_mse_fwd_kernel, dim3(grid_1d((long)M * C, 256, 1024)), dim3(256), 0, (hipStream_t)stream, pred, flow, mask,
                       acc, M, C);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int masked_mse_bwd_impl(const float* pred, const float* flow, const uint8_t* mask, const float* acc, const float* dloss,
                               float* dpred, int M, int C, void* stream) {
    if (M <= 0 || C <= 0) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3(grid_1d((long)M * C)), dim3(256), 0, (hipStream_t)stream, pred, flow, mask, acc,
                       dloss, dpred, M, C);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int char_embed_fwd_impl(const int64_t* tok, const float* W, float* out, int B, int nt, int T, int D, int V, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (D <= 0 || (D & 3) || V <= 0 || nt < 0) return E2K_ERR_SHAPE;
    if (!W || !out || (nt > 0 && !tok)) return E2K_ERR_ARG;
    if (((uintptr_t)W | (uintptr_t)out) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(char_embed_fwd_kernel, dim3(grid_1d((long)B * T * (D / 4))), dim3(256), 0, (hipStream_t)stream, (const long*)tok, W,
                       out, B, nt, T, D, V);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int char_embed_bwd_impl(const int64_t* tok, const float* dout, float* dW, int B, int nt, int T, int D, int V, void* stream) {
    if (B <= 0 || T <= 0) return 0;
    if (D <= 0 || V <= 0 || nt < 0) return E2K_ERR_SHAPE;
    if (!dout || !dW || (nt > 0 && !tok)) return E2K_ERR_ARG;
    hipLaunchKernelGGL(char_embed_bwd_kernel, dim3(grid_1d((long)B * T * D)), dim3(256), 0, (hipStream_t)stream, (const long*)tok, dout, dW,
                       B, nt, T, D, V);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int duration_head_fwd_impl(const float* embed, const uint8_t* mask, const float* w, float* pooled, float* z, float* pred, int B,
                                  int T, int D, void* stream) {
    if (B <= 0) return 0;
    if (T <= 0 || D <= 0) return E2K_ERR_SHAPE;
    if (!embed || !w || !pooled || !z || !pred) return E2K_ERR_ARG;
    hipLaunchKernelGGL(duration_head_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, embed, mask, w, pooled, z, pred, T, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int duration_head_bwd_impl(const float* dpred, const float* z, const float* pooled, const uint8_t* mask, const float* w,
                                  float* dembed, float* dw, int B, int T, int D, void* stream) {
    if (B <= 0) return 0;
    if (T <= 0 || D <= 0) return E2K_ERR_SHAPE;
    if (!dpred || !z || !pooled || !w || !dembed || !dw) return E2K_ERR_ARG;
    hipLaunchKernelGGL(duration_head_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dpred, z, pooled, mask, w, dembed, dw, T, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int fill_bytes_impl(void* dst, int value, int64_t nbytes, void* stream) {
    if (nbytes <= 0) return 0;
    if (!dst) return E2K_ERR_ARG;
    hipError_t e = hipMemsetAsync(dst, value, (size_t)nbytes, (hipStream_t)stream);
    return e == hipSuccess ? 0 : 1000 + (int)e;
}

static int fill_bytes_2d_impl(void* dst, int64_t pitch, int value, int64_t width, int64_t rows, void* stream) {
    if (width <= 0 || rows <= 0) return 0;
    if (!dst || pitch < width) return E2K_ERR_ARG;
    hipLaunchKernelGGL(fill2d_kernel, dim3(grid_1d(width * rows)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)dst,
                       (long)pitch, value, (long)width, (long)rows);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cast_f32_impl(const void* src_bf16, float* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (((uintptr_t)src_bf16 | (uintptr_t)dst) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_1d(n / 8 + 1)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_bf16, dst,
                       (long)n);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int sigmoid_f32_impl(const float* src, float* dst, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_1d(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int build_masks_impl(const uint8_t* mask, uint8_t* kmask, uint8_t* mask_n, int B, int T, int R, int Npad, void* stream) {
    if (B <= 0 || T <= 0 || R < 0 || Npad < T + R) return E2K_ERR_SHAPE;
    if (!kmask) return E2K_ERR_ARG;
    hipLaunchKernelGGL(build_masks_kernel, dim3(grid_1d((long)B * Npad)), dim3(256), 0, (hipStream_t)stream, mask, kmask, mask_n,
                       B, T, R, Npad);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int stream_pack_fwd_impl(const float* x, const float* abs_pos, const float* regs, void* X, int B, int T, int R, int D,
                                void* stream) {
    if (B <= 0 || T <= 0 || R < 0 || D <= 0 || (D & 7)) return E2K_ERR_SHAPE;
    if (((uintptr_t)X) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(grid_1d((long)B * (T + R) * (D / 8))), dim3(256), 0, (hipStream_t)stream, x, abs_pos,
                       regs, (bf16_t*)X, B, T, R, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int stream_pack_bwd_impl(const void* dX, float* dx, float* dregs, float* dabs, int B, int T, int R, int D, void* stream) {
    if (B <= 0 || T <= 0 || R < 0 || D <= 0 || (D & 7)) return E2K_ERR_SHAPE;
    if (((uintptr_t)dX) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(grid_1d((long)(T + R) * (D / 8), 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dX, dx, dregs, dabs, B, T, R, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int stream_unpack_fwd_impl(const void* X, void* xsum, int B, int T, int R, int D, void* stream) {
    if (B <= 0 || T <= 0 || R < 0 || D <= 0 || (D & 7)) return E2K_ERR_SHAPE;
    if (((uintptr_t)X | (uintptr_t)xsum) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(unpack_fwd_kernel, dim3(grid_1d((long)B * T * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X,
                       (bf16_t*)xsum, B, T, R, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int stream_unpack_bwd_impl(const void* dxs, void* dX, int B, int T, int R, int D, void* stream) {
    if (B <= 0 || T <= 0 || R < 0 || D <= 0 || (D & 7)) return E2K_ERR_SHAPE;
    if (((uintptr_t)dX | (uintptr_t)dxs) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(unpack_bwd_kernel, dim3(grid_1d((long)B * (T + R) * (D / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dxs, (bf16_t*)dX, B, T, R, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int time_cond_fwd_impl(const float* times, const float* fw, const float* W, const float* bias, float* four, float* pre,
                              float* out, int B, int D, void* stream) {
    if (B <= 0 || D <= 0 || (D & 1)) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(fourier_kernel, dim3(grid_1d((long)B * (D + 1))), dim3(256), 0, (hipStream_t)stream, times, fw, four, B, D);
    E2K_CHECK_LAUNCH();
    hipLaunchKernelGGL(time_mlp_fwd_kernel, dim3((D + 3) / 4), dim3(256), 0, (hipStream_t)stream, four, W, bias, pre, out, B, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int time_cond_bwd_impl(const float* dout, const float* four, const float* pre, float* dW, float* dbias, int B, int D,
                              void* stream) {
    if (B <= 0 || D <= 0 || (D & 1)) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(time_mlp_bwd_kernel, dim3((D + 3) / 4), dim3(256), 0, (hipStream_t)stream, dout, four, pre, dW, dbias, B, D);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int cond_bwd_prep_impl(float* dcond, const float* gates, void* dcb, void* dct, float* gbias, int B, int L, int D, int KB,
                              void* stream) {
    if (B <= 0 || L <= 0 || D <= 0 || KB < B) return E2K_ERR_SHAPE;
    hipLaunchKernelGGL(cond_bwd_prep_kernel, dim3(grid_1d((long)4 * L * D)), dim3(256), 0, (hipStream_t)stream, dcond, gates,
                       (bf16_t*)dcb, (bf16_t*)dct, gbias, B, L, D, KB);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int transpose_f32_impl(const float* in, int64_t ld, float* out, int R, int C, void* stream) {
    if (R <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3(grid_1d((long)R * C)), dim3(256), 0, (hipStream_t)stream, in, (long)ld, out, R, C);
    E2K_CHECK_LAUNCH();
    return 0;
}

// ---- C ABI

extern "C" int e2k_fill_bytes(void* dst, int value, int64_t nbytes, void* stream) {
    return e2k::dispatch("fill_bytes", fill_bytes_impl, dst, value, nbytes, stream);
}
extern "C" int e2k_fill_bytes_2d(void* dst, int64_t pitch, int value, int64_t width, int64_t rows, void* stream) {
    return e2k::dispatch("fill_bytes_2d", fill_bytes_2d_impl, dst, pitch, value, width, rows, stream);
}
extern "C" int e2k_cast_f32(const void* src_bf16, float* dst, int64_t n, void* stream) {
    return e2k::dispatch("cast_f32", cast_f32_impl, src_bf16, dst, n, stream);
}
extern "C" int e2k_sigmoid_f32(const float* src, float* dst, int64_t n, void* stream) {
    return e2k::dispatch("sigmoid_f32", sigmoid_f32_impl, src, dst, n, stream);
}
extern "C" int e2k_build_masks(const uint8_t* mask, uint8_t* kmask, uint8_t* mask_n, int B, int T, int R, int Npad, void* stream) {
    return e2k::dispatch("build_masks", build_masks_impl, mask, kmask, mask_n, B, T, R, Npad, stream);
}
extern "C" int e2k_stream_pack_fwd(const float* x, const float* abs_pos, const float* regs, void* X, int B, int T, int R, int D,
                                   void* stream) {
    return e2k::dispatch("stream_pack_fwd", stream_pack_fwd_impl, x, abs_pos, regs, X, B, T, R, D, stream);
}
extern "C" int e2k_stream_pack_bwd(const void* dX, float* dx, float* dregs, float* dabs, int B, int T, int R, int D, void* stream) {
    return e2k::dispatch("stream_pack_bwd", stream_pack_bwd_impl, dX, dx, dregs, dabs, B, T, R, D, stream);
}
extern "C" int e2k_stream_unpack_fwd(const void* X, void* xsum, int B, int T, int R, int D, void* stream) {
    return e2k::dispatch("stream_unpack_fwd", stream_unpack_fwd_impl, X, xsum, B, T, R, D, stream);
}
extern "C" int e2k_stream_unpack_bwd(const void* dxs, void* dX, int B, int T, int R, int D, void* stream) {
    return e2k::dispatch("stream_unpack_bwd", stream_unpack_bwd_impl, dxs, dX, B, T, R, D, stream);
}
extern "C" int e2k_time_cond_fwd(const float* times, const float* fw, const float* W, const float* bias, float* four, float* pre,
                                 float* out, int B, int D, void* stream) {
    return e2k::dispatch("time_cond_fwd", time_cond_fwd_impl, times, fw, W, bias, four, pre, out, B, D, stream);
}
extern "C" int e2k_time_cond_bwd(const float* dout, const float* four, const float* pre, float* dW, float* dbias, int B, int D,
                                 void* stream) {
    return e2k::dispatch("time_cond_bwd", time_cond_bwd_impl, dout, four, pre, dW, dbias, B, D, stream);
}
extern "C" int e2k_cond_bwd_prep(float* dcond, const float* gates, void* dcb, void* dct, float* gbias, int B, int L, int D, int KB,
                                 void* stream) {
    return e2k::dispatch("cond_bwd_prep", cond_bwd_prep_impl, dcond, gates, dcb, dct, gbias, B, L, D, KB, stream);
}
extern "C" int e2k_transpose_f32(const float* in, int64_t ld, float* out, int R, int C, void* stream) {
    return e2k::dispatch("transpose_f32", transpose_f32_impl, in, ld, out, R, C, stream);
}
extern "C" int e2k_flow_pack(const float* x0, const float* x1, const float* t, const uint8_t* span_mask, void* w_bf16, void* cond_bf16, int64_t ldp,
                             float* flow, float* cond, int B, int T, int C, int Cpad, void* stream) {
    return e2k::dispatch("flow_pack", flow_pack_impl, x0, x1, t, span_mask, w_bf16, cond_bf16, ldp, flow, cond, B, T, C, Cpad, stream);
}

extern "C" int e2k_cast_pad_bf16(const float* src, int64_t lds_, void* dst, int64_t ldd, int R, int C, int Cpad, void* stream) {
    return e2k::dispatch("cast_pad_bf16", cast_pad_bf16_impl, src, lds_, dst, ldd, R, C, Cpad, stream);
}
extern "C" int e2k_masked_mse_fwd(const float* pred, const float* flow, const uint8_t* mask, float* acc, int M, int C, void* stream) {
    return e2k::dispatch("masked_mse_fwd", masked_mse_fwd_impl, pred, flow, mask, acc, M, C, stream);
}
extern "C" int e2k_masked_mse_bwd(const float* pred, const float* flow, const uint8_t* mask, const float* acc, const float* dloss,
                                  float* dpred, int M, int C, void* stream) {
    return e2k::dispatch("masked_mse_bwd", masked_mse_bwd_impl, pred, flow, mask, acc, dloss, dpred, M, C, stream);
}

extern "C" int e2k_cfg_combine(const float* pred, const float* null_pred, float* out, int B, int64_t L, float cfg_strength,
                               float keep_parallel_frac, int remove_parallel, void* stream) {
    return e2k::dispatch("cfg_combine", cfg_combine_impl, pred, null_pred, out, B, L, cfg_strength, keep_parallel_frac, remove_parallel, stream);
}

extern "C" int e2k_char_embed_fwd(const int64_t* tok, const float* W, float* out, int B, int nt, int T, int D, int V, void* stream) {
    return e2k::dispatch("char_embed_fwd", char_embed_fwd_impl, tok, W, out, B, nt, T, D, V, stream);
}
extern "C" int e2k_char_embed_bwd(const int64_t* tok, const float* dout, float* dW, int B, int nt, int T, int D, int V, void* stream) {
    return e2k::dispatch("char_embed_bwd", char_embed_bwd_impl, tok, dout, dW, B, nt, T, D, V, stream);
}
extern "C" int e2k_duration_head_fwd(const float* embed, const uint8_t* mask, const float* w, float* pooled, float* z, float* pred,
                                     int B, int T, int D, void* stream) {
    return e2k::dispatch("duration_head_fwd", duration_head_fwd_impl, embed, mask, w, pooled, z, pred, B, T, D, stream);
}
extern "C" int e2k_duration_head_bwd(const float* dpred, const float* z, const float* pooled, const uint8_t* mask, const float* w,
                                     float* dembed, float* dw, int B, int T, int D, void* stream) {
    return e2k::dispatch("duration_head_bwd", duration_head_bwd_impl, dpred, z, pooled, mask, w, dembed, dw, B, T, D, stream);
}
