// Optimizer-side kernels over flat fp32 buffers: the reference trainer does
//     accelerator.clip_grad_norm_(model.parameters(), max_grad_norm)      trainer.py:272-273
//     optimizer.step()            (adam_atan2_pytorch.adopt.Adopt)        trainer.py:183,275
//     ema_model.update()          (ema_pytorch.EMA)                       trainer.py:170,279
// as ~10 multi-tensor passes over every parameter.  Here: one reduction pass over the gradients (sum of squares, fp64
// accumulation) and ONE fused pass that applies the clip factor (computed on the device from that sum, no host sync),
// the ADOPT update, decoupled weight decay, refreshes the bf16 compute shadow and
// optionally folds the gradient back to zero -- 4 reads + 3-4 writes per element.  HBM bound.
#include "e2k_device.h"
#include <e2k_asm.h>
#include "plan.h"
#include "../../include/e2k.h"

using namespace e2k;

namespace {

// streaming accesses: every optimizer buffer is read and written once per step and is far larger than the caches (cfg3: 2.9 GB each), so
// the loads and stores carry the non-temporal hint (sum of squares 5.1 -> 5.8 TB/s, profiles/r05k_optim_kernels_ab.json)
__device__ __forceinline__ f32x4 ldv(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ void stv(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, double* out) {
    __shared__ double red[4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads per thread in flight
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldv(x + 4 * (i + u * stride));
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fma((double)v[u][r], (double)v[u][r], acc[r]);
    }
    for (; i < n4; i += stride) {
        const f32x4 v = ldv(x + 4 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fma((double)v[r], (double)v[r], acc[r]);
    }
    double a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[4 * n4 + threadIdx.x]; a += (double)v * v; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

struct AdoptArgs {
    float* p; const float* g; float* m; float* v; bf16_t* shadow;
    long n;
    float lr, beta1, beta2, eps, wd, max_norm, clamp;     // clamp = step^0.25 (ADOPT update clipping)
    const double* gsumsq;                                  // device: sum of squares of ALL gradients (may be null)
    int first;                                             // step 0: only v = g^2
    // second parameter group (Adopt keeps `steps` PER PARAMETER and skips parameters whose .grad is None -- the text
    // stream's on the steps whose classifier-free-guidance coin drops the text, trainer.py:183,275 / e2_tts.py:1261):
    // elements inside one of the `nranges` sorted [start, end) element ranges use (first_b, clamp_b), or are left
    // untouched when active_b == 0.  Range bounds are multiples of 4 elements.
    const int* ranges; int nranges; int first_b; float clamp_b; int active_b;
    // the exponential moving average of the parameters that ema_pytorch.EMA.update() computes right after the step (trainer.py:279),
    // folded into this pass while the new parameter value is in registers: ema += (p_new - ema) * ema_omd.  Null: not this step.
    // Elements of a group that is skipped keep their parameter and still move their average (the average follows the model).
    float* ema; float ema_omd;
};

// one element: returns the new parameter value.  v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the correctly rounded sequences (~25
// instructions per element): |u| differs by <= 3 ulp before the clamp, far inside the 2e-5 the oracle comparison allows
// (adam_atan2_pytorch.adopt.Adopt.step: the weight decay `p.mul_(1 - lr * wd)` comes FIRST -- before the state is created, i.e. also on the
//  step that only sets v = g^2 -- and a.wd is the effective factor: the caller divides weight_decay by the initial lr when decoupled_wd)
__device__ __forceinline__ float adopt_one(const AdoptArgs& a, bool first, float clamp, float p, float g, float& m, float& v) {
    p = p * (1.f - a.lr * a.wd);
    if (first) { v = g * g; return p; }
    const float q = g * fast_rcp(fmaxf(fast_sqrt(v), a.eps));
    const float u = fminf(fmaxf(q, -clamp), clamp);
    m = m + (1.f - a.beta1) * (u - m);
    p = p - a.lr * m;
    v = v + (1.f - a.beta2) * (g * g - v);
    return p;
}

// Blocks of 512 16-byte groups are dealt round-robin over the grid, so that at any moment the chip reads one compact window of each of
// the streams (round 3 gave each workgroup one contiguous span)
__global__ __launch_bounds__(256) void adopt_kernel(AdoptArgs a) {
    // clip factor of torch.nn.utils.clip_grad_norm_: min(1, max_norm / (total_norm + 1e-6))
    float cs = 1.f;
    if (a.gsumsq && a.max_norm > 0.f) cs = fminf(1.f, a.max_norm / ((float)sqrt(*a.gsumsq) + 1e-6f));
    const long n4 = a.n >> 2;
    // the ranges of the second group, staged once per workgroup (at most 128 ranges)
    __shared__ int rng[256];
    const int nr = a.nranges;
    for (int k = threadIdx.x; k < 2 * nr; k += 256) rng[k] = a.ranges[k];
    if (nr) __syncthreads();
    auto last_start = [&](long e) {      // number of ranges whose start is <= e (binary search over the sorted starts)
        int lo = 0, hi = nr;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long)rng[2 * mid] <= e) lo = mid + 1; else hi = mid; }
        return lo;
    };
    auto in_b = [&](long e) { const int lo = last_start(e); return lo > 0 && e < (long)rng[2 * lo - 1]; };
    auto one = [&](long i, bool grp_b, f32x4 p, f32x4 g, f32x4 m, f32x4 v, f32x4 e) {
        float pv[4] = {p[0], p[1], p[2], p[3]};
        bool first = a.first; float clamp = a.clamp;
        bool upd = true;
        if (grp_b) {
            upd = a.active_b;                               // no gradient this step: parameter, moments and step count stay
            first = a.first_b; clamp = a.clamp_b;
        }
        if (upd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { float mm = m[r], vv = v[r]; pv[r] = adopt_one(a, first, clamp, p[r], g[r] * cs, mm, vv); m[r] = mm; v[r] = vv; }
            stv(a.p + 4 * i, f32x4{pv[0], pv[1], pv[2], pv[3]});
            stv(a.m + 4 * i, m);
            stv(a.v + 4 * i, v);
            if (a.shadow) st<u32x2>(a.shadow + 4 * i, pack4(pv));
        }
        if (a.ema) stv(a.ema + 4 * i, e + (f32x4{pv[0], pv[1], pv[2], pv[3]} - e) * a.ema_omd);
    };
    // Blocks of 512 groups (two sweeps of the 256 threads in flight).  Which parameter group a block belongs to is decided once per
    // block from the sorted ranges, with workgroup-uniform state: a block is outside every range, inside one, or -- only at the few
    // range boundaries -- mixed, and only then do its lanes search.  Blocks of a group without a gradient are skipped before their
    // loads (unless the moving average has to follow them).  (Round 3: the per-lane binary search of every 16-byte group put seven
    // dependent LDS reads in front of each group's stores.)
    const long b1 = n4, bstep = (long)gridDim.x * 512;
    for (long base = (long)blockIdx.x * 512; base < b1; base += bstep) {
        const long e0 = 4 * base, e1 = 4 * min(base + 512, b1);
        int kind = 0;                                       // 0: outside every range, 1: inside range ri, 2: mixed
        if (nr) {
            int ri = last_start(e0);                        // -> the first range that ends past the block's start
            if (ri > 0 && (long)rng[2 * ri - 1] > e0) --ri;
            if (ri < nr && (long)rng[2 * ri] < e1) kind = ((long)rng[2 * ri] <= e0 && (long)rng[2 * ri + 1] >= e1) ? 1 : 2;
        }
        if (kind == 1 && !a.active_b && !a.ema) continue;
        const long i = base + threadIdx.x, j = i + 256;
        const bool hi = i < b1, hj = j < b1;
        f32x4 p0, g0, m0, v0, p1, g1, m1, v1, e0v, e1v;
        if (a.ema) {
            if (hi) e0v = ldv(a.ema + 4 * i);
            if (hj) e1v = ldv(a.ema + 4 * j);
        }
        if (kind == 1 && !a.active_b) {                     // only the average moves
            if (hi) p0 = ldv(a.p + 4 * i);
            if (hj) p1 = ldv(a.p + 4 * j);
            g0 = m0 = v0 = g1 = m1 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            if (hi) { p0 = ldv(a.p + 4 * i); g0 = ldv(a.g + 4 * i); m0 = ldv(a.m + 4 * i); v0 = ldv(a.v + 4 * i); }
            if (hj) { p1 = ldv(a.p + 4 * j); g1 = ldv(a.g + 4 * j); m1 = ldv(a.m + 4 * j); v1 = ldv(a.v + 4 * j); }
        }
        if (hi) one(i, kind == 1 || (kind == 2 && in_b(4 * i)), p0, g0, m0, v0, e0v);
        if (hj) one(j, kind == 1 || (kind == 2 && in_b(4 * j)), p1, g1, m1, v1, e1v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long i = 4 * n4 + threadIdx.x;
        float mm = a.m[i], vv = a.v[i];
        const float pn = adopt_one(a, a.first, a.clamp, a.p[i], a.g[i] * cs, mm, vv);  // (the tail is never inside a range)
        a.p[i] = pn; a.m[i] = mm; a.v[i] = vv;
        if (a.shadow) a.shadow[i] = f2bf(pn);
        if (a.ema) a.ema[i] += (pn - a.ema[i]) * a.ema_omd;
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* ema, const float* p, long n, float one_minus_decay) {
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {        // four independent 16-byte streams per thread in flight
        f32x4 e[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = ldv(ema + 4 * (i + u * stride)); w[u] = ldv(p + 4 * (i + u * stride)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) stv(ema + 4 * (i + u * stride), e[u] + (w[u] - e[u]) * one_minus_decay);
    }
    for (; i < n4; i += stride) {
        f32x4 e = ldv(ema + 4 * i);
        const f32x4 w = ldv(p + 4 * i);
        stv(ema + 4 * i, e + (w - e) * one_minus_decay);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const long j = 4 * n4 + threadIdx.x; ema[j] += (p[j] - ema[j]) * one_minus_decay; }
}

int grid_for(long n) {
    long g = (n / 4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
// the update pass: one workgroup per 512-group block (57 KB of traffic each; 350 000 workgroups at cfg3).  Many short workgroups dealt by
// the dispatcher keep the eight XCDs level where 4096 long ones drift apart: 3.15 against 3.38 ms alone, 4.04 against 4.37 ms with the
// average folded in (6.4 TB/s of the 6.3 a float4 copy reaches; profiles/r05m_optim_kernels_ab.json, measured with a grid cap switch
// that is gone again).
int adopt_grid_for(long n) {
    long g = (n / 4 + 511) / 512;
    return (int)(g < 1 ? 1 : (g > (1L << 22) ? (1L << 22) : g));
}

// ---- gradient slab <-> bf16 wire format of the data-parallel exchange (ddp.py; replaces the implicit DDP reducer of
// trainer.py:155-162,190-192): wire = bf16(g * scale) in ONE pass into a preallocated buffer, and g = float(wire) back --
// 6 B per element each way instead of the three tensor-library passes (scale, round, copy: 20 B and a 58-MB allocation
// per slab) that cost 11.5 ms per cfg3 step on the side stream (profiles/r02_force_ddp_ab.jsonl)
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ wire, long n, float scale) {
    const long nv = n >> 3, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        const f32x4 a = ld<f32x4>(g + i * 8) * scale, b = ld<f32x4>(g + i * 8 + 4) * scale;
        float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        st<u32x4>(wire + i * 8, pack8(f));
    }
    if (blockIdx.x == 0)
        for (long i = nv * 8 + threadIdx.x; i < n; i += 256) wire[i] = f2bf(g[i] * scale);
}
__global__ __launch_bounds__(256) void grad_unpack_kernel(const bf16_t* __restrict__ wire, float* __restrict__ g, long n) {
    const long nv = n >> 3, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        float f[8];
        unpack8(ld<u32x4>(wire + i * 8), f);
        st<f32x4>(g + i * 8, f32x4{f[0], f[1], f[2], f[3]});
        st<f32x4>(g + i * 8 + 4, f32x4{f[4], f[5], f[6], f[7]});
    }
    if (blockIdx.x == 0)
        for (long i = nv * 8 + threadIdx.x; i < n; i += 256) g[i] = bf2f(wire[i]);
}

// bf16 wire, fp32 sum (ddp._GradSync wire_fp32_sum): after the all-to-all a rank holds `world` copies of ITS shard, one per peer, as rows
// of (world, per) bf16; out[i] = bf16(sum_r float(recv[r][i])) -- the sum accumulated in fp32 in rank order, ONE rounding
__global__ __launch_bounds__(256) void shard_sum_kernel(const bf16_t* __restrict__ recv, bf16_t* __restrict__ out, long per, int world) {
    const long nv = per >> 3, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < world; ++r) {
            float f[8];
            unpack8(ld<u32x4>(recv + (long)r * per + i * 8), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f[e];
        }
        st<u32x4>(out + i * 8, pack8(acc));
    }
}

}  // namespace

static int shard_sum_impl(const void* recv, void* out, int64_t per, int world, void* stream) {
    if (per <= 0 || world <= 0) return 0;
    if (!recv || !out) return E2K_ERR_ARG;
    if ((per & 7) || (((uintptr_t)recv | (uintptr_t)out) & 15)) return E2K_ERR_ALIGN;
    long gr = (per / 8 + 255) / 256; if (gr > 512) gr = 512; if (gr < 1) gr = 1;          // (a modest grid, as grad_pack: it runs NEXT to the backward pass)
    hipLaunchKernelGGL(shard_sum_kernel, dim3((int)gr), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)recv, (bf16_t*)out, (long)per, world);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int sumsq_f32_impl(const float* x, int64_t n, double* out, void* stream) {
    if (n <= 0) return 0;
    if (!x || !out) return E2K_ERR_ARG;
    if ((uintptr_t)x & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int adopt_step_impl(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                              const double* gsumsq, int step, int step_b, int active_b, const int32_t* ranges, int nranges,
                              float* ema, float ema_decay, void* stream) {
    if (n <= 0) return 0;
    if (!p || !g || !m || !v || step < 0 || step_b < 0 || nranges < 0 || nranges > 128 || (nranges && !ranges)) return E2K_ERR_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15 || ((uintptr_t)shadow_bf16 & 7)) return E2K_ERR_ALIGN;
    AdoptArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.shadow = (bf16_t*)shadow_bf16; a.n = n;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.max_norm = max_grad_norm;
    a.clamp = sqrtf(sqrtf((float)step));
    a.gsumsq = gsumsq; a.first = step == 0;
    a.ranges = ranges; a.nranges = nranges; a.first_b = step_b == 0; a.clamp_b = sqrtf(sqrtf((float)step_b)); a.active_b = active_b;
    a.ema = ema; a.ema_omd = 1.f - ema_decay;
    hipLaunchKernelGGL(adopt_kernel, dim3(adopt_grid_for(n)), dim3(256), 0, (hipStream_t)stream, a);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int ema_update_impl(float* ema, const float* p, int64_t n, float decay, void* stream) {
    if (n <= 0) return 0;
    if (!ema || !p) return E2K_ERR_ARG;
    if (((uintptr_t)ema | (uintptr_t)p) & 15) return E2K_ERR_ALIGN;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, ema, p, (long)n, 1.f - decay);
    E2K_CHECK_LAUNCH();
    return 0;
}

static int grad_pack_impl(const float* g, void* wire, int64_t n, float scale, void* stream) {
    if (n <= 0) return 0;
    if (!g || !wire) return E2K_ERR_ARG;
    if (((uintptr_t)g | (uintptr_t)wire) & 15) return E2K_ERR_ALIGN;
    // a modest grid: the exchange runs on a side stream NEXT to the backward pass and should not take the chip from it
    long gr = (n / 8 + 255) / 256; if (gr > 512) gr = 512; if (gr < 1) gr = 1;
    hipLaunchKernelGGL(grad_pack_kernel, dim3((int)gr), dim3(256), 0, (hipStream_t)stream, g, (bf16_t*)wire, (long)n, scale);
    E2K_CHECK_LAUNCH();
    return 0;
}
static int grad_unpack_impl(const void* wire, float* g, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (!g || !wire) return E2K_ERR_ARG;
    if (((uintptr_t)g | (uintptr_t)wire) & 15) return E2K_ERR_ALIGN;
    long gr = (n / 8 + 255) / 256; if (gr > 512) gr = 512; if (gr < 1) gr = 1;
    hipLaunchKernelGGL(grad_unpack_kernel, dim3((int)gr), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)wire, g, (long)n);
    E2K_CHECK_LAUNCH();
    return 0;
}

// ---- C ABI: every compute entry point goes through e2k::dispatch (plan.h) so that a launch plan can record it

extern "C" int e2k_sumsq_f32(const float* x, int64_t n, double* out, void* stream) {
    return e2k::dispatch("sumsq_f32", sumsq_f32_impl, x, n, out, stream);
}

extern "C" int e2k_adopt_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                              const double* gsumsq, int step, void* stream) {
    return e2k::dispatch("adopt_step", adopt_step_impl, p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, max_grad_norm, gsumsq, step, 0, 1, (const int32_t*)nullptr, 0, (float*)nullptr, 0.f, stream);
}

extern "C" int e2k_adopt_step_groups(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                                     const double* gsumsq, int step, int step_b, int active_b, const int32_t* ranges, int nranges,
                                     void* stream) {
    return e2k::dispatch("adopt_step_groups", adopt_step_impl, p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, max_grad_norm, gsumsq, step, step_b, active_b, ranges, nranges, (float*)nullptr, 0.f, stream);
}

extern "C" int e2k_adopt_step_ema(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                                  const double* gsumsq, int step, int step_b, int active_b, const int32_t* ranges, int nranges,
                                  float* ema, float ema_decay, void* stream) {
    if (!ema) return E2K_ERR_ARG;
    return e2k::dispatch("adopt_step_ema", adopt_step_impl, p, g, m, v, shadow_bf16, n, lr, beta1, beta2, eps, weight_decay, max_grad_norm, gsumsq, step, step_b, active_b, ranges, nranges, ema, ema_decay, stream);
}

extern "C" int e2k_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream) {
    return e2k::dispatch("ema_update", ema_update_impl, ema, p, n, decay, stream);
}

extern "C" int e2k_grad_pack_bf16(const float* g, void* wire, int64_t n, float scale, void* stream) {
    return e2k::dispatch("grad_pack_bf16", grad_pack_impl, g, wire, n, scale, stream);
}

extern "C" int e2k_grad_unpack_bf16(const void* wire, float* g, int64_t n, void* stream) {
    return e2k::dispatch("grad_unpack_bf16", grad_unpack_impl, wire, g, n, stream);
}

extern "C" int e2k_shard_sum_bf16(const void* recv_bf16, void* out_bf16, int64_t per, int world, void* stream) {
    return e2k::dispatch("shard_sum_bf16", shard_sum_impl, recv_bf16, out_bf16, per, world, stream);
}
