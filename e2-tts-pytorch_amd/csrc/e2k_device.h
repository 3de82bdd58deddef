// Device-side helpers shared by every kernel in csrc/ (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace e2k {

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }
// round-to-nearest-even, NaN kept quiet (same rule torch uses for float->bfloat16)
// (gfx950: one v_cvt_pk_bf16_f32 per pair)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    f32x2_ v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bflo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

template <class T> __device__ __forceinline__ T ld(const void* p) { return *reinterpret_cast<const T*>(p); }
template <class T> __device__ __forceinline__ void st(void* p, T v) { *reinterpret_cast<T*>(p) = v; }

// 8 bf16 (16 B) <-> 8 floats
__device__ __forceinline__ void unpack8(u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = bflo(v[i]); f[2 * i + 1] = bfhi(v[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2bf(f[2 * i], f[2 * i + 1]);
    return v;
}
__device__ __forceinline__ void unpack4(u32x2 v, float* f) {
    f[0] = bflo(v[0]); f[1] = bfhi(v[0]); f[2] = bflo(v[1]); f[3] = bfhi(v[1]);
}
__device__ __forceinline__ u32x2 pack4(const float* f) {
    u32x2 v; v[0] = pack2bf(f[0], f[1]); v[1] = pack2bf(f[2], f[3]); return v;
}

// rotary position embedding on one interleaved pair (x-transformers rotate_half on (d/2, 2) pairs; e2_tts.py:875,911): ONE definition
// with explicit fused multiply-adds, shared by qkv_post_fwd_kernel and the QKV GEMM's rotating epilogue (gemm.hip, nt_stage_readback
// EPI 2) so that the two produce the same bits.
// The second component is  fma(x0, s, x1 c)  and NOT  fma(x1, c, x0 s): the latter compiles (hipcc 7.2, -O3) to
// `v_pk_mul_f32 .. op_sel_hi:[0,1] neg_hi:[1,0]` -- a source broadcast from its low register, negated in the high lane only -- and the
// kernel with that form did not reproduce its own results next to kernels of another launch lane on MI355X (round 6: four models, every
// run; profiles/r06t_rotary_form_vs_lanes.txt).  tests/test_abi.py refuses the form anywhere in the library's device code.
__device__ __forceinline__ void rot_pair(float& x0, float& x1, float c, float s) {
    const float a = fmaf(x0, c, -(x1 * s)), b = fmaf(x0, s, x1 * c);
    x0 = a; x1 = b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
// tanh via one exp; exact to fp32 rounding for |x| < ~40, saturates cleanly beyond
__device__ __forceinline__ float tanhf_(float x) {
    float e = __expf(-2.0f * fabsf(x));
    float t = (1.0f - e) / (1.0f + e);
    return x < 0.f ? -t : t;
}

// stateless counter hash -> 32 random bits (dropout masks; restated bit-exactly in oracle/dropout_hash.py)
__device__ __forceinline__ unsigned fmix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
// per-(seed, stream) base (hoisted out of inner loops) and per-element hash: one multiply-add chain + one finaliser
__device__ __forceinline__ unsigned rand_base(unsigned seed, unsigned stream) { return fmix32(seed ^ (stream * 0x9e3779b1u)); }
__device__ __forceinline__ unsigned rand_at(unsigned base, unsigned row, unsigned col) {
    return fmix32(base + row * 0x85ebca77u + col * 0xc2b2ae3du);
}
__device__ __forceinline__ unsigned rand_u32(unsigned seed, unsigned stream, unsigned row, unsigned col) {
    return rand_at(rand_base(seed, stream), row, col);
}

// exact (erf) GELU and the GEGLU dropout keep factor: used by geglu_kernel (elementwise.hip) and by the GEGLU epilogue of
// the 256 x 256 NT GEMM (gemm.hip) -- one definition, so that the two produce the same bits
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float keep_scale(unsigned seed, unsigned stream, unsigned row, unsigned col, unsigned thresh, float inv_keep) {
    // one hash per pair of columns: low / high 16 bits
    unsigned h = rand_u32(seed, stream, row, col >> 1);
    unsigned r16 = (col & 1) ? (h >> 16) : (h & 0xffffu);
    return r16 >= thresh ? inv_keep : 0.f;
}

// ---- one-wave-per-row access: lane owns VEC consecutive elements in each of NCH chunks of 64*VEC (D = 64*VEC*NCH)
template <int VEC> __device__ __forceinline__ void load_vec(const bf16_t* p, float* f);
template <> __device__ __forceinline__ void load_vec<8>(const bf16_t* p, float* f) { unpack8(ld<u32x4>(p), f); }
template <> __device__ __forceinline__ void load_vec<4>(const bf16_t* p, float* f) { unpack4(ld<u32x2>(p), f); }
template <> __device__ __forceinline__ void load_vec<2>(const bf16_t* p, float* f) { unsigned u = ld<unsigned>(p); f[0] = bflo(u); f[1] = bfhi(u); }
template <int VEC> __device__ __forceinline__ void store_vec(bf16_t* p, const float* f);
template <> __device__ __forceinline__ void store_vec<8>(bf16_t* p, const float* f) { st<u32x4>(p, pack8(f)); }
template <> __device__ __forceinline__ void store_vec<4>(bf16_t* p, const float* f) { st<u32x2>(p, pack4(f)); }
template <> __device__ __forceinline__ void store_vec<2>(bf16_t* p, const float* f) { st<unsigned>(p, pack2bf(f[0], f[1])); }

template <int VEC, int NCH> __device__ __forceinline__ void load_row(const bf16_t* row, int lane, float* f) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_vec<VEC>(row + c * 64 * VEC + lane * VEC, f + c * VEC);
}
template <int VEC, int NCH> __device__ __forceinline__ void store_row(bf16_t* row, int lane, const float* f) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) store_vec<VEC>(row + c * 64 * VEC + lane * VEC, f + c * VEC);
}
// packed (raw bf16 pairs) row loads for software prefetch: half the registers of the unpacked floats
template <int VEC, int NCH> __device__ __forceinline__ void load_raw_row(const bf16_t* row, int lane, unsigned* w) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const bf16_t* q = row + c * 64 * VEC + lane * VEC;
        if (VEC == 8) { u32x4 v = ld<u32x4>(q); w[c * 4] = v[0]; w[c * 4 + 1] = v[1]; w[c * 4 + 2] = v[2]; w[c * 4 + 3] = v[3]; }
        else if (VEC == 4) { u32x2 v = ld<u32x2>(q); w[c * 2] = v[0]; w[c * 2 + 1] = v[1]; }
        else w[c] = ld<unsigned>(q);
    }
}
template <int N> __device__ __forceinline__ void unpack_raw_row(const unsigned* w, float* f) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) { f[2 * i] = bflo(w[i]); f[2 * i + 1] = bfhi(w[i]); }
}
template <int VEC, int NCH> __device__ __forceinline__ void load_row_f32(const float* row, int lane, float* f) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) f[c * VEC + v] = row[c * 64 * VEC + lane * VEC + v];
}

}  // namespace e2k

#define E2K_ROW_DISPATCH(D, FN, ...)                              \
    switch (D) {                                                  \
        case 128: rc = FN<2, 1>(__VA_ARGS__); break;              \
        case 256: rc = FN<4, 1>(__VA_ARGS__); break;              \
        case 384: rc = FN<2, 3>(__VA_ARGS__); break;              \
        case 512: rc = FN<8, 1>(__VA_ARGS__); break;              \
        case 640: rc = FN<2, 5>(__VA_ARGS__); break;              \
        case 768: rc = FN<4, 3>(__VA_ARGS__); break;              \
        case 896: rc = FN<2, 7>(__VA_ARGS__); break;              \
        case 1024: rc = FN<8, 2>(__VA_ARGS__); break;             \
        case 1280: rc = FN<4, 5>(__VA_ARGS__); break;             \
        case 1792: rc = FN<4, 7>(__VA_ARGS__); break;             \
        case 1536: rc = FN<8, 3>(__VA_ARGS__); break;             \
        case 2048: rc = FN<8, 4>(__VA_ARGS__); break;             \
        default: rc = E2K_ERR_SHAPE;                              \
    }

#define E2K_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return 1000 + (int)e__;       \
    } while (0)
