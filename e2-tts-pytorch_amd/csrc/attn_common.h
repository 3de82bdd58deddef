// Shared by the attention translation units (attn.hip: layout kernels, register-staged fallback kernels, first-generation ring
// kernels, C ABI; attn32.hip: the 32-rows-per-wave ring kernels): arguments, soft-clamp arithmetic, dropout counter hash,
// XCD-aware workgroup numbering.  Everything here is internal to those two files (anonymous namespace on purpose).
#pragma once
#include "e2k_device.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"

namespace {

using namespace e2k;

constexpr int RKM = 4096;             // bytes of LDS the ring kernels reserve for the key mask of a batch row (Npad <= 4096)

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float CLAMP = 50.f;
constexpr float NEG_BIG = -1e30f;     // initial running max
constexpr float NEG_MASK = -2e30f;    // masked score (below NEG_BIG so that exp2(masked - max) is 0 even before any valid key)

// Dropout keep decisions: four 16-bit uniform samples per hash for the keys 4j .. 4j+3 of query row q
// (hrow = rand_base(seed, stream) + q * 0x85ebca77 is hoisted by the caller; key j keeps iff its sample >= thresh;
// oracle/dropout_hash.py restates this bit for bit).  Round 4: the two 64-bit words come from three 24 x 24 -> 32-bit
// multiply-adds (v_mad_u32_u24, a full-rate instruction; the 32-bit v_mul_lo_u32 of the murmur finaliser used before issues at a
// quarter of that rate and was 13 % of the forward's vector-ALU clocks) with xor-shifts in between; the addend of each multiply
// carries the bits the 24-bit multiplicand drops.  Avalanche (every input bit flips every output bit with probability 0.5 +-
// 0.007 over 1e5 inputs) and the grid statistics of the masks (keep rate, sample / decision correlations between neighbouring
// keys and queries, per-row and per-column drop-count variance against the binomial) equal the finaliser's: tools/probes/drop_hash_stats.py.
__device__ __forceinline__ unsigned mad24(unsigned x, unsigned y, unsigned z) { return (x & 0xffffffu) * (y & 0xffffffu) + z; }      // (selected as v_mad_u32_u24)
__device__ __forceinline__ void drop4(unsigned hrow, unsigned key4, unsigned& w0, unsigned& w1) {
    const unsigned h = hrow + key4 * 0xc2b2ae3du;
    unsigned x = h ^ (h >> 16);
    x = mad24(x, 0x85ebcbu, h >> 8);
    x ^= x >> 13;
    unsigned a = mad24(x, 0xc2b2afu, x >> 11);
    a ^= a >> 15;
    unsigned b = mad24(a, 0x9e3779u, x >> 7);
    b ^= b >> 12;
    w0 = a;
    w1 = b;
}
__device__ __forceinline__ unsigned drop_sample(unsigned w0, unsigned w1, int j) {      // j = key & 3
    const unsigned w = (j & 2) ? w1 : w0;
    return (j & 1) ? (w >> 16) : (w & 0xffffu);
}
// stream of the counter hash for (attention call, batch * head): top bit set, so that it can never meet a GEGLU stream
// (those are the plain call ids) whatever the batch size
__device__ __forceinline__ unsigned attn_stream(unsigned stream_id, unsigned bh) { return 0x80000000u | (stream_id << 16) | bh; }

// soft-clamp: tanh(raw * scale / 50) via one v_exp_f32 and one v_rcp_f32: tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1)
__device__ __forceinline__ float clamp_tanh(float raw, float k2) { return 1.f - 2.f * fast_rcp(fast_exp2(raw * k2) + 1.f); }
// cl2 * tanh(.) in one fma after the rcp
__device__ __forceinline__ float clamp_tanh_scaled(float raw, float k2, float cl2) { return fmaf(-2.f * cl2, fast_rcp(fast_exp2(raw * k2) + 1.f), cl2); }

// Polynomial tanh for |x| <= 0.75 (max relative error 3.1e-5, minimax fit in x^2), two values per packed-fp32
// instruction and no transcendental: soft-clamp arguments are raw * scale / 50, i.e. |raw * scale| <= 37.5 -- every
// realistic attention logit.  Each 64-key tile takes this path only when a wave vote says all of its scores are in
// range; otherwise the exp2 / rcp form above runs (same result to fp32 rounding).  The input scale kx = scale / 50 and
// an output factor `out` (log2(e) * 50 in the forward, 1 in the backward) are folded into the coefficients:
//   out * tanh(s kx) = s (a0 + a1 w + a2 w^2 + a3 w^3),  w = s^2,  a_i = out * kx^(2i+1) * c_i        (5 packed instructions)
typedef float f32x2_ __attribute__((ext_vector_type(2)));
constexpr float TANH_POLY_MAX = 0.75f;
struct ClampPoly { float a0, a1, a2, a3; };
__device__ __forceinline__ ClampPoly clamp_poly(float kx, float out) {
    const float k2 = kx * kx, k1 = out * kx;
    return ClampPoly{k1 * 0.999968926f, k1 * k2 * -0.332331483f, k1 * k2 * k2 * 0.125959146f, k1 * k2 * k2 * k2 * -0.0338411346f};
}
__device__ __forceinline__ f32x2_ clamp2(f32x2_ s, const ClampPoly& c) {
    const f32x2_ w = s * s;
    f32x2_ pl = w * c.a3 + c.a2;
    pl = pl * w + c.a1;
    pl = pl * w + c.a0;
    return s * pl;
}
// largest |score| of a lane's 16 scores (v_max3_f32 chains)
__device__ __forceinline__ float abs_max16(const f32x4 (&s)[4]) {
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a = fmaxf(fmaxf(fabsf(s[t][0]), fabsf(s[t][1])), a);
        a = fmaxf(fmaxf(fabsf(s[t][2]), fabsf(s[t][3])), a);
    }
    return a;
}

// the 8 mask bytes (each 0 / 1) of a lane's keys -> 8 bits
__device__ __forceinline__ unsigned mask_bits(unsigned long long m) { return (unsigned)((m * 0x0102040810204080ull) >> 56); }

struct AttnArgs {
    const bf16_t *Q, *K, *V, *QT, *KT, *VT;   // (B,h,N,64) / (B,h,64,Npad)
    const uint8_t* kmask;                      // (B, Npad), 0 beyond N
    const float* gate;                         // (B,h,N)
    bf16_t* O; bf16_t* Og;                     // (B*N, h*64) token-major: un-gated / gated (+query-masked)
    float* lse2;                               // (B,h,N)  log2-domain log-sum-exp
    int B, H, N, Npad;
    float scale; unsigned seed, stream_id, thresh; float inv_keep;
    const unsigned* seed_dev;                  // if set, the dropout seed is read from device memory (plan replay)
    int probe;                                 // E2K_ATTN_PROBE_* bits (bottleneck probes of the forward: results are wrong on purpose)
    // optional: keep decisions of the dropout as ballot words, written by the forward and read by the backward instead
    // of re-hashing: [b*h][key tile][query tile][wave 0..3][slot 4t+r] uint64, bit (16 g + l15) = lane of the forward
    unsigned long long* dropbits;
    // backward
    const bf16_t* dOg;                         // (B*N, h*64)
    bf16_t* dO; bf16_t* dOT;                   // head-major / transposed
    float* delta; float* dgate_pre;            // (B,h,N)
    bf16_t *dQ, *dK, *dV;                      // (B,h,N,64)
    int xcd_map;                               // ring kernels: re-number the workgroups so that a (batch, head) row stays on one XCD
};

// The ring kernels run on a 1-D grid of (64-row tiles) x heads x batch.  Workgroups are handed to the 8 XCDs round-robin in launch
// order, so with the plain numbering the 17 tiles of a (batch, head) row -- which all stream the SAME K / V (the backward: Q / dO)
// tiles -- sit on all 8 XCDs and every L2 holds every row in flight (60 rows x 270 KB against 4 MB).  Re-numbered, an XCD gets a
// contiguous run of rows: 7-8 rows in flight per L2, each tile of K / V fetched once per row instead of once per XCD.
struct RingWG { int x, h, b; };
__device__ __forceinline__ RingWG ring_wg(const AttnArgs& p, int nx) {
    const int nwg = (int)gridDim.x;
    int lin = (int)blockIdx.x;
    if (p.xcd_map) {
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    }
    RingWG w;
    w.x = lin % nx;
    const int bh = lin / nx;
    w.h = bh % p.H;
    w.b = bh / p.H;
    return w;
}

}  // namespace
