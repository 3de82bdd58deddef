// Attention ring kernels, second generation: 32 rows per wave on 32x32x16 MFMAs (forward, dQ, dK/dV).
//
// Same arithmetic as attn.hip's first-generation ring kernels (x_transformers.Attention with softclamp_logits / gate_value_heads as
// the reference calls it, e2_tts.py:641,689,875,911; oracle Attention): S = 50 tanh(q.k / 8 / 50), key mask, fp32 softmax without a
// running maximum (the soft-clamp bounds the logits), dropout on the probabilities from the counter hash, O = P V, head gates.
// What changed is the formulation per score:
//   * a wave owns 32 rows (queries in the forward / dQ kernels, keys in the dK,dV kernel) instead of 16: one 8-KB LDS tile read
//     feeds 8 MFMAs of 32x32x16 (512 matrix-pipe clocks) instead of 16 of 16x16x32 (256): half the LDS bytes, half the LDS-DMA
//     bytes (128-row workgroups) and half the barriers per score;
//   * swapped products (S^T = K Q^T): lane (j = l & 31, hi = l >> 5) holds column j -- ONE query -- and 16 of the 32 rows of each
//     block, so row sums are plain per-lane adds (one cross-half add at the very end);
//   * the rows of the K-type LDS tiles are stored with bits 2 and 3 of the row index swapped, which makes a lane's accumulator
//     registers 8s .. 8s+7 exactly the 8 consecutive reduction slots (keys 16s + 8hi .. + 7) the next MFMA wants as its B operand:
//     P goes from the exp2 to the second MFMA through v_cvt_pk_bf16_f32 only, no cross-lane step, no LDS;
//   * soft-clamp tanh as a CUBIC (3 instructions) when every |logit| of the block is <= 7.5, the degree-7 polynomial up to 37.5,
//     the exp2 / rcp form beyond (wave votes);
//   * 16-byte epilogue stores (v_permlane32_swap pairs the two half-waves' 8-byte pieces).
// LDS images: a [64 rows][64 bf16] tile is 64 rows of eight 16-byte chunks; chunk c of row r sits at position c ^ ((r >> 1) & 7),
// conflict-free for the ds_read_b128 fragment reads (lane i reads row i, chunk 2 ks + hi).  Produced by permuting the per-lane
// SOURCE address of the LDS-DMA (the destination of a global_load_lds is lane-linear).
#include "e2k_device.h"
#include "plan.h"
#include <e2k_asm.h>
#include "../../include/e2k.h"
#include "attn_common.h"
#include "attn32.h"

using namespace e2k;

namespace {

constexpr int FSTAGE = 16384;         // one ring stage: two [64][64] bf16 tiles

// LDS row of a K-type tile that holds tile row R (and back: an involution): bits 2 and 3 trade places
__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

// soft-clamp tiers (wave-uniform): 0 cubic, 1 degree 7, 2 exp2 / rcp
constexpr float TANH_CUBIC_MAX = 0.15f;      // |logit| <= 7.5: max relative error of the cubic 8.4e-6 (the degree-7 fit: 1.1e-5 up to 0.75)
struct Clamp32 {
    float c0, c1;               // tier 0: out * tanh(s kx) = s (c0 + c1 s^2)
    ClampPoly p7;               // tier 1
    float k2, cl;               // tier 2: cl * tanh via exp2(s k2)
};
__device__ __forceinline__ Clamp32 clamp32(float kx, float out) {
    Clamp32 c;
    c.c0 = out * kx * 0.99999162f;              // minimax (relative) fit of tanh(x) / x in x^2 on |x| <= 0.15
    c.c1 = out * kx * kx * kx * -0.3303569f;
    c.p7 = clamp_poly(kx, out);
    c.k2 = 2.f * LOG2E * kx;
    c.cl = out;
    return c;
}
// two scores per packed-fp32 instruction (v_pk_mul_f32 / v_pk_fma_f32: the vector ALU is 16 lanes wide, a wave64 instruction takes
// 4 clocks whether it carries one fp32 operation per lane or two)
template <int TIER>
__device__ __forceinline__ f32x2_ clamp_eval2(f32x2_ s, const Clamp32& c) {
    if (TIER == 0) {
        const f32x2_ w = s * s;
        return s * (w * c.c1 + c.c0);
    } else if (TIER == 1) {
        return clamp2(s, c.p7);
    } else {
        return f32x2_{clamp_tanh_scaled(s[0], c.k2, c.cl), clamp_tanh_scaled(s[1], c.k2, c.cl)};
    }
}
__device__ __forceinline__ float abs_max_16(const f32x16& s, float a) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) a = fmaxf(fmaxf(fabsf(s[r]), fabsf(s[r + 1])), a);
    return a;
}
__device__ __forceinline__ int clamp_tier(float am, float kx) {
    if (wave_all(am * kx <= TANH_CUBIC_MAX)) return 0;
    return wave_all(am * kx <= TANH_POLY_MAX) ? 1 : 2;
}

// the upper half-wave's x and the lower half-wave's y trade places (two dwords each)
__device__ __forceinline__ void swap_pair(u32x2& x, u32x2& y) {
    unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
    lane32_swap(x0, y0);
    lane32_swap(x1, y1);
    x = u32x2{x0, x1};
    y = u32x2{y0, y1};
}
__device__ __forceinline__ bf16x8 pack8f(const float* f) { return __builtin_bit_cast(bf16x8, pack8(f)); }

// the 16 x 4 fragment-read offsets of a lane inside a [64][64] tile: chunk 2 j + hi of row l31 (+ 32 per row block, added as an
// immediate)
__device__ __forceinline__ void frag_offsets(int l31, int hi, int (&off)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) off[j] = l31 * 128 + (((2 * j + hi) ^ swz(l31)) << 4);
}

// row-per-lane epilogue: acc[db][r] = column (32 db + (r & 3) + 8 (r >> 2) + 4 hi) of this lane's row, scaled and stored as bf16 in
// 16-byte pieces: the two half-waves exchange their 8-byte pieces (v_permlane32_swap), lower half stores columns 16 g .. + 7, upper half
// 16 g + 8 .. + 15 of every group of 16 columns
template <class F>
__device__ __forceinline__ void store_rows32(const f32x16 (&acc)[2], int hi, bool ok, F&& put) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            float a[4] = {acc[db][4 * g], acc[db][4 * g + 1], acc[db][4 * g + 2], acc[db][4 * g + 3]};
            float b[4] = {acc[db][4 * g + 4], acc[db][4 * g + 5], acc[db][4 * g + 6], acc[db][4 * g + 7]};
            put(db, g, hi, ok, a, b);
        }
}

// ------------------------------------------------------------------------------------------------ forward

// The 16 scores of one 32-key block of a lane (s[r]: key 16 (r >> 3) + 8 hi + (r & 7) of the block) -> soft-clamp, exp2, row sums,
// dropout, the two packed B operands of the second MFMA; mk[r]: the keep decisions of score r as a wave ballot.
//   hk: counter of the block's first group of four keys for this lane; kmb: key-mask bits, bit 16 s2 + e <-> score 8 s2 + e
template <int TIER, bool DROP, bool SHARE, bool MASKED, int PROBE>
__device__ __forceinline__ void fwd_block(const f32x16& s, int kb, const Clamp32& cc, f32x2_ (&lsum2)[2], unsigned hk, unsigned thresh,
                                          unsigned kmb, bf16x8 (&pf)[2], unsigned long long (&mk)[16]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        float pr[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2_ z = clamp_eval2<TIER>(f32x2_{s[8 * s2 + e], s[8 * s2 + e + 1]}, cc);
            if (MASKED) {           // masked keys (only the last tile or two): exp2 gives an exact 0
                z[0] = ((kmb >> (16 * s2 + e)) & 1u) ? z[0] : NEG_MASK;
                z[1] = ((kmb >> (16 * s2 + e + 1)) & 1u) ? z[1] : NEG_MASK;
            }
            pr[e] = (PROBE & 1) ? z[0] : fast_exp2(z[0]);
            pr[e + 1] = (PROBE & 1) ? z[1] : fast_exp2(z[1]);
            lsum2[(e >> 1) & 1] += f32x2_{pr[e], pr[e + 1]};          // softmax denominators are taken BEFORE dropout
        }
        if (DROP) {
            // keys 32 kb + 16 s2 + 8 hi + e of the tile: two groups of four consecutive keys = two counter values
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned w0, w1;
                if (PROBE & 2) { w0 = hk + (unsigned)(4 * s2 + half) * 0x10001u; w1 = hk ^ 0x55aa1234u; }
                else drop4(hk, (unsigned)(4 * s2 + half), w0, w1);
                const bool kp[4] = {(w0 & 0xffffu) >= thresh, (w0 >> 16) >= thresh, (w1 & 0xffffu) >= thresh, (w1 >> 16) >= thresh};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = 4 * half + j;
                    pr[e] = kp[j] ? pr[e] : 0.f;
                    if (SHARE) mk[8 * s2 + e] = wave_ballot(kp[j]);
                }
            }
        }
        pf[s2] = pack8f(pr);
    }
}

// One workgroup = 128 query rows of one (batch, head): 4 waves x 32 rows.  K / V^T tiles of 64 keys go HBM -> LDS by global_load_lds
// into a 2-stage ring (counted wait + ONE raw barrier per tile, the next tile in flight during the computation of this one).
// PROBE (E2K_ATTN32_PROBE, bottleneck probes: WRONG RESULTS on purpose): 1 no exp2, 2 no counter hash, 4 no score MFMAs, 8 no LDS fragment
// reads, 16 no output MFMAs, 32 no LDS-DMA after the first two tiles, 64 no barriers
template <bool DROP, bool SHARE, int PROBE = 0>
__global__ __launch_bounds__(256, 4) void attn_fwd32_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FSTAGE + RKM];
    lds_declare(smem, sizeof(smem));
    unsigned char* const kms = smem + 2 * FSTAGE;                 // key mask of this batch row: one byte of bits per 8 keys
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = (p.N + 63) / 64, nq = (p.N + 127) / 128, nqb = (p.N + 31) / 32;
    const RingWG wg = ring_wg(p, nq);
    const int h = wg.h, b = wg.b;
    const long bh = (long)b * p.H + h;
    // (the partly filled last tile of a row -- one live wave of four at N = 1056 -- as a launch of its own whose waves split the keys of the
    //  live block was measured: 95.9 us against 90 for this single launch, in which those workgroups overlap the end of the others;
    //  profiles/r05h_attn32_tail_split_ab.json)
    const int qb = wg.x * 4 + wave;                               // 32-row block of this wave
    const int q = qb * 32 + l31;
    const bool qin = q < p.N;
    const bool live = qb * 32 < p.N;                              // (wave-uniform) a wave past the end only stages tiles
    const unsigned dstream = attn_stream(p.stream_id, (unsigned)bh);

    for (int j = tid; j < p.Npad / 8; j += 256) kms[j] = (unsigned char)mask_bits(ld<unsigned long long>(p.kmask + (long)b * p.Npad + j * 8));
    wait_lgkm0();                 // (the first barrier of the tile loop is a raw one: it publishes what has been WRITTEN)

    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = qin ? ld<bf16x8>(p.Q + (bh * p.N + q) * DH + ks * 16 + hi * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};

    // staging: a tile = 8 wave instructions of 8 LDS rows; wave w issues instructions 2w, 2w + 1 of K and of V^T
    const bf16_t* Kbase = p.K + bh * p.N * DH;
    const bf16_t* VTbase = p.VT + bh * DH * p.Npad;
    int krow[2];
    unsigned kcol[2], voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (wave * 2 + u) * 8 + (lane >> 3), c = (lane & 7) ^ swz(r);
        krow[u] = swap23(r);
        kcol[u] = (unsigned)(c * 16);
        voff[u] = (unsigned)(((long)r * p.Npad + c * 8) * 2);
    }
    auto issue = [&](int t, int stage) __attribute__((always_inline)) {
        const int k0 = t * 64;
        unsigned char* S = smem + stage * FSTAGE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // keys past the end: any valid row (masked below).  Wave-uniform bases + 32-bit per-lane offsets (no 64-bit per-lane pointers
            // kept across the loop)
            const unsigned koff = (unsigned)min(k0 + krow[u], p.N - 1) * (DH * 2) + kcol[u];
            glds16((const char*)Kbase + koff, S + (wave * 2 + u) * 1024);
            glds16((const char*)VTbase + ((unsigned)k0 * 2 + voff[u]), S + 8192 + (wave * 2 + u) * 1024);
        }
    };
    int foff[4];
    frag_offsets(l31, hi, foff);

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    f32x2_ lsum2[2] = {f32x2_{0.f, 0.f}, f32x2_{0.f, 0.f}};
    const float kx = p.scale / CLAMP;
    const Clamp32 cc = clamp32(kx, CLAMP * LOG2E);
    const unsigned hrow = rand_base(p.seed_dev ? *p.seed_dev : p.seed, dstream) + (unsigned)q * 0x85ebca77u;
    const unsigned hlane = hrow + (unsigned)(2 * hi) * 0xc2b2ae3du;

    // every ordinary global load of the prologue is waited for BEFORE the first LDS-DMA (vmcnt is counted in order)
    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(hrow));
    issue(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        const int k0 = kt * 64;
        wait_vmcnt<0>();                  // tile kt has landed for this wave ...
        if (!(PROBE & 64)) barrier_raw(); // ... and for every wave; everyone has finished tile kt - 1
        if (kt + 1 < nt && !((PROBE & 32) && kt >= 1)) issue(kt + 1, (kt + 1) & 1);
        if (!live) continue;
        const unsigned char* Kt = smem + (kt & 1) * FSTAGE;
        const unsigned char* Vt = Kt + 8192;
        // key-mask bits of this lane's keys: byte 4 kb + 2 s2 + hi of the tile's eight
        const unsigned long long km8 = ld<unsigned long long>(kms + (k0 >> 3)) >> (8 * hi);
        const bool allk = wave_all((km8 & 0x00ff00ff00ff00ffull) == 0x00ff00ff00ff00ffull);
        // one 32-key block at a time from the score MFMAs to the output MFMAs (16 score registers live instead of 32)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = (PROBE & 8) ? qf[(ks + 1) & 3] : ld<bf16x8>(Kt + foff[ks] + kb * 4096);
                if (PROBE & 4) s[ks] += __uint_as_float(__builtin_bit_cast(u32x4, kf)[0] & 0x3fffffffu);
                else s = mfma32(kf, qf[ks], s);
            }
            bf16x8 pf[2];
            unsigned long long mk[16];
            const int tier = clamp_tier(abs_max_16(s, 0.f), kx);
            const unsigned kmb = (unsigned)(km8 >> (32 * kb));
            const unsigned hk = hlane + ((unsigned)(k0 >> 2) + 8 * kb) * 0xc2b2ae3du;
            if (allk) {
                if (tier == 0) fwd_block<0, DROP, SHARE, false, PROBE>(s, kb, cc, lsum2, hk, p.thresh, kmb, pf, mk);
                else if (tier == 1) fwd_block<1, DROP, SHARE, false, PROBE>(s, kb, cc, lsum2, hk, p.thresh, kmb, pf, mk);
                else fwd_block<2, DROP, SHARE, false, PROBE>(s, kb, cc, lsum2, hk, p.thresh, kmb, pf, mk);
            } else {                    // tiles with masked keys (the last one or two of a sequence): one generic path
                fwd_block<2, DROP, SHARE, true, PROBE>(s, kb, cc, lsum2, hk, p.thresh, kmb, pf, mk);
            }
            if (DROP && SHARE) {
                // the compare masks of the block (SGPR pairs) leave through the scalar data cache: no vector instruction
                unsigned long long* dw = p.dropbits + (((long)bh * nt + kt) * nqb + qb) * 32 + 16 * kb;
                sstore_masks16(dw, mk);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = (PROBE & 8) ? qf[2 * s2 + db] : ld<bf16x8>(Vt + foff[2 * kb + s2] + db * 4096);
                    if (PROBE & 16) o[db][s2] += __uint_as_float((__builtin_bit_cast(u32x4, vf)[0] ^ __builtin_bit_cast(u32x4, pf[s2])[db]) & 0x3fffffffu);
                    else o[db] = mfma32(vf, pf[s2], o[db]);
                }
            }
        }
    }
    if (DROP && SHARE) sstore_flush();     // (every wave: write the scalar data cache back before the kernel ends)
    float lsum = (lsum2[0][0] + lsum2[0][1]) + (lsum2[1][0] + lsum2[1][1]);
    if (!live) return;
    lsum += lane32_other(lsum);                 // a row's keys are split over lanes l and l + 32
    const float inv = lsum > 0.f ? (DROP ? p.inv_keep : 1.f) / lsum : 0.f;
    const float gt = qin ? p.gate[bh * p.N + q] : 0.f;
    const bool qkeep = qin && p.kmask[(long)b * p.Npad + q] != 0;
    if (qin && hi == 0) p.lse2[bh * p.N + q] = log2f(fmaxf(lsum, 1e-37f));
    const long orow = ((long)b * p.N + (qin ? q : 0)) * ((long)p.H * DH) + h * DH;
    store_rows32(o, hi, qin, [&](int db, int g, int hi_, bool ok, const float* a, const float* c) __attribute__((always_inline)) {
        float v[8], vg[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = qkeep ? a[e] * inv : 0.f;
            v[4 + e] = qkeep ? c[e] * inv : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) vg[e] = v[e] * gt;
        // v[0..3]: columns 8 g + 4 hi + 0..3 (+ 32 db), v[4..7]: the same of group g + 1
        u32x2 x = pack4(v), y = pack4(v + 4), xg = pack4(vg), yg = pack4(vg + 4);
        swap_pair(x, y);
        swap_pair(xg, yg);
        // lower half: own x (columns 8 g .. + 3) + the upper half's x (8 g + 4 .. + 7); upper half: the lower half's y (8 (g + 1) .. + 3) + own y
        const int col = 32 * db + 8 * (g + hi_);
        if (ok) {
            st<u32x4>(p.O + orow + col, u32x4{x[0], x[1], y[0], y[1]});
            st<u32x4>(p.Og + orow + col, u32x4{xg[0], xg[1], yg[0], yg[1]});
        }
    });
}

// ------------------------------------------------------------------------------------------------ backward: dQ

// Transposing fragment reads (ds_read_b64_tr_b16) out of a K-type tile image: the A operand of an MFMA whose rows are the tile's 64
// COLUMNS (dh) and whose reduction slots are tile rows.  Lane l = (i = l & 31, hi): row block db gives MFMA row i = column 32 db + i;
// its eight slots 8 hi .. 8 hi + 7 of slab (kb, s2) are the tile rows 32 kb + 16 s2 + 8 hi + e, which sit at LDS rows
// 32 kb + 16 s2 + 4 hi + (e & 3) + 8 (e >> 2) (swap23).  One instruction serves a 16-lane group: lane (q16 = l & 15) passes the
// address of 4 consecutive columns 4 (q16 & 3) .. of LDS row (q16 >> 2) of the group's [4 rows][16 columns] block and receives
// column q16 of the four rows.  The group of lanes 16 .. 31 (and 48 .. 63) covers columns 16 .. 31 of the row block.
struct TrOff { int lo[2], hi_[2]; };      // [db]: byte offsets (inside the tile) of the reads for e = 0..3 and e = 4..7, slab (0, 0)
__device__ __forceinline__ TrOff tr_offsets(int lane) {
    const int q16 = lane & 15, cg = (lane >> 4) & 1, hi = lane >> 5;
    TrOff t;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int col = 32 * db + 16 * cg + 4 * (q16 & 3);           // first of the 4 columns this lane addresses
#pragma unroll
        for (int eh = 0; eh < 2; ++eh) {
            const int row = 4 * hi + (q16 >> 2) + 8 * eh;
            const int off = row * 128 + ((((col >> 3)) ^ swz(row)) << 4) + (col & 7) * 2;
            (eh ? t.hi_ : t.lo)[db] = off;
        }
    }
    return t;
}
// rows + 16 s2 + 32 kb: swz(row + 16 s2 + 32 kb) = swz(row) ^ ... only when 16 s2 >> 1 = 8 s2 leaves (row >> 1) & 7 alone: it does (8, 16 = 0 mod 8)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* T, const TrOff& t, int db, int imm) {
    s16x4_ a, b;
    lds_tr_issue(a, T + t.lo[db], imm);
    lds_tr_issue(b, T + t.hi_[db], imm);
    lds_tr_wait(a, b);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// soft-clamp tanh of the 16 scores of a block: three wave-uniform tiers (clamp_tier), out-of-line so that the rest of the block's
// arithmetic exists once
__device__ __forceinline__ void clamp_block(const f32x16& s, int tier, const Clamp32& cc, f32x2_ (&th)[8]) {
    if (tier == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) th[r] = clamp_eval2<0>(f32x2_{s[2 * r], s[2 * r + 1]}, cc);
    } else if (tier == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) th[r] = clamp_eval2<1>(f32x2_{s[2 * r], s[2 * r + 1]}, cc);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) th[r] = clamp_eval2<2>(f32x2_{s[2 * r], s[2 * r + 1]}, cc);
    }
}
// keeps a rarely taken, wave-uniform branch a branch (left alone the compiler turns its selects into unconditional ones on the hot path)
__device__ __forceinline__ void cold_path() { asm volatile("" ::: "memory"); }

// 16 scores (th = tanh of them) + 16 dP of one 32-key block of a lane -> dS^T (packed, two slabs of 8 keys) -> dQ^T += K^T dS^T
//   dw: the forward's 16 compare masks of the block (SHARE); Kb: the block's 32 rows of the K tile image
template <bool DROP, bool SHARE>
__device__ __forceinline__ void dq_block(const f32x2_ (&th)[8], const f32x16& dp, float cl2, float lse, float dl, float inv_keep,
                                         unsigned hk, unsigned thresh, const unsigned long long* dw, bool allk, unsigned kmb, const unsigned char* Kb,
                                         const TrOff& tro, f32x16 (&dq)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        float dsv[8];
        bool kp[8];
        if (DROP && !SHARE) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned w0, w1;
                drop4(hk, (unsigned)(4 * s2 + half), w0, w1);
                kp[4 * half] = (w0 & 0xffffu) >= thresh;
                kp[4 * half + 1] = (w0 >> 16) >= thresh;
                kp[4 * half + 2] = (w1 & 0xffffu) >= thresh;
                kp[4 * half + 3] = (w1 >> 16) >= thresh;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int r = 8 * s2 + e;
            const f32x2_ t = th[r >> 1];
            const f32x2_ arg = t * cl2 - lse;
            const f32x2_ pv = {fast_exp2(arg[0]), fast_exp2(arg[1])};
            f32x2_ t1 = {dp[r], dp[r + 1]};
            if (DROP) {
                const bool k0 = SHARE ? wave_inverse_ballot(sload64(dw + r)) : kp[e];
                const bool k1 = SHARE ? wave_inverse_ballot(sload64(dw + r + 1)) : kp[e + 1];
                t1[0] = k0 ? t1[0] : 0.f;
                t1[1] = k1 ? t1[1] : 0.f;
                t1 = t1 * inv_keep - dl;
            } else {
                t1 = t1 - dl;
            }
            const f32x2_ t2 = 1.f - t * t;
            const f32x2_ ds = (pv * t2) * t1;
            dsv[e] = ds[0];
            dsv[e + 1] = ds[1];
        }
        if (!allk) {            // masked keys contribute nothing (only the last tile or two of a sequence)
            cold_path();
#pragma unroll
            for (int e = 0; e < 8; ++e) dsv[e] = ((kmb >> (16 * s2 + e)) & 1u) ? dsv[e] : 0.f;
        }
        const bf16x8 df = pack8f(dsv);
#pragma unroll
        for (int db = 0; db < 2; ++db) dq[db] = mfma32(tr_frag(Kb, tro, db, s2 * 2048), df, dq[db]);
    }
}

// dQ: same sweep and lane layout as the forward (lane = one query, 16 + 16 keys of each 32-key block); K and V tiles (both as
// K-type images) in a 2-stage LDS-DMA ring; dS^T feeds dQ^T = K^T dS^T with K^T read out of the row-major K tile by transposing reads.
template <bool DROP, bool SHARE, int WPS>          // WPS: waves per SIMD the register budget is compiled for
__global__ __launch_bounds__(256, WPS) void attn_dq32_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FSTAGE + RKM];
    lds_declare(smem, sizeof(smem));
    unsigned char* const kms = smem + 2 * FSTAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = (p.N + 63) / 64, nq = (p.N + 127) / 128, nqb = (p.N + 31) / 32;
    const RingWG wg = ring_wg(p, nq);
    const int h = wg.h, b = wg.b;
    const long bh = (long)b * p.H + h;
    const int qb = wg.x * 4 + wave;
    const int q = qb * 32 + l31;
    const bool qin = q < p.N;
    const bool live = qb * 32 < p.N;
    const unsigned dstream = attn_stream(p.stream_id, (unsigned)bh);

    for (int j = tid; j < p.Npad / 8; j += 256) kms[j] = (unsigned char)mask_bits(ld<unsigned long long>(p.kmask + (long)b * p.Npad + j * 8));
    wait_lgkm0();

    // The head of the backward pass rides in this kernel's prologue (round 6: it was a launch of its own, attn_bwd_prep_kernel, 48 times a
    // step): dO = dOg * gate (0 on masked query rows) straight from the token-major upstream gradient -- a lane holds exactly the 4 x 8
    // channels of its query that the dO^T fragments need --, delta = sum_d dO O and the gate's gradient.  dO and delta are also WRITTEN:
    // the dK / dV kernel that follows in the stream reads them for every query.
    bf16x8 qf[4], dof[4];
    float dl = 0.f;
    {
        const long trow = ((long)b * p.N + (qin ? q : 0)) * ((long)p.H * DH) + h * DH + hi * 8;
        const float gt = qin ? p.gate[bh * p.N + q] : 0.f;
        const bool qkeep = qin && p.kmask[(long)b * p.Npad + q] != 0;
        float dot = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = qin ? ld<bf16x8>(p.Q + (bh * p.N + q) * DH + ks * 16 + hi * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            float og[8], ov[8], d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { og[e] = 0.f; ov[e] = 0.f; }
            if (qin) {
                unpack8(ld<u32x4>(p.dOg + trow + ks * 16), og);
                unpack8(ld<u32x4>(p.O + trow + ks * 16), ov);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dot = fmaf(og[e], ov[e], dot);
                d[e] = qkeep ? og[e] * gt : 0.f;
            }
            const u32x4 pk = pack8(d);
            dof[ks] = __builtin_bit_cast(bf16x8, pk);
            if (qin) st<u32x4>(p.dO + (bh * p.N + q) * DH + ks * 16 + hi * 8, pk);
        }
        dot += lane32_other(dot);                      // the other half of the row's channels (lane l ^ 32)
        if (!qkeep) dot = 0.f;
        dl = dot * gt;
        if (qin && hi == 0) {
            p.delta[bh * p.N + q] = dl;
            p.dgate_pre[bh * p.N + q] = dl * (1.f - gt);
        }
    }
    const float kx = p.scale / CLAMP;
    const Clamp32 cc = clamp32(kx, 1.f);
    const float cl2 = CLAMP * LOG2E;
    // dS = P (dP - delta) (1 - th^2) scale: the trailing `scale` is folded into the exponent of P (lse - log2(scale))
    const float lse = qin ? p.lse2[bh * p.N + q] - log2f(p.scale) : 1e30f;
    const unsigned hrow = rand_base(p.seed_dev ? *p.seed_dev : p.seed, dstream) + (unsigned)q * 0x85ebca77u;
    const unsigned hlane = hrow + (unsigned)(2 * hi) * 0xc2b2ae3du;

    const bf16_t* Kbase = p.K + bh * p.N * DH;
    const bf16_t* Vbase = p.V + bh * p.N * DH;
    // (LDS row r + 8 of the second instruction: swap23 adds 4 to the tile row, the chunk swizzle flips bit 2 -- derived, not kept in registers)
    const int srow0 = swap23(wave * 16 + (lane >> 3));
    const unsigned scol0 = (unsigned)(((lane & 7) ^ swz(wave * 16 + (lane >> 3))) * 16);
    auto issue = [&](int t, int stage) __attribute__((always_inline)) {
        const int k0 = t * 64;
        unsigned char* S = smem + stage * FSTAGE;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // (a wave-uniform base + ONE 32-bit per-lane offset for both tensors: four 64-bit per-lane pointers kept across the loop were what
            //  the 168-register budget spilled)
            const unsigned off = (unsigned)min(k0 + srow0 + 4 * u, p.N - 1) * (DH * 2) + (scol0 ^ (64u * u));
            glds16((const char*)Kbase + off, S + (wave * 2 + u) * 1024);
            glds16((const char*)Vbase + off, S + 8192 + (wave * 2 + u) * 1024);
        }
    };
    int foff[4];
    frag_offsets(l31, hi, foff);
    const TrOff tro = tr_offsets(lane);

    f32x16 dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

    asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]), "v"(dof[0]), "v"(dof[1]), "v"(dof[2]), "v"(dof[3]), "v"(lse), "v"(dl), "v"(hrow));
    issue(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        const int k0 = kt * 64;
        wait_vmcnt<0>();
        barrier_raw();
        if (kt + 1 < nt) issue(kt + 1, (kt + 1) & 1);
        if (!live) continue;
        const unsigned char* Kt = smem + (kt & 1) * FSTAGE;
        const unsigned char* Vr = Kt + 8192;
        const unsigned long long km8 = ld<unsigned long long>(kms + (k0 >> 3)) >> (8 * hi);
        const bool allk = wave_all((km8 & 0x00ff00ff00ff00ffull) == 0x00ff00ff00ff00ffull);
        const unsigned long long* dropw = (DROP && SHARE) ? p.dropbits + (((long)bh * nt + kt) * nqb + qb) * 32 : nullptr;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = ld<bf16x8>(Kt + foff[ks] + kb * 4096);
                const bf16x8 vf = ld<bf16x8>(Vr + foff[ks] + kb * 4096);
                s = mfma32(kf, qf[ks], s);
                dp = mfma32(vf, dof[ks], dp);             // dP^T = V dO^T
                if (WPS == 3 && ks == 1) sched_fence();  // (at most four fragments in flight: eight cost 16 registers the 168-register budget does not have)
            }
            const int tier = clamp_tier(abs_max_16(s, 0.f), kx);
            const unsigned kmb = (unsigned)(km8 >> (32 * kb));
            const unsigned hk = hlane + ((unsigned)(k0 >> 2) + 8 * kb) * 0xc2b2ae3du;
            const unsigned long long* dw = (DROP && SHARE) ? dropw + 16 * kb : nullptr;
            const unsigned char* Kb = Kt + kb * 4096;
            f32x2_ th[8];
            clamp_block(s, tier, cc, th);
            dq_block<DROP, SHARE>(th, dp, cl2, lse, dl, p.inv_keep, hk, p.thresh, dw, allk, kmb, Kb, tro, dq);
        }
    }
    if (!live) return;
    const long orow = (bh * p.N + (qin ? q : 0)) * DH;
    store_rows32(dq, hi, qin, [&](int db, int g, int hi_, bool ok, const float* a, const float* c) __attribute__((always_inline)) {
        u32x2 x = pack4(a), y = pack4(c);
        swap_pair(x, y);
        if (ok) st<u32x4>(p.dQ + orow + 32 * db + 8 * (g + hi_), u32x4{x[0], x[1], y[0], y[1]});
    });
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV

// 16 scores (th = tanh of them) + 16 dP of one 32-query block of a lane (its key against queries 16 s2 + 8 hi + e of the block) -> P
// (dropped, packed) and dS (packed) -> dV^T += dO^T P, dK^T += Q^T dS.  lse8 / del8: LDS, the block's lse / delta values from this
// lane's first query on; nvalid: queries of this lane (counted from its first one) that lie inside the sequence (tail tile only);
// hq: dropout counter of (first query, key >> 2); dbits: the forward's keep bits of the block for this key (bit 16 s2 + e)
template <bool DROP, bool SHARE>
__device__ __forceinline__ void dkv_block(const f32x2_ (&th)[8], const f32x16& dp, float cl2, float scale, float inv_keep,
                                          const float* lse8, const float* del8, bool tail, int nvalid, unsigned hq, int key3, unsigned thresh,
                                          unsigned dbits, const unsigned char* Qb, const unsigned char* dOb, const TrOff& tro, f32x16 (&dk)[2],
                                          f32x16 (&dv)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        float pd[8], dsv[8];
#pragma unroll
        for (int eh = 0; eh < 2; ++eh) {
            f32x4 ls4 = ld<f32x4>(lse8 + 16 * s2 + 4 * eh);
            const f32x4 dl4 = ld<f32x4>(del8 + 16 * s2 + 4 * eh);
            if (tail) {              // rows past the end of the sequence: p = exp2(-1e30) = 0
                cold_path();
#pragma unroll
                for (int j = 0; j < 4; ++j) ls4[j] = 16 * s2 + 4 * eh + j < nvalid ? ls4[j] : 1e30f;
            }
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int e = 4 * eh + j, r = 8 * s2 + e;
                const f32x2_ t = th[r >> 1];
                const f32x2_ arg = t * cl2 - f32x2_{ls4[j], ls4[j + 1]};
                const f32x2_ pr = {fast_exp2(arg[0]), fast_exp2(arg[1])};
                f32x2_ pk = pr, t1 = {dp[r], dp[r + 1]};
                const f32x2_ dl2 = {dl4[j], dl4[j + 1]};
                if (DROP) {
                    if (SHARE) {             // the keep bit as an all-ones / all-zeros word (v_bfe_i32), ANDed onto the two values it switches
                        const unsigned m0 = 0u - ((dbits >> (16 * s2 + e)) & 1u), m1 = 0u - ((dbits >> (16 * s2 + e + 1)) & 1u);
                        pk[0] = __uint_as_float(__float_as_uint(pr[0]) & m0);
                        pk[1] = __uint_as_float(__float_as_uint(pr[1]) & m1);
                        t1[0] = __uint_as_float(__float_as_uint(t1[0]) & m0);
                        t1[1] = __uint_as_float(__float_as_uint(t1[1]) & m1);
                    } else {
                        unsigned w0, w1;
                        drop4(hq + (unsigned)(16 * s2 + e) * 0x85ebca77u, 0u, w0, w1);
                        const bool k0 = drop_sample(w0, w1, key3) >= thresh;
                        drop4(hq + (unsigned)(16 * s2 + e + 1) * 0x85ebca77u, 0u, w0, w1);
                        const bool k1 = drop_sample(w0, w1, key3) >= thresh;
                        pk[0] = k0 ? pr[0] : 0.f;
                        pk[1] = k1 ? pr[1] : 0.f;
                        t1[0] = k0 ? t1[0] : 0.f;
                        t1[1] = k1 ? t1[1] : 0.f;
                    }
                    t1 = t1 * inv_keep - dl2;         // (1 / (1 - p) is applied to dV once at the end)
                } else {
                    t1 = t1 - dl2;
                }
                const f32x2_ t2 = (t * -scale) * t + scale;      // (1 - th^2) * scale
                const f32x2_ ds = (pr * t2) * t1;
                pd[e] = pk[0];
                pd[e + 1] = pk[1];
                dsv[e] = ds[0];
                dsv[e + 1] = ds[1];
            }
        }
        // (no key masking here: a lane's scores all belong to ITS key, whose dK / dV row is zeroed at the end)
        const bf16x8 pf = pack8f(pd), df = pack8f(dsv);
#pragma unroll
        for (int db = 0; db < 2; ++db) dv[db] = mfma32(tr_frag(dOb, tro, db, s2 * 2048), pf, dv[db]);
#pragma unroll
        for (int db = 0; db < 2; ++db) dk[db] = mfma32(tr_frag(Qb, tro, db, s2 * 2048), df, dk[db]);
    }
}

// One workgroup = 128 keys (a wave owns 32: lane = one key), sweep over 64-query tiles.  S = Q K^T with the query rows of the Q / dO
// tiles stored as K-type images, so that the accumulators (rows = queries) are directly the B operands of dV^T = dO^T P and
// dK^T = Q^T dS, whose A operands come out of the same tiles through transposing reads.  The 64 lse / delta values of a query tile
// ride along in the ring stage.
constexpr int DSTAGE32 = FSTAGE + 512;

template <bool DROP, bool SHARE>
__global__ __launch_bounds__(256, 2) void attn_dkv32_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * DSTAGE32];
    lds_declare(smem, sizeof(smem));
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform_i(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nt = (p.N + 63) / 64, nk = (p.N + 127) / 128, nqb = (p.N + 31) / 32;
    const RingWG wg = ring_wg(p, nk);
    const int h = wg.h, b = wg.b;
    const long bh = (long)b * p.H + h;
    const int kb32 = wg.x * 4 + wave;                             // 32-key block of this wave
    const int key = kb32 * 32 + l31;
    const bool kin = key < p.N;
    const bool live = kb32 * 32 < p.N;
    const int kkeep = kin && p.kmask[(long)b * p.Npad + (kin ? key : 0)] != 0;
    const unsigned dstream = attn_stream(p.stream_id, (unsigned)bh);

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = kin ? ld<bf16x8>(p.K + (bh * p.N + key) * DH + ks * 16 + hi * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        vf[ks] = kin ? ld<bf16x8>(p.V + (bh * p.N + key) * DH + ks * 16 + hi * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const float kx = p.scale / CLAMP;
    const Clamp32 cc = clamp32(kx, 1.f);
    const float cl2 = CLAMP * LOG2E;
    const unsigned hkey = rand_base(p.seed_dev ? *p.seed_dev : p.seed, dstream) + ((unsigned)key >> 2) * 0xc2b2ae3du;

    const bf16_t* Qbase = p.Q + bh * p.N * DH;
    const bf16_t* dObase = p.dO + bh * p.N * DH;
    const float* lsebase = p.lse2 + bh * p.N;
    const float* delbase = p.delta + bh * p.N;
    // (LDS row r + 8 of the second instruction: swap23 adds 4 to the tile row, the chunk swizzle flips bit 2 -- derived, not kept in registers)
    const int srow0 = swap23(wave * 16 + (lane >> 3));
    const unsigned scol0 = (unsigned)(((lane & 7) ^ swz(wave * 16 + (lane >> 3))) * 16);
    auto issue = [&](int t, int stage) __attribute__((always_inline)) {
        const int q0 = t * 64;
        unsigned char* S = smem + stage * DSTAGE32;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned off = (unsigned)min(q0 + srow0 + 4 * u, p.N - 1) * (DH * 2) + (scol0 ^ (64u * u));
            glds16((const char*)Qbase + off, S + (wave * 2 + u) * 1024);
            glds16((const char*)dObase + off, S + 8192 + (wave * 2 + u) * 1024);
        }
        if (wave < 2) glds4((wave == 0 ? lsebase : delbase) + min(q0 + lane, p.N - 1), S + FSTAGE + wave * 256);
    };
    int foff[4];
    frag_offsets(l31, hi, foff);
    const TrOff tro = tr_offsets(lane);

    // Shared dropout masks.  The forward kept (query, key) in the ballot word [key tile][32-query block][16 kbf + rf], bit
    // (query & 31) + 32 hif, where (kbf, rf = 8 sf + ef, hif) is the position of the key in the forward's lane layout: key & 63 =
    // 32 kbf + 16 sf + 8 hif + ef.  This lane's key is fixed, so per 32-query block it needs ONE 32-bit half word; its scores of the
    // block (queries 16 s2 + 8 hi + e) are the bits 16 s2 + 8 hi + e of it.
    const unsigned char* dbase = nullptr;
    unsigned dnext[2] = {0u, 0u};
    if (DROP && SHARE) {
        const int k6 = key & 63;
        const int wf = 16 * (k6 >> 5) + 8 * ((k6 >> 4) & 1) + (k6 & 7), hif = (k6 >> 3) & 1;
        dbase = (const unsigned char*)(p.dropbits + ((long)bh * nt + (key >> 6 < nt ? key >> 6 : 0)) * nqb * 32 + wf) + 4 * hif;
        dnext[0] = ld<unsigned>(dbase);
        dnext[1] = 1 < nqb ? ld<unsigned>(dbase + 32 * 8) : 0u;
    }
    asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]), "v"(hkey), "v"(kkeep),
                 "v"(dnext[0]), "v"(dnext[1]));
    issue(0, 0);
    for (int qt = 0; qt < nt; ++qt) {
        const int q0 = qt * 64;
        wait_vmcnt<0>();
        barrier_raw();
        const unsigned dword[2] = {dnext[0] >> (8 * hi), dnext[1] >> (8 * hi)};
        if (qt + 1 < nt) {
            if (DROP && SHARE) {
                dnext[0] = ld<unsigned>(dbase + (long)(2 * qt + 2) * 32 * 8);
                dnext[1] = 2 * qt + 3 < nqb ? ld<unsigned>(dbase + (long)(2 * qt + 3) * 32 * 8) : 0u;
            }
            issue(qt + 1, (qt + 1) & 1);
        }
        if (!live) continue;
        const unsigned char* Qt = smem + (qt & 1) * DSTAGE32;
        const unsigned char* dOt = Qt + 8192;
        const float* lse_s = (const float*)(Qt + FSTAGE);
        const float* del_s = lse_s + 64;
        const bool tail = q0 + 64 > p.N;
#pragma unroll
        for (int qb2 = 0; qb2 < 2; ++qb2) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 qfr = ld<bf16x8>(Qt + foff[ks] + qb2 * 4096);
                const bf16x8 dofr = ld<bf16x8>(dOt + foff[ks] + qb2 * 4096);
                s = mfma32(qfr, kf[ks], s);
                dp = mfma32(dofr, vf[ks], dp);
            }
            const int tier = clamp_tier(abs_max_16(s, 0.f), kx);
            const unsigned char* Qb = Qt + qb2 * 4096;
            const unsigned char* dOb = dOt + qb2 * 4096;
            const int qrow = q0 + 32 * qb2 + 8 * hi;              // + 16 s2 + e: this lane's queries
            const unsigned hq = hkey + (unsigned)qrow * 0x85ebca77u;
            f32x2_ th[8];
            clamp_block(s, tier, cc, th);
            dkv_block<DROP, SHARE>(th, dp, cl2, p.scale, p.inv_keep, lse_s + 32 * qb2 + 8 * hi, del_s + 32 * qb2 + 8 * hi, tail, p.N - qrow, hq, key & 3,
                                   p.thresh, dword[qb2], Qb, dOb, tro, dk, dv);
        }
    }
    if (!live) return;
    const long orow = (bh * p.N + (kin ? key : 0)) * DH;
    const float vs = DROP ? p.inv_keep : 1.f;
    store_rows32(dk, hi, kin, [&](int db, int g, int hi_, bool ok, const float* a, const float* c) __attribute__((always_inline)) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = kkeep ? a[e] : 0.f; v[4 + e] = kkeep ? c[e] : 0.f; }
        u32x2 x = pack4(v), y = pack4(v + 4);
        swap_pair(x, y);
        if (ok) st<u32x4>(p.dK + orow + 32 * db + 8 * (g + hi_), u32x4{x[0], x[1], y[0], y[1]});
    });
    store_rows32(dv, hi, kin, [&](int db, int g, int hi_, bool ok, const float* a, const float* c) __attribute__((always_inline)) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = kkeep ? a[e] * vs : 0.f; v[4 + e] = kkeep ? c[e] * vs : 0.f; }
        u32x2 x = pack4(v), y = pack4(v + 4);
        swap_pair(x, y);
        if (ok) st<u32x4>(p.dV + orow + 32 * db + 8 * (g + hi_), u32x4{x[0], x[1], y[0], y[1]});
    });
}

}  // namespace

namespace e2k_attn32 {

void fwd(const void* attn_args, bool drop, bool share, hipStream_t st) {
    const AttnArgs& a = *(const AttnArgs*)attn_args;
    const dim3 grid(((a.N + 127) / 128) * a.H * a.B), block(256);
    const char* pr = getenv("E2K_ATTN32_PROBE");
    const int probe = pr ? atoi(pr) : 0;
    if (probe && drop && share) {
        switch (probe) {
#define E2K_P(X) case X: hipLaunchKernelGGL((attn_fwd32_kernel<true, true, X>), grid, block, 0, st, a); return;
            E2K_P(1) E2K_P(2) E2K_P(3) E2K_P(4) E2K_P(8) E2K_P(12) E2K_P(16) E2K_P(28) E2K_P(32) E2K_P(64) E2K_P(96) E2K_P(31) E2K_P(127)
#undef E2K_P
        }
    }
    if (drop && share) hipLaunchKernelGGL((attn_fwd32_kernel<true, true>), grid, block, 0, st, a);
    else if (drop) hipLaunchKernelGGL((attn_fwd32_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_fwd32_kernel<false, false>), grid, block, 0, st, a);
}

void bwd_dq(const void* attn_args, bool drop, bool share, hipStream_t st) {
    const AttnArgs& a = *(const AttnArgs*)attn_args;
    const dim3 grid(((a.N + 127) / 128) * a.H * a.B), block(256);
    if (drop && share) hipLaunchKernelGGL((attn_dq32_kernel<true, true, 3>), grid, block, 0, st, a);
    else if (drop) hipLaunchKernelGGL((attn_dq32_kernel<true, false, 3>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_dq32_kernel<false, false, 3>), grid, block, 0, st, a);
}

void bwd_dkv(const void* attn_args, bool drop, bool share, hipStream_t st) {
    const AttnArgs& a = *(const AttnArgs*)attn_args;
    const dim3 grid(((a.N + 127) / 128) * a.H * a.B), block(256);
    if (drop && share) hipLaunchKernelGGL((attn_dkv32_kernel<true, true>), grid, block, 0, st, a);
    else if (drop) hipLaunchKernelGGL((attn_dkv32_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_dkv32_kernel<false, false>), grid, block, 0, st, a);
}

}  // namespace e2k_attn32
