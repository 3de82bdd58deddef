// TEST INFRASTRUCTURE: host model of csrc/e2k_asm.h (see tests/emu/hip/hip_runtime.h).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

namespace e2k {

typedef short s16x4_ __attribute__((ext_vector_type(4)));

inline s16x4_ lds_read_tr16_b64(const void* lds_ptr) {
    s16x4_ mine;
    memcpy(&mine, lds_ptr, 8);
    emu::Wave& w = emu::cur_wave();
    emu::Fiber& f = emu::cur_fiber();
    int p = f.par;
    f.par ^= 1;
    int lane = emu::lane_id();
    memcpy(w.slot[p][lane], &mine, 8);
    emu::wave_rendezvous(w);
    int base = lane & ~15, q = lane & 15;
    s16x4_ r;
    for (int j = 0; j < 4; ++j) {
        s16x4_ src;
        memcpy(&src, w.slot[p][base + 4 * j + (q >> 2)], 8);
        r[j] = src[q & 3];
    }
    return r;
}

// asynchronous form (device: inline assembly + lds_tr_wait): the model reads at issue
inline void lds_tr_issue(s16x4_& dst, const void* lds_ptr, int imm_off) { dst = lds_read_tr16_b64((const char*)lds_ptr + imm_off); }
template <class... T> inline void lds_tr_wait(T&...) {}

inline float uniform_f(float v) { return v; }
inline int uniform_i(int v) { return v; }
inline float wave_sum_fast(float v) {          // same result up to summation order
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
inline float wave_sum_last(float v) { return wave_sum_fast(v); }     // (valid in every lane here; lane 63 is what is used)
inline void wave_sum_last4(float& a, float& b, float& c, float& d) {
    a = wave_sum_fast(a); b = wave_sum_fast(b); c = wave_sum_fast(c); d = wave_sum_fast(d);
}
inline float group8_sum(float v) {
    for (int m = 1; m <= 4; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
inline float lane_bcast(float v, int L) { return __shfl(v, L); }
inline float sload(const float* p) { return *p; }
inline float fast_rsq(float x) { return 1.0f / sqrtf(x); }
inline float fast_sqrt(float x) { return sqrtf(x); }
inline bool wave_all(bool pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v &= __shfl_xor(v, m);
    return v != 0;
}
inline unsigned long long wave_ballot(bool pred) {
    unsigned long long v = pred ? (1ull << emu::lane_id()) : 0ull;
    for (int m = 32; m >= 1; m >>= 1) v |= __shfl_xor(v, m);
    return v;
}
inline bool wave_inverse_ballot(unsigned long long m) { return (m >> emu::lane_id()) & 1ull; }
inline unsigned wave_writelane(unsigned old, unsigned val, int lane_sel) { return emu::lane_id() == lane_sel ? val : old; }      // (val is wave-uniform)
inline unsigned long long sload64(const unsigned long long* p) { return *p; }
inline float fast_exp2(float x) { return exp2f(x); }
inline float fast_rcp(float x) { return 1.0f / x; }

// host model of v_mfma_f32_32x32x16_bf16 (fragment layout: csrc/e2k_asm.h)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_ __attribute__((ext_vector_type(8)));
inline f32x16 mfma32(bf16x8_ a, bf16x8_ b, f32x16 c) {
    struct AB { bf16x8_ a, b; };
    emu::Wave& w = emu::cur_wave();
    emu::Fiber& f = emu::cur_fiber();
    int p = f.par;
    f.par ^= 1;
    int lane = emu::lane_id();
    AB ab{a, b};
    memcpy(w.slot[p][lane], &ab, sizeof(AB));
    emu::wave_rendezvous(w);
    const int j = lane & 31, hi = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int h = 0; h < 2; ++h) {
            AB ra, rb;
            memcpy(&ra, w.slot[p][i + 32 * h], sizeof(AB));
            memcpy(&rb, w.slot[p][j + 32 * h], sizeof(AB));
            for (int e = 0; e < 8; ++e) acc = fmaf(emu_bf2f(ra.a[e]), emu_bf2f(rb.b[e]), acc);
        }
        d[r] = acc;
    }
    return d;
}
inline void lane32_swap(unsigned& a, unsigned& b) {
    struct P { unsigned a, b; };
    const int lane = emu::lane_id();
    const P other = emu::wave_exchange(P{a, b}, lane ^ 32);
    if (lane < 32) b = other.a; else a = other.b;
}
inline float lane32_other(float v) { return __shfl_xor(v, 32); }
inline void lane16_swap(unsigned& a, unsigned& b) {
    struct P { unsigned a, b; };
    const int lane = emu::lane_id();
    const P other = emu::wave_exchange(P{a, b}, lane ^ 16);
    if ((lane >> 4) & 1) a = other.b; else b = other.a;       // odd rows of a <-> even rows of b
}
template <int N> inline void wave_sum_rows(float (&v)[N]) {      // same swap / add order as the device version (fp32 sums in the same order)
    for (int i = 0; i < N / 2; ++i) {
        unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + N / 2]);
        lane32_swap(a, b);
        v[i] = __uint_as_float(a) + __uint_as_float(b);
    }
    for (int i = 0; i < N / 4; ++i) {
        unsigned a = __float_as_uint(v[i]), b = __float_as_uint(v[i + N / 4]);
        lane16_swap(a, b);
        v[i] = __uint_as_float(a) + __uint_as_float(b);
    }
    for (int i = 0; i < N / 4; ++i) {
        v[i] += __shfl_xor(v[i], 1);
        v[i] += __shfl_xor(v[i], 2);
        v[i] += __shfl(v[i], (emu::lane_id() & ~7) | (7 - (emu::lane_id() & 7)));       // row_half_mirror
        v[i] += __shfl(v[i], (emu::lane_id() & ~15) | (15 - (emu::lane_id() & 15)));    // row_mirror
    }
}
template <int N> inline constexpr int wave_sum_rows_index(int i, int r) { return i + (N / 4) * (r & 1) + (N / 2) * (r >> 1); }

template <bool WAIT = true>
inline void sstore_masks16(unsigned long long* dst, const unsigned long long (&m)[16]) {      // (every lane writes the same 128 bytes)
    for (int i = 0; i < 16; ++i) dst[i] = m[i];
}
inline void sstore_masks8(unsigned long long* dst, unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3,
                          unsigned long long m4, unsigned long long m5, unsigned long long m6, unsigned long long m7) {
    dst[0] = m0; dst[1] = m1; dst[2] = m2; dst[3] = m3; dst[4] = m4; dst[5] = m5; dst[6] = m6; dst[7] = m7;
}
inline void sstore_flush() {}
inline void wait_lgkm0() {}
template <int N> inline void wait_vmcnt() { emu::land_pending(N); }
inline void barrier_keep_vm() { emu::block_rendezvous(); }
inline void barrier_raw() { emu::block_rendezvous(); }
inline void wave_sync() { (void)__shfl_xor(0, 1); }       // fibers of a wave run one after the other: a true rendezvous
template <int P> inline void set_prio() {}
inline void sched_fence() {}
template <class T> inline void pin_vgpr(T&) {}
inline void order_memory() {}

// the static LDS array the calling kernel's LDS-DMA copies must land in (set by lds_declare at kernel entry; a block's
// fibers all run on one host thread, and so do its `static thread_local` __shared__ arrays)
inline thread_local const char* g_lds_lo = nullptr;
inline thread_local const char* g_lds_hi = nullptr;
inline void lds_declare(const void* base, unsigned bytes) { g_lds_lo = (const char*)base; g_lds_hi = g_lds_lo + bytes; }
inline void lds_check(const void* dst, int n) {
    if (g_lds_lo == nullptr || (const char*)dst < g_lds_lo || (const char*)dst + n > g_lds_hi) {
        fprintf(stderr, "emu: global_load_lds destination %p (+%d) outside the kernel's declared LDS array [%p, %p)\n", dst, n, (const void*)g_lds_lo, (const void*)g_lds_hi);
        abort();
    }
}
// host model of global_load_lds_dwordx4: lane l copies its 16 bytes to (wave-uniform) lds_base + 16*l
inline void glds16(const void* gsrc, void* lds_base) {
    void* dst = (char*)lds_base + 16 * emu::lane_id();
    lds_check(dst, 16);
    if (emu::g_glds_late) {
        emu::PendingCopy c;
        memcpy(c.data, gsrc, 16);
        c.dst = dst;
        c.size = 16;
        emu::cur_fiber().pending.push_back(c);
    } else {
        memcpy(dst, gsrc, 16);
    }
}
// global_load_lds_dword: lane l copies 4 bytes to lds_base + 4*l
inline void glds4(const void* gsrc, void* lds_base) {
    void* dst = (char*)lds_base + 4 * emu::lane_id();
    lds_check(dst, 4);
    if (emu::g_glds_late) {
        emu::PendingCopy c;
        memcpy(c.data, gsrc, 4);
        c.dst = dst;
        c.size = 4;
        emu::cur_fiber().pending.push_back(c);
    } else {
        memcpy(dst, gsrc, 4);
    }
}

}  // namespace e2k
