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
