// gfx950-only instruction wrappers (no host equivalent).  The host logic-checker shadows this
// header with tests/emu/e2k_asm.h, which models the same semantics in plain C++.
#pragma once

namespace e2k {

typedef short s16x4_ __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16: every lane passes the LDS address of 4 consecutive bf16 (8 B).  Within each
// 16-lane group the 16x4 block {lane q -> its 4 elements} is transposed: lane q receives, as element j,
// the (q & 3)-th element of what lane 4*j + (q >> 2) addressed.  With lane q addressing row (q >> 2),
// columns 4*(q & 3).. of a row-major [4][16] block, lane q ends up with column q, rows 0..3.
__device__ __forceinline__ s16x4_ lds_read_tr16_b64(const void* lds_ptr) {
    typedef s16x4_ __attribute__((address_space(3))) * lp_t;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(lds_ptr));
}

// The same instruction through inline assembly, for kernels that prefetch with global_load_lds.  The builtin above carries
// no information about WHICH LDS bytes it reads, so the compiler orders it after every LDS-DMA in flight: it puts an
// s_waitcnt vmcnt(0) in front of each group of transposing reads, which drains the prefetch queue once per phase (seen in
// the ISA of the round-2 gemm_tn_256_kernel: five vmcnt(0) per loop iteration next to the hand-counted waits).  This form
// is opaque to the compiler: the caller orders it against the DMA exactly like the ordinary ds_reads (counted wait_vmcnt +
// barrier), and calls lds_tr_wait(regs...) before the first use of the results (s_waitcnt lgkmcnt(0); the empty
// assembly statements tie the result registers to the wait so that no consumer is scheduled above it).
// imm_off: byte offset that is a compile-time constant after unrolling (DS offset field, < 65536).
__device__ __forceinline__ void lds_tr_issue(s16x4_& dst, const void* lds_ptr, int imm_off) {
    typedef const __attribute__((address_space(3))) void* lp_t;
    const unsigned a = (unsigned)(unsigned long long)(lp_t)(lds_ptr);
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "n"(imm_off) : "memory");
}
template <class T> __device__ __forceinline__ void lds_tie(T& r) { asm volatile("" : "+v"(r)); }
template <class... T> __device__ __forceinline__ void lds_tr_wait(T&... r) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    (lds_tie(r), ...);
}

// raw v_exp_f32 (2^x, no denormal fix-up: results below 2^-126 flush to 0) and v_rcp_f32 (1 ulp)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// tell the compiler a value is wave-uniform (moves it to an SGPR; scalar loads / scalar operands downstream)
__device__ __forceinline__ float uniform_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

// wave64 sum on the VALU (DPP row operations, no LDS crossbar): quad swaps, half-row / row mirrors, then
// row_bcast15 / row_bcast31 carry the row sums up to lane 63, which is broadcast through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_(float v) {
    int x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, x);
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = dpp_add_<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add_<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add_<0x140, 0xf>(v);     // row_mirror   -> every lane of a 16-lane row holds the row sum
    v = dpp_add_<0x142, 0xa>(v);     // row_bcast15  -> rows 1, 3 += rows 0, 2
    v = dpp_add_<0x143, 0xc>(v);     // row_bcast31  -> rows 2, 3 += row 1 (= rows 0+1)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// load through the constant address space: with a wave-uniform address this becomes s_load (scalar cache, lgkmcnt),
// which neither occupies VGPRs nor forces a vmcnt(0) drain of vector loads that are still in flight.  Only for data
// that no wave of the running kernel writes.
__device__ __forceinline__ float sload(const float* p) {
    typedef const __attribute__((address_space(4))) float* cp_t;
    return *(cp_t)(p);
}

// same DPP reduction without the broadcast: the wave total is valid in lane 63 only
__device__ __forceinline__ float wave_sum_last(float v) {
    v = dpp_add_<0xB1, 0xf>(v);
    v = dpp_add_<0x4E, 0xf>(v);
    v = dpp_add_<0x141, 0xf>(v);
    v = dpp_add_<0x140, 0xf>(v);
    v = dpp_add_<0x142, 0xa>(v);
    v = dpp_add_<0x143, 0xc>(v);
    return v;
}
// four wave sums at once, totals valid in lane 63.  Written as v_add_f32 with the DPP modifier on the operand (the
// builtin form above costs three instructions per step once the compiler packs the adds into v_pk_add_f32: old-value
// init + v_mov_dpp + half a packed add).  The four chains are interleaved, which also covers the two wait states a DPP
// read needs after a VALU write of its source; the leading s_nop covers the producer of `a`.
#define E2K_DPP4_(ctrl)                                        \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                       \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n"                       \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n"                       \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n"
__device__ __forceinline__ void wave_sum_last4(float& a, float& b, float& c, float& d) {
    asm("s_nop 1\n"
        E2K_DPP4_("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        E2K_DPP4_("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        E2K_DPP4_("row_half_mirror row_mask:0xf bank_mask:0xf")
        E2K_DPP4_("row_mirror row_mask:0xf bank_mask:0xf")
        E2K_DPP4_("row_bcast:15 row_mask:0xa bank_mask:0xf")
        E2K_DPP4_("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef E2K_DPP4_
// sum over each aligned group of 8 lanes (every lane of the group gets the group sum)
__device__ __forceinline__ float group8_sum(float v) {
    v = dpp_add_<0xB1, 0xf>(v);
    v = dpp_add_<0x4E, 0xf>(v);
    v = dpp_add_<0x141, 0xf>(v);
    return v;
}
// value of lane L (wave-uniform index) as a wave-uniform scalar (v_readlane_b32 -> SGPR)
__device__ __forceinline__ float lane_bcast(float v, int L) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), L));
}
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }       // v_sqrt_f32, 1 ulp

// wave vote: true when the predicate holds in every lane (all 64 lanes active at the call sites)
__device__ __forceinline__ bool wave_all(bool pred) { return __builtin_amdgcn_ballot_w64(pred) == ~0ull; }

// lane predicate <-> 64-bit wave mask.  ballot is the SGPR pair a v_cmp writes anyway; inverse_ballot hands a mask to
// v_cndmask as its select operand (no per-lane unpacking)
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
__device__ __forceinline__ bool wave_inverse_ballot(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// v_writelane_b32: `old` with the register of lane `lane_sel` (a compile-time constant here) replaced by the wave-uniform `val`
// (inline assembly: clang declares the writelane builtin for the device pass only, and this header is also seen by the host pass.
//  The data operand is an SGPR a v_cmp may just have written: the ISA lists a wait-state requirement only for an SGPR used as
//  the LANE SELECT of v_readlane / v_writelane, which is an immediate here)
__device__ __forceinline__ unsigned wave_writelane(unsigned old, unsigned val, int lane_sel) {
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(lane_sel));
    return old;
}
// 64-bit load through the constant address space (s_load_dwordx2 when the address is wave-uniform), see sload()
__device__ __forceinline__ unsigned long long sload64(const unsigned long long* p) {
    typedef const __attribute__((address_space(4))) unsigned long long* cp_t;
    return *(cp_t)(p);
}

// v_mfma_f32_32x32x16_bf16: A 32x16 (lane l: row l & 31, k-slots 8 (l >> 5) .. + 7), B 16x32 (lane l: column l & 31, the same
// k-slots), C / D 32x32: lane l holds column l & 31, register r <-> row (r & 3) + 8 (r >> 2) + 4 (l >> 5)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_ __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma32(bf16x8_ a, bf16x8_ b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// v_permlane32_swap_b32: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (the other two halves stay)
__device__ __forceinline__ void lane32_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// v_permlane16_swap_b32: the odd rows (16 lanes each) of `a` trade places with the even rows of `b`
__device__ __forceinline__ void lane16_swap(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
// value of the same register in lane (l ^ 32)
__device__ __forceinline__ float lane32_other(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    lane32_swap(a, b);              // lanes 0..31: b = upper half's value; lanes 32..63: a = lower half's value
    return __builtin_bit_cast(float, (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) < 32u) ? b : a);
}

// Wave totals of N per-lane values (N a multiple of 4), left TRANSPOSED over the rows of the wave: on return v[i], i < N / 4, holds in
// every lane of row r (lanes 16 r .. 16 r + 15) the total of value i + (N / 4) (r & 1) + (N / 2) (r >> 1).  The two swap levels halve the
// number of live partial sums each (one swap + one add per PAIR of values), only the last N / 4 registers go through the four in-row DPP
// levels: 2.5 N instructions where the plain butterfly (wave_sum_last4) takes 6 N.
template <int K> __device__ __forceinline__ void row_sum_dpp(float* v) {          // K <= 4 registers: sum over the 16 lanes of each row
    static_assert(K >= 1 && K <= 4, "");
#define E2K_DPPR_(ctrl)                                                                     \
    asm("s_nop 1\n"                                                                         \
        "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                                               \
        : "+v"(v[0]));                                                                      \
    if constexpr (K > 1) asm("v_add_f32_dpp %0, %0, %0 " ctrl "\n" : "+v"(v[1]));           \
    if constexpr (K > 2) asm("v_add_f32_dpp %0, %0, %0 " ctrl "\n" : "+v"(v[2]));           \
    if constexpr (K > 3) asm("v_add_f32_dpp %0, %0, %0 " ctrl "\n" : "+v"(v[3]));
    E2K_DPPR_("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
    E2K_DPPR_("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
    E2K_DPPR_("row_half_mirror row_mask:0xf bank_mask:0xf")
    E2K_DPPR_("row_mirror row_mask:0xf bank_mask:0xf")
#undef E2K_DPPR_
}
template <int N> __device__ __forceinline__ void wave_sum_rows(float (&v)[N]) {
    static_assert(N % 4 == 0 && N >= 4, "");
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        unsigned a = __builtin_bit_cast(unsigned, v[i]), b = __builtin_bit_cast(unsigned, v[i + N / 2]);
        lane32_swap(a, b);           // lanes 0..31: (a, b) = value i of lanes l, l + 32; lanes 32..63: value i + N / 2
        v[i] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        unsigned a = __builtin_bit_cast(unsigned, v[i]), b = __builtin_bit_cast(unsigned, v[i + N / 4]);
        lane16_swap(a, b);           // even rows: value i (+ N / 2) summed over its row pair; odd rows: value i + N / 4 (+ N / 2)
        v[i] = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    constexpr int R = N / 4;
#pragma unroll
    for (int i = 0; i + 4 <= R; i += 4) row_sum_dpp<4>(v + i);
    if constexpr (R % 4 != 0) row_sum_dpp<R % 4>(v + (R & ~3));
}
// value index held by register i of the lanes of row r after wave_sum_rows<N>
template <int N> __device__ __forceinline__ constexpr int wave_sum_rows_index(int i, int r) { return i + (N / 4) * (r & 1) + (N / 2) * (r >> 1); }

// 16 wave masks (SGPR pairs, e.g. fresh from v_cmp) -> 128 consecutive bytes at a wave-uniform address, through the scalar data cache
// (s_store_dwordx2: no vector instruction, no VGPR).  The statement waits for its own stores (lgkmcnt) before the SGPRs may be reused;
// the leading s_nop covers a VALU write of an SGPR read by the scalar memory instruction.  The cache is written back by
// sstore_flush(), which every wave that stored must execute before it ends.
template <bool WAIT = true>
__device__ __forceinline__ void sstore_masks16(unsigned long long* dst, const unsigned long long (&m)[16]) {
    typedef __attribute__((address_space(1))) unsigned long long* gp_t;
#define E2K_SST16_ \
        "s_nop 4\n" \
        "s_store_dwordx2 %1, %0, 0x0\n s_store_dwordx2 %2, %0, 0x8\n s_store_dwordx2 %3, %0, 0x10\n s_store_dwordx2 %4, %0, 0x18\n" \
        "s_store_dwordx2 %5, %0, 0x20\n s_store_dwordx2 %6, %0, 0x28\n s_store_dwordx2 %7, %0, 0x30\n s_store_dwordx2 %8, %0, 0x38\n" \
        "s_store_dwordx2 %9, %0, 0x40\n s_store_dwordx2 %10, %0, 0x48\n s_store_dwordx2 %11, %0, 0x50\n s_store_dwordx2 %12, %0, 0x58\n" \
        "s_store_dwordx2 %13, %0, 0x60\n s_store_dwordx2 %14, %0, 0x68\n s_store_dwordx2 %15, %0, 0x70\n s_store_dwordx2 %16, %0, 0x78\n"
#define E2K_SST16_OPS_ \
        :: "s"((gp_t)dst), "s"(m[0]), "s"(m[1]), "s"(m[2]), "s"(m[3]), "s"(m[4]), "s"(m[5]), "s"(m[6]), "s"(m[7]), \
           "s"(m[8]), "s"(m[9]), "s"(m[10]), "s"(m[11]), "s"(m[12]), "s"(m[13]), "s"(m[14]), "s"(m[15]) : "memory"
    if (WAIT) asm volatile(E2K_SST16_ "s_waitcnt lgkmcnt(0)" E2K_SST16_OPS_);
    else asm volatile(E2K_SST16_ "s_nop 0" E2K_SST16_OPS_);          // (probe: are the data SGPRs read at issue?)
#undef E2K_SST16_
#undef E2K_SST16_OPS_
}
// eight masks (64 bytes)
__device__ __forceinline__ void sstore_masks8(unsigned long long* dst, unsigned long long m0, unsigned long long m1, unsigned long long m2, unsigned long long m3,
                                              unsigned long long m4, unsigned long long m5, unsigned long long m6, unsigned long long m7) {
    typedef __attribute__((address_space(1))) unsigned long long* gp_t;
    asm volatile(
        "s_nop 4\n"
        "s_store_dwordx2 %1, %0, 0x0\n s_store_dwordx2 %2, %0, 0x8\n s_store_dwordx2 %3, %0, 0x10\n s_store_dwordx2 %4, %0, 0x18\n"
        "s_store_dwordx2 %5, %0, 0x20\n s_store_dwordx2 %6, %0, 0x28\n s_store_dwordx2 %7, %0, 0x30\n s_store_dwordx2 %8, %0, 0x38\n"
        "s_waitcnt lgkmcnt(0)"
        :: "s"((gp_t)dst), "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7) : "memory");
}
__device__ __forceinline__ void sstore_flush() { asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory"); }

// counted wait on outstanding vector-memory ops (LDS-DMA included) and a raw workgroup barrier that does NOT drain
// them: lets global_load_lds prefetches stay in flight across barriers (cdna_hip_programming.md T3+T4)
// this wave's LDS operations (and scalar loads) have completed: what a raw barrier does NOT wait for
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier_keep_vm() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// bare workgroup barrier: no counter is drained (ds_reads and global_load_lds issued before it stay in flight across it;
// the consumer's own s_waitcnt decides what must have landed).  The sched_barriers pin the instruction groups on either
// side: without them the scheduler is free to move MFMAs across, which is legal but undoes a hand-made phase schedule.
__device__ __forceinline__ void barrier_raw() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// issue priority of this wave (0..3): raised around an MFMA group so that it wins over the other wave of the SIMD,
// which is in its load phase (cdna_hip_programming.md T5: only useful when the waves of a SIMD are in different phases)
// lanes of a wave exchanging data through LDS: the hardware runs them in lockstep (the compiler's lgkmcnt wait orders the
// read after the write); this only keeps the compiler from moving the accesses across.  The host model rendezvouses here.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }
template <int P> __device__ __forceinline__ void set_prio() { __builtin_amdgcn_s_setprio(P); }
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Pins a value into its registers at this point of the program: arithmetic that produced it cannot sink below, arithmetic that uses
// it cannot rise above (sched_barrier alone orders memory operations only: instruction selection places pure arithmetic where it
// likes).  With order_memory() a fully unrolled loop keeps the load / compute interleaving it was written with.  No instructions.
template <class T> __device__ __forceinline__ void pin_vgpr(T& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void order_memory() { asm volatile("" ::: "memory"); }

// Names the static LDS array that this kernel's global_load_lds copies land in.  No code on the GPU; the host model of the
// kernels (tests/emu) records the range and aborts on an LDS-DMA whose destination leaves it (an out-of-range destination
// on the GPU lands in whatever a co-resident workgroup keeps at that LDS address).
__device__ __forceinline__ void lds_declare(const void*, unsigned) {}

// global_load_lds_dwordx4: asynchronous 16-byte-per-lane copy HBM -> LDS that bypasses the VGPRs.  The LDS destination
// is wave-uniform: lane l lands at lds_base + 16*l (the global source address is per lane).  Completion is tracked by
// vmcnt; a following __syncthreads() drains it (cdna_hip_programming.md section 5).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    typedef __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)(gsrc), (lp_t)(lds_base), 16, 0, 0);
}
// global_load_lds_dword: 4 bytes per lane, lane l lands at lds_base + 4*l
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_base) {
    typedef __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)(gsrc), (lp_t)(lds_base), 4, 0, 0);
}

}  // namespace e2k
