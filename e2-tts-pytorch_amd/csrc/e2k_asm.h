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

// global_load_lds_dwordx4: asynchronous 16-byte-per-lane copy HBM -> LDS that bypasses the VGPRs.  The LDS destination
// is wave-uniform: lane l lands at lds_base + 16*l (the global source address is per lane).  Completion is tracked by
// vmcnt; a following __syncthreads() drains it (cdna_hip_programming.md section 5).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    typedef __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)(gsrc), (lp_t)(lds_base), 16, 0, 0);
}

}  // namespace e2k
