// Launchers of the 32-rows-per-wave attention ring kernels (attn32.hip), called from the C ABI entry points in attn.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace e2k_attn32 {

// a: the AttnArgs of attn_common.h (passed as an opaque pointer: the struct lives in an anonymous namespace of each translation unit,
// with one definition in the shared header)
void fwd(const void* attn_args, bool drop, bool share, hipStream_t st);
void bwd_dq(const void* attn_args, bool drop, bool share, hipStream_t st);
void bwd_dkv(const void* attn_args, bool drop, bool share, hipStream_t st);

}  // namespace e2k_attn32
