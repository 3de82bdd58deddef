// Hardware-interaction probe (MI355X): does an LDS exchange of one kernel break when another kernel on another stream runs
// ds_read_b64_tr_b16 / global_load_lds on the same CUs?  Built by tools/probes/lds_victim/build.sh, driven by run.py.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// victim 1: lane 63 of every wave writes 24 floats (EXEC-masked ds_write_b128), one barrier, lanes < 32 add two waves' values
extern "C" __global__ __launch_bounds__(256) void victim_exchange(int iters, unsigned* errors) {
    __shared__ __attribute__((aligned(16))) float red[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (lane == 63) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                f32x4 v = {(float)(it * 100 + wave * 32 + g * 4), (float)(it * 100 + wave * 32 + g * 4 + 1),
                           (float)(it * 100 + wave * 32 + g * 4 + 2), (float)(it * 100 + wave * 32 + g * 4 + 3)};
                *reinterpret_cast<f32x4*>(&red[it & 1][wave][g * 4]) = v;
            }
        }
        __syncthreads();
        if (lane < 24) {
            const int w0 = wave & ~1;
            const float s = red[it & 1][w0][lane] + red[it & 1][w0 + 1][lane];
            const float want = (float)(it * 100 + w0 * 32 + lane) + (float)(it * 100 + (w0 + 1) * 32 + lane);
            if (s != want) ++bad;
        }
    }
    if (bad) atomicAdd(errors, bad);
}

// victim 2: a constant table in LDS, every lane reads 16 bytes at a lane-dependent offset over and over
extern "C" __global__ __launch_bounds__(256) void victim_table(int iters, unsigned* errors) {
    __shared__ __attribute__((aligned(16))) float tab[6 * 512];
    const int tid = threadIdx.x;
    for (int i = tid; i < 6 * 512; i += 256) tab[i] = (float)i;
    __syncthreads();
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int off = ((tid * 4 + it * 64) % (6 * 512 - 4)) & ~3;
        f32x4 v = *reinterpret_cast<const volatile f32x4*>(&tab[off]);
        if (v[0] != (float)off || v[1] != (float)(off + 1) || v[2] != (float)(off + 2) || v[3] != (float)(off + 3)) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}

// victim 3: DPP wave reduction (row operations + row_bcast) of known values
extern "C" __global__ __launch_bounds__(256) void victim_dpp(int iters, unsigned* errors) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        float a = (float)(lane + it), b = (float)(2 * lane), c = 1.f, d = (float)(lane & 3);
        asm volatile("s_nop 1\n"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (lane == 63 && (a != (float)(2016 + 64 * it) || b != 4032.f || c != 64.f || d != 96.f)) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
}

// co-runner: nothing but transposing LDS reads
extern "C" __global__ __launch_bounds__(256) void spam_tr(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[32768];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) ((unsigned*)buf)[i] = i;
    __syncthreads();
    typedef s16x4 __attribute__((address_space(3))) * lp_t;
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(buf + ((tid * 8 + it * 512) & 32760)));
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 0x7fffffff) sink[0] = 1.f;
}
// co-runner: nothing but plain 8-byte LDS reads (control)
extern "C" __global__ __launch_bounds__(256) void spam_plain(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[32768];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) ((unsigned*)buf)[i] = i;
    __syncthreads();
    int acc = 0;
    for (int it = 0; it < iters; ++it) {
        s16x4 v = *reinterpret_cast<const volatile s16x4*>(buf + ((tid * 8 + it * 512) & 32760));
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 0x7fffffff) sink[0] = 1.f;
}
// co-runner: LDS-DMA (global_load_lds_dwordx4) into its own LDS
extern "C" __global__ __launch_bounds__(256) void spam_glds(int iters, const unsigned char* src, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef __attribute__((address_space(1))) void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_global_load_lds((gp_t)(src + ((blockIdx.x * 256 + tid) * 16 + it * 4096) % (1 << 20)), (lp_t)(buf + wave * 8192 + (it & 7) * 1024), 16, 0, 0);
    }
    __syncthreads();
    if (buf[tid] == 0xff && buf[tid + 256] == 0xfe) sink[0] = 1.f;
}

// co-runner: holds 64 KB of LDS per workgroup (two per CU) and burns time without touching it much
extern "C" __global__ __launch_bounds__(256, 2) void spam_occupy(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char buf[65536];
    const int tid = threadIdx.x;
    buf[tid] = (unsigned char)tid;
    __syncthreads();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        acc = acc * 1.0001f + (float)buf[(tid + it) & 255];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

extern "C" int probe_launch(int which, int grid, int iters, void* p0, void* p1, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (which) {
        case 0: hipLaunchKernelGGL(victim_exchange, dim3(grid), dim3(256), 0, st, iters, (unsigned*)p0); break;
        case 1: hipLaunchKernelGGL(victim_table, dim3(grid), dim3(256), 0, st, iters, (unsigned*)p0); break;
        case 2: hipLaunchKernelGGL(victim_dpp, dim3(grid), dim3(256), 0, st, iters, (unsigned*)p0); break;
        case 10: hipLaunchKernelGGL(spam_tr, dim3(grid), dim3(256), 0, st, iters, (float*)p0); break;
        case 11: hipLaunchKernelGGL(spam_plain, dim3(grid), dim3(256), 0, st, iters, (float*)p0); break;
        case 12: hipLaunchKernelGGL(spam_glds, dim3(grid), dim3(256), 0, st, iters, (const unsigned char*)p1, (float*)p0); break;
        case 13: hipLaunchKernelGGL(spam_occupy, dim3(grid), dim3(256), 0, st, iters, (float*)p0); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
