// TEST INFRASTRUCTURE ONLY -- host-side logic checker for the HIP kernels in
// e2-tts-pytorch_amd/csrc.  It is NOT a CPU fallback: nothing in the shipped
// package includes, links or loads it.  tests/emu/build_emu.py compiles the very
// same kernel sources (no #ifdefs in them) against this header instead of the
// real <hip/hip_runtime.h>, producing tests/emu/libe2k_emu.so, so that index
// math, reductions, MFMA fragment plumbing and barrier placement can be checked
// on a machine with no GPU before GPU minutes are spent.
//
// Execution model: one OS thread runs one workgroup at a time; the workgroup's
// threads are fibers (a minimal x86-64 stack switch) scheduled round-robin.  __syncthreads() is a true
// rendezvous over the block, wave-level exchanges (__shfl*, MFMA) a true
// rendezvous over the 64-lane wave.  __shared__ maps to `static thread_local`
// (one block per OS thread => block-shared).  "Device" pointers are host pointers.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
struct emu_graph;
static inline bool emu_capture_memset(void* p, int v, size_t n);
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (!emu_capture_memset(p, v, n)) memset(p, v, n); return 0; }
// events: the host model runs every launch synchronously; elapsed times are host wall-clock (plan profiling logic only)
#include <chrono>
typedef std::chrono::steady_clock::time_point* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new std::chrono::steady_clock::time_point(); return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { *e = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(*b - *a).count();
    return 0;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
#define hipStreamNonBlocking 1u
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int dummy; *s = (hipStream_t)&dummy; return 0; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
// launch lanes (csrc/plan.h): everything runs synchronously here, so a wait has nothing to wait for -- but waiting for an
// event that was never recorded (a no-op on the GPU, i.e. a missing ordering) is reported as an error
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) {
    return *e == std::chrono::steady_clock::time_point() ? 900 : 0;
}

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Wave {
    alignas(16) unsigned char slot[2][kWave][64];
    int arrived = 0;
    unsigned gen = 0;
    int nlanes = kWave;
};

// global_load_lds (LDS-DMA) is asynchronous on the GPU: the bytes may land any time between the issue and the wave's
// own `s_waitcnt vmcnt`.  The model can run both extremes: by default a copy lands AT ISSUE (earliest possible: exposes
// write-after-read mistakes, i.e. restaging a buffer that is still being read); with E2K_EMU_GLDS_LATE=1 it lands only
// when the issuing lane's counted wait (or a __syncthreads, which drains vmcnt on the GPU) forces it (latest possible:
// exposes read-before-landed mistakes in counted-vmcnt pipelines).
struct PendingCopy {
    unsigned char data[16];
    void* dst;
    int size = 16;
};
inline bool g_glds_late = false;

// Fiber switch: callee-saved registers + stack pointer (x86-64 SysV).  glibc's swapcontext also saves / restores the
// signal mask with a system call on every switch, which dominated the model's run time (every MFMA and every barrier is
// a rendezvous of 64 / 256 / 512 fibers).
struct Ctx { void* sp = nullptr; };
__attribute__((naked, noinline)) static void emu_switch(Ctx* /*from: rdi*/, Ctx* /*to: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq (%rsi), %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}

struct Fiber {
    std::vector<PendingCopy> pending;       // this lane's LDS-DMA copies that have not landed yet (late mode), oldest first
    Ctx ctx;
    dim3 tid;
    int lin = 0;
    int par = 0;
    bool done = false;
    char* stack = nullptr;
};

struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int cur = 0;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    Ctx sched;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    long progress = 0;      // bumped whenever a rendezvous completes or a fiber finishes (deadlock detection)
    const std::function<void()>* body = nullptr;
};

inline thread_local Block* g_blk = nullptr;

inline Fiber& cur_fiber() { return g_blk->fibers[g_blk->cur]; }
// let all but the newest `keep` LDS-DMA copies of the calling lane land (s_waitcnt vmcnt(keep))
inline void land_pending(int keep) {
    Fiber& f = cur_fiber();
    int n = (int)f.pending.size() - keep;
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) memcpy(f.pending[i].dst, f.pending[i].data, f.pending[i].size);
    f.pending.erase(f.pending.begin(), f.pending.begin() + n);
}
inline void yield() {
    Fiber& f = cur_fiber();
    emu_switch(&f.ctx, &g_blk->sched);
}
inline void block_rendezvous() {
    Block* b = g_blk;
    unsigned gen = b->bar_gen;
    if (++b->bar_arrived == b->nthreads) {
        b->bar_arrived = 0;
        b->bar_gen++;
        b->progress++;
    } else {
        while (b->bar_gen == gen) yield();
    }
}
inline Wave& cur_wave() { return g_blk->waves[cur_fiber().lin / kWave]; }
inline int lane_id() { return cur_fiber().lin % kWave; }
inline void wave_rendezvous(Wave& w) {
    unsigned gen = w.gen;
    if (++w.arrived == w.nlanes) {
        w.arrived = 0;
        w.gen++;
        g_blk->progress++;
    } else {
        while (w.gen == gen) yield();
    }
}

static void trampoline() {
    Block* b = g_blk;
    (*b->body)();
    land_pending(0);
    b->fibers[b->cur].done = true;
    emu_switch(&b->fibers[b->cur].ctx, &b->sched);
}

inline void run_block(Block& blk, const std::function<void()>& body) {
    g_blk = &blk;
    blk.body = &body;
    blk.bar_arrived = 0;
    int n = blk.nthreads;
    for (int i = 0; i < n; ++i) {
        Fiber& f = blk.fibers[i];
        f.done = false;
        f.pending.clear();
        f.par = 0;
        f.lin = i;
        f.tid = dim3(i % blk.bdim.x, (i / blk.bdim.x) % blk.bdim.y, i / (blk.bdim.x * blk.bdim.y));
        // first switch "returns" into trampoline(): [6 callee-saved slots][&trampoline][fake return address], with the
        // stack pointer at function entry = 8 mod 16 as after a call
        uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
        uint64_t* sp = (uint64_t*)top;
        *(--sp) = 0;
        *(--sp) = (uint64_t)(uintptr_t)&trampoline;
        for (int r = 0; r < 6; ++r) *(--sp) = 0;
        f.ctx.sp = sp;
    }
    int nw = (n + kWave - 1) / kWave;
    for (int w = 0; w < nw; ++w) {
        blk.waves[w].arrived = 0;
        blk.waves[w].nlanes = std::min(kWave, n - w * kWave);
    }
    int remaining = n;
    int idle_rounds = 0;
    while (remaining > 0) {
        const long before = blk.progress;
        for (int i = 0; i < n; ++i) {
            Fiber& f = blk.fibers[i];
            if (f.done) continue;
            blk.cur = i;
            emu_switch(&blk.sched, &f.ctx);
            if (f.done) { --remaining; blk.progress++; }
        }
        idle_rounds = (blk.progress == before) ? idle_rounds + 1 : 0;
        if (idle_rounds > 2) {
            fprintf(stderr, "emu: DEADLOCK in block (%u,%u,%u): %d threads stuck at a barrier / wave exchange that the "
                            "others never reach (divergent __syncthreads/__shfl or early return)\n",
                    blk.bid.x, blk.bid.y, blk.bid.z, remaining);
            abort();
        }
    }
    g_blk = nullptr;
}

// Fiber stacks are pooled for the life of the process: a fresh 256-KB malloc per fiber per launch is an mmap, a page fault or
// two and a munmap each -- ~16 k system calls per launch with 8 workers x 512 fibers, which was most of the suite's sys time.
inline std::mutex& stack_mutex() { static std::mutex m; return m; }
inline std::vector<char*>& stack_pool() { static std::vector<char*> v; return v; }
inline char* stack_get() {
    {
        std::lock_guard<std::mutex> g(stack_mutex());
        auto& v = stack_pool();
        if (!v.empty()) { char* s = v.back(); v.pop_back(); return s; }
    }
    return (char*)malloc(kStack);
}
inline void stack_put(char* s) {
    std::lock_guard<std::mutex> g(stack_mutex());
    stack_pool().push_back(s);
}

// Helper threads live for the life of the process (a launch used to create and join up to 8 threads): `run(n, fn)` has n
// helpers execute fn next to the caller and returns when all are done.  A block's `static thread_local` LDS arrays therefore
// keep what the previous launch on that thread left in them -- as real LDS does; a kernel must not count on zeroed LDS.
// (Processes are started with `spawn` in the tests, never forked after the pool exists.)
struct HelperPool {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    const std::function<void()>* fn = nullptr;
    long epoch = 0;
    int want = 0, pending = 0;
    void ensure(int n) {
        while ((int)threads.size() < n) {
            const int idx = (int)threads.size();
            threads.emplace_back([this, idx]() {
                long seen = 0;
                for (;;) {
                    const std::function<void()>* f = nullptr;
                    {
                        std::unique_lock<std::mutex> g(m);
                        cv_job.wait(g, [&] { return epoch != seen; });
                        seen = epoch;
                        if (idx < want) f = fn;
                    }
                    if (f) {
                        (*f)();
                        std::lock_guard<std::mutex> g(m);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
            threads.back().detach();
        }
    }
    void run(int n, const std::function<void()>& f) {
        {
            std::unique_lock<std::mutex> g(m);
            ensure(n);
            fn = &f; want = n; pending = n; ++epoch;
        }
        cv_job.notify_all();
        f();
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [&] { return pending == 0; });
        want = 0;
    }
};
inline HelperPool& helper_pool() { static HelperPool* p = new HelperPool; return *p; }      // (leaked on purpose: detached threads outlive statics)

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    long nblocks = (long)grid.x * grid.y * grid.z;
    int nthreads = block.x * block.y * block.z;
    if (nblocks == 0 || nthreads == 0) return;
    g_glds_late = getenv("E2K_EMU_GLDS_LATE") && atoi(getenv("E2K_EMU_GLDS_LATE")) != 0;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("E2K_EMU_THREADS")) hw = atoi(e);
    if (hw < 1) hw = 1;
    long nworkers = std::min<long>(hw, nblocks);
    std::atomic<long> next{0};
    const std::function<void()> worker = [&]() {
        Block blk;
        blk.bdim = block;
        blk.gdim = grid;
        blk.nthreads = nthreads;
        blk.fibers.resize(nthreads);
        blk.waves.resize((nthreads + kWave - 1) / kWave);
        for (auto& f : blk.fibers) f.stack = stack_get();
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= nblocks) break;
            blk.bid = dim3(b % grid.x, (b / grid.x) % grid.y, b / ((long)grid.x * grid.y));
            run_block(blk, body);
        }
        for (auto& f : blk.fibers) stack_put(f.stack);
    };
    if (nworkers == 1) {
        worker();
    } else {
        helper_pool().run((int)nworkers - 1, worker);      // the calling thread is the first worker
    }
}

template <class T>
inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "exchange payload too large");
    Wave& w = cur_wave();
    Fiber& f = cur_fiber();
    int p = f.par;
    f.par ^= 1;
    memcpy(w.slot[p][lane_id()], &v, sizeof(T));
    wave_rendezvous(w);
    T r;
    int s = src_lane;
    if (s < 0 || s >= w.nlanes) s = lane_id();
    memcpy(&r, w.slot[p][s], sizeof(T));
    return r;
}

}  // namespace emu

#define threadIdx (emu::cur_fiber().tid)
#define blockIdx (emu::g_blk->bid)
#define blockDim (emu::g_blk->bdim)
#define gridDim (emu::g_blk->gdim)
#define warpSize 64

// stream capture -> graph (csrc/plan.hip, e2k_query_plan_graph_capture): between hipStreamBeginCapture and hipStreamEndCapture NOTHING
// executes -- kernel launches and memsets are appended to the graph as closures -- and hipGraphLaunch runs them in capture order (the
// host model has no concurrency between lanes: capture order is a valid topological order of the graph).  One capture per thread.
struct emu_graph { std::vector<std::function<void()>> nodes; };
typedef emu_graph* hipGraph_t;
typedef emu_graph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline thread_local emu_graph* emu_capture = nullptr;
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    if (emu_capture) return 901;
    emu_capture = new emu_graph();
    return 0;
}
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) {
    if (!emu_capture) return 901;
    *g = emu_capture;
    emu_capture = nullptr;
    return 0;
}
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) { *e = new emu_graph(*g); return 0; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) {
    for (auto& f : e->nodes) f();
    return 0;
}
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return 0; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return 0; }

static inline bool emu_capture_memset(void* p, int v, size_t n) {
    if (!emu_capture) return false;
    emu_capture->nodes.push_back([=]() { memset(p, v, n); });
    return true;
}

template <class K, class... A>
inline void hipLaunchKernelGGL(K k, dim3 g, dim3 b, size_t, hipStream_t, A... args) {
    if (emu_capture) { emu_capture->nodes.push_back([=]() { emu::launch(g, b, [=]() { k(args...); }); }); return; }
    emu::launch(g, b, [=]() { k(args...); });
}

static inline void __syncthreads() {
    emu::land_pending(0);           // the compiler drains vmcnt before the barrier of a __syncthreads when LDS-DMA is in flight
    emu::block_rendezvous();
}
static inline void __builtin_amdgcn_s_barrier() { emu::block_rendezvous(); }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
#define __builtin_amdgcn_readfirstlane(x) (x)

template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu::wave_exchange(v, emu::lane_id() ^ mask); }
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu::wave_exchange(v, src); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) { return emu::wave_exchange(v, emu::lane_id() + d); }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
// separately rounded product / sum (never contracted into an fma)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline void sincosf_(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }

static inline float atomicAdd(float* p, float v) {
    unsigned* up = (unsigned*)p;
    unsigned old = __atomic_load_n(up, __ATOMIC_RELAXED);
    for (;;) {
        float nf = __uint_as_float(old) + v;
        unsigned nu = __float_as_uint(nf);
        if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return __uint_as_float(old);
    }
}
static inline double atomicAdd(double* p, double v) {       // (blocks run on several OS threads: a real CAS loop)
    unsigned long long* up = (unsigned long long*)p;
    unsigned long long old = __atomic_load_n(up, __ATOMIC_RELAXED);
    for (;;) {
        double od;
        memcpy(&od, &old, 8);
        double nd = od + v;
        unsigned long long nu;
        memcpy(&nu, &nd, 8);
        if (__atomic_compare_exchange_n(up, &old, nu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return od;
    }
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }      // (blocks run on several OS threads)
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- MFMA (gfx950) ----
// v_mfma_f32_16x16x32_bf16: A 16x32 (lane l: row l&15, k-slots 8*(l>>4)+j), B 32x16 (lane l: col l&15, same
// k-slots), C/D: col = l&15, row = 4*(l>>4)+reg  (cdna_hip_programming.md section 3).
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
static inline float emu_bf2f(short s) { return __uint_as_float(((unsigned)(unsigned short)s) << 16); }
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
    struct AB { emu_bf16x8 a, b; };
    emu::Wave& w = emu::cur_wave();
    emu::Fiber& f = emu::cur_fiber();
    int p = f.par;
    f.par ^= 1;
    int lane = emu::lane_id();
    AB ab{a, b};
    memcpy(w.slot[p][lane], &ab, sizeof(AB));
    emu::wave_rendezvous(w);
    int j = lane & 15, g = lane >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * g + r;
        float acc = d[r];
        for (int gg = 0; gg < 4; ++gg) {
            AB ra, rb;
            memcpy(&ra, w.slot[p][i + 16 * gg], sizeof(AB));
            memcpy(&rb, w.slot[p][j + 16 * gg], sizeof(AB));
            for (int e = 0; e < 8; ++e) acc = fmaf(emu_bf2f(ra.a[e]), emu_bf2f(rb.b[e]), acc);
        }
        d[r] = acc;
    }
    return d;
}
