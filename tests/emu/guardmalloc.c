/* TEST INFRASTRUCTURE (never part of the product): an LD_PRELOAD shim that turns every tensor-sized host allocation into
 * its own mapping with an inaccessible page right behind (default) or right in front (E2K_GUARD_UNDER=1) of it, so that a
 * kernel of the host logic-checker build (tests/emu) that reads or writes even one element past a buffer dies with SIGSEGV
 * at the faulting instruction instead of silently touching a neighbour.  On the GPU such an access only faults when the
 * neighbouring virtual page happens to be unmapped (the end of a caching-allocator segment), i.e. rarely and depending on
 * the box; this makes it deterministic.
 *
 *   gcc -O2 -shared -fPIC -o libguardmalloc.so guardmalloc.c -ldl -lpthread
 *   LD_PRELOAD=tests/emu/libguardmalloc.so python -m pytest tests -m "not gpu" -k emu
 *
 * Only posix_memalign / aligned_alloc / memalign requests (what c10::alloc_cpu issues for tensor storage) of at least
 * E2K_GUARD_MIN bytes (default 64) are guarded; everything else goes to the C library.  Guarded virtual addresses are never
 * handed out again (a freed region stays PROT_NONE: use after free faults as well). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#define PAGE 4096UL
static unsigned long ARENA_BYTES = 1UL << 40;        /* 1 TB of reserved address space (nothing committed) */
static unsigned char* arena;
static unsigned long arena_next;                     /* bump pointer, in bytes */
static uint32_t* pages_of;                           /* [page index of a region's first page] -> data pages of the region */
static uint64_t* size_of;                            /* requested size, same index */
static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static int under = 0;
static unsigned long min_bytes = 64;
static int ready = 0;

static void (*real_free)(void*);
static void* (*real_realloc)(void*, size_t);

__attribute__((constructor)) static void init_once(void) {
    if (ready) return;
    real_free = dlsym(RTLD_NEXT, "free");
    real_realloc = dlsym(RTLD_NEXT, "realloc");
    const char* e = getenv("E2K_GUARD_UNDER");
    under = e && e[0] == '1';
    e = getenv("E2K_GUARD_MIN");
    if (e) min_bytes = strtoul(e, 0, 10);
    arena = mmap(0, ARENA_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    unsigned long np = ARENA_BYTES / PAGE;
    pages_of = mmap(0, np * sizeof(uint32_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    size_of = mmap(0, np * sizeof(uint64_t), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (arena == MAP_FAILED || pages_of == MAP_FAILED || size_of == MAP_FAILED) {
        fprintf(stderr, "[guardmalloc] cannot reserve the arena\n");
        abort();
    }
    ready = 1;
}

static int ours(const void* p) {
    return ready && (const unsigned char*)p >= arena && (const unsigned char*)p < arena + ARENA_BYTES;
}

static void* guarded_alloc(size_t align, size_t size) {
    if (align < 64) align = 64;
    unsigned long data_pages = (size + align + PAGE - 1) / PAGE;
    pthread_mutex_lock(&mu);
    init_once();
    unsigned long off = arena_next;
    arena_next += (data_pages + 2) * PAGE;           /* [guard][data ...][guard] */
    pthread_mutex_unlock(&mu);
    if (arena_next > ARENA_BYTES) {
        fprintf(stderr, "[guardmalloc] arena exhausted\n");
        abort();
    }
    unsigned char* first = arena + off + PAGE;
    if (mprotect(first, data_pages * PAGE, PROT_READ | PROT_WRITE)) return 0;
    unsigned long idx = (off + PAGE) / PAGE;
    pages_of[idx] = (uint32_t)data_pages;
    size_of[idx] = size;
    unsigned char* p;
    if (under) {
        p = first;                                   /* page aligned: any read below p faults */
    } else {
        unsigned long end = data_pages * PAGE;
        p = first + ((end - size) & ~(align - 1));   /* as far right as the alignment allows: at most align - 1 slack bytes */
    }
    return p;
}

static void guarded_free(void* p) {
    unsigned long off = (unsigned char*)p - arena;
    unsigned long idx = off / PAGE;
    while (idx > 0 && pages_of[idx] == 0) --idx;     /* the region's first data page carries the length */
    unsigned long n = pages_of[idx];
    if (n == 0 || off / PAGE >= idx + n) {
        fprintf(stderr, "[guardmalloc] free of %p: not the start of a guarded region\n", p);
        abort();
    }
    pages_of[idx] = 0;
    /* drop the pages and leave the range inaccessible for good */
    mmap(arena + idx * PAGE, n * PAGE, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
}

int posix_memalign(void** out, size_t align, size_t size) {
    static int (*real)(void**, size_t, size_t);
    if (!real) real = dlsym(RTLD_NEXT, "posix_memalign");
    if (size < min_bytes || size < 64) return real(out, align, size);
    void* p = guarded_alloc(align, size);
    if (!p) return ENOMEM;
    *out = p;
    return 0;
}

void* aligned_alloc(size_t align, size_t size) {
    void* p = 0;
    return posix_memalign(&p, align, size) ? 0 : p;
}

void* memalign(size_t align, size_t size) {
    void* p = 0;
    return posix_memalign(&p, align, size) ? 0 : p;
}

void free(void* p) {
    if (!p) return;
    if (ours(p)) {
        guarded_free(p);
        return;
    }
    if (!real_free) real_free = dlsym(RTLD_NEXT, "free");
    real_free(p);
}

void* realloc(void* p, size_t size) {
    if (p && ours(p)) {
        unsigned long idx = ((unsigned char*)p - arena) / PAGE;
        while (idx > 0 && pages_of[idx] == 0) --idx;
        size_t old = size_of[idx];
        void* q = 0;
        if (posix_memalign(&q, 64, size ? size : 64)) return 0;
        memcpy(q, p, old < size ? old : size);
        guarded_free(p);
        return q;
    }
    if (!real_realloc) real_realloc = dlsym(RTLD_NEXT, "realloc");
    return real_realloc(p, size);
}
