/* tools/guard/guardalloc.c -- DIAGNOSIS TOOL (LD_PRELOAD), not part of the product.
 *
 * An Electric-Fence style allocator for ONE size window of the process's heap, made to name the writer of the stray
 * 4-byte host store of rounds 3-5 ("What's weak" 1): a numpy array of 920 bytes, allocated right after
 * three HIP contexts were closed, changed under the test's feet.
 *
 *   - malloc / calloc / realloc / memalign of GUARD_MIN <= size <= GUARD_MAX bytes (default 600 .. 1400) get pages of their
 *     own out of one reserved address range, the block's END on a page boundary and an unmapped page behind it;
 *   - free() of such a block makes its pages PROT_NONE and NEVER hands them out again (the range is reserved up front, 64 GiB of
 *     address space, and only grows): a CPU store through a stale pointer -- from the library, the HIP runtime's threads,
 *     ctypes, anything -- dies at the store, and the SIGSEGV handler below prints the faulting address, what the block was, and
 *     the native backtrace of the storing thread;
 *   - GUARD_RO_AFTER_WRITE is not attempted here: the test's own instrument (LWHIP_TEST_GUARD=1, tests/test_hip_parity.py) keeps
 *     the oracle's input arrays in read-only pages, which tells a CPU writer (fault) from a DMA writer (content changes with no
 *     fault).
 *
 * Everything else goes to the C library's allocator (dlsym RTLD_NEXT).  Build: gcc -O2 -fPIC -shared -o guardalloc.so
 * guardalloc.c -ldl;  use: LD_PRELOAD=tools/guard/guardalloc.so python -m pytest ...
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <execinfo.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#define PAGE 4096ul
#define ARENA_BYTES (64ul << 30)
#define MAGIC 0x6775617264616c6cull /* "guardall" */

typedef struct
{
    uint64_t magic;
    size_t mapLen;  /* pages of this block, the header's included (the trailing fence page is not) */
    size_t userLen;
    void* user;
    uint64_t serial;
    void* bt[6];    /* who allocated it (return addresses) */
} Hdr;

static void* (*real_malloc)(size_t);
static void* (*real_calloc)(size_t, size_t);
static void* (*real_realloc)(void*, size_t);
static void (*real_free)(void*);
static void* (*real_memalign)(size_t, size_t);

static char* arena;
static _Atomic size_t arenaUsed;
static _Atomic uint64_t serial, nAlloc, nFree;
static size_t gMin = 600, gMax = 1400;
static int ready, inBt;
static __thread int busy;

/* (dlsym itself calloc()s while we resolve the real functions) */
static char boot[65536];
static size_t bootUsed;
static void* boot_alloc(size_t n)
{
    n = (n + 15) & ~15ul;
    if (bootUsed + n > sizeof boot)
        _exit(97);
    void* p = boot + bootUsed;
    bootUsed += n;
    return p;
}
static int in_boot(const void* p) { return (const char*)p >= boot && (const char*)p < boot + sizeof boot; }
static int in_arena(const void* p) { return arena && (const char*)p >= arena && (const char*)p < arena + ARENA_BYTES; }

static void on_segv(int sig, siginfo_t* si, void* uc)
{
    (void)uc;
    char buf[512];
    const char* a = (const char*)si->si_addr;
    int n = snprintf(buf, sizeof buf, "\nGUARDALLOC: signal %d at address %p (%s the guarded range)\n", sig, (void*)a,
                     in_arena(a) ? "INSIDE" : "outside");
    (void)!write(2, buf, n);
    if (in_arena(a))
    {
        /* the block: walk back to a page that holds a header (headers of freed blocks are PROT_NONE too: make readable) */
        char* pg = (char*)((uintptr_t)a & ~(PAGE - 1));
        for (int back = 0; back < 8 && pg >= arena; ++back, pg -= PAGE)
        {
            mprotect(pg, PAGE, PROT_READ);
            const Hdr* h = (const Hdr*)pg;
            if (h->magic == MAGIC)
            {
                n = snprintf(buf, sizeof buf,
                             "GUARDALLOC: block #%llu, %zu bytes at %p (fault at +%ld), %s; allocated from:\n",
                             (unsigned long long)h->serial, h->userLen, h->user, (long)(a - (const char*)h->user),
                             "freed or fence page");
                (void)!write(2, buf, n);
                backtrace_symbols_fd(h->bt, 6, 2);
                break;
            }
        }
    }
    (void)!write(2, "GUARDALLOC: backtrace of the faulting thread:\n", 46);
    void* bt[48];
    int d = backtrace(bt, 48);
    backtrace_symbols_fd(bt, d, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

static void init(void)
{
    static _Atomic int started;
    if (ready || atomic_exchange(&started, 1))
        return;
    real_malloc = dlsym(RTLD_NEXT, "malloc");
    real_calloc = dlsym(RTLD_NEXT, "calloc");
    real_realloc = dlsym(RTLD_NEXT, "realloc");
    real_free = dlsym(RTLD_NEXT, "free");
    real_memalign = dlsym(RTLD_NEXT, "memalign");
    const char* v = getenv("GUARD_MIN");
    if (v)
        gMin = strtoul(v, 0, 0);
    v = getenv("GUARD_MAX");
    if (v)
        gMax = strtoul(v, 0, 0);
    arena = mmap(0, ARENA_BYTES, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (arena == MAP_FAILED)
        arena = 0;
    /* (backtrace() loads libgcc on first use and that mallocs: do it once here, outside any handler) */
    void* bt[4];
    inBt = 1;
    backtrace(bt, 4);
    inBt = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    ready = 1;
    if (getenv("GUARD_VERBOSE"))
        fprintf(stderr, "GUARDALLOC: on, sizes %zu..%zu, range %p + 64 GiB\n", gMin, gMax, (void*)arena);
}

static void* guarded(size_t size, size_t align)
{
    if (!arena || busy)
        return 0;
    if (align < 16)
        align = 16;
    const size_t body = (size + align - 1) & ~(align - 1);
    /* [header page .. data pages][fence page]: the block ends at the fence (rounded to the alignment) */
    const size_t pages = 1 + (body + PAGE - 1) / PAGE;
    const size_t need = (pages + 1) * PAGE;
    const size_t off = atomic_fetch_add(&arenaUsed, need);
    if (off + need > ARENA_BYTES)
        return 0;
    char* base = arena + off;
    if (mprotect(base, pages * PAGE, PROT_READ | PROT_WRITE) != 0)
        return 0; /* (out of mappings: vm.max_map_count) */
    Hdr* h = (Hdr*)base;
    h->magic = MAGIC;
    h->mapLen = pages * PAGE;
    h->userLen = size;
    h->user = base + pages * PAGE - body;
    h->serial = atomic_fetch_add(&serial, 1);
    busy = 1;
    memset(h->bt, 0, sizeof h->bt);
    if (!inBt)
    {
        void* bt[8];
        int d = backtrace(bt, 8);
        for (int i = 2; i < d && i < 8; ++i)
            h->bt[i - 2] = bt[i];
    }
    busy = 0;
    atomic_fetch_add(&nAlloc, 1);
    return h->user;
}

static Hdr* hdr_of(void* p)
{
    char* pg = (char*)(((uintptr_t)p - 1) & ~(PAGE - 1));
    for (int back = 0; back < 1024 && pg >= arena; ++back, pg -= PAGE)
    {
        Hdr* h = (Hdr*)pg;
        if (h->magic == MAGIC && h->user == p)
            return h;
        if (h->magic == MAGIC)
            break;
    }
    return 0;
}

size_t malloc_usable_size(void* p);
static int want(size_t n) { return ready && n >= gMin && n <= gMax; }

void* malloc(size_t n)
{
    if (!real_malloc)
    {
        if (!ready && !arena)
            init();
        if (!real_malloc)
            return boot_alloc(n);
    }
    if (want(n))
    {
        void* p = guarded(n, 16);
        if (p)
            return p;
    }
    return real_malloc(n);
}

void* calloc(size_t a, size_t b)
{
    if (!real_calloc)
    {
        void* p = boot_alloc(a * b); /* (static storage: zero) */
        return p;
    }
    const size_t n = a * b;
    if (want(n))
    {
        void* p = guarded(n, 16); /* (fresh anonymous pages: zero) */
        if (p)
            return p;
    }
    return real_calloc(a, b);
}

void free(void* p)
{
    if (!p || in_boot(p))
        return;
    if (in_arena(p))
    {
        Hdr* h = hdr_of(p);
        if (!h)
        {
            fprintf(stderr, "GUARDALLOC: free(%p): not the start of a live guarded block (double free?)\n", p);
            abort();
        }
        const size_t len = h->mapLen;
        /* quarantine for good: the pages go back to the kernel, the addresses never come back */
        madvise((char*)h + PAGE, len - PAGE, MADV_DONTNEED);
        mprotect(h, len, PROT_NONE);
        atomic_fetch_add(&nFree, 1);
        return;
    }
    real_free(p);
}

void* realloc(void* p, size_t n)
{
    if (!real_realloc)
        init();
    if (p && in_boot(p))
    {
        void* q = malloc(n);
        memcpy(q, p, n); /* (boot blocks are small; over-read stays inside the static array) */
        return q;
    }
    if (p && in_arena(p))
    {
        Hdr* h = hdr_of(p);
        void* q = malloc(n);
        if (q && h)
            memcpy(q, p, h->userLen < n ? h->userLen : n);
        free(p);
        return q;
    }
    if (want(n))
    {
        /* growing / shrinking into the window: move it under guard (the old size is not known: malloc_usable_size) */
        void* q = guarded(n, 16);
        if (q)
        {
            if (p)
            {
                const size_t old = malloc_usable_size(p);
                memcpy(q, p, old < n ? old : n);
                real_free(p);
            }
            return q;
        }
    }
    return real_realloc(p, n);
}

void* memalign(size_t al, size_t n)
{
    if (!real_memalign)
        init();
    if (want(n) && al <= PAGE)
    {
        void* p = guarded(n, al);
        if (p)
            return p;
    }
    return real_memalign(al, n);
}
void* aligned_alloc(size_t al, size_t n) { return memalign(al, n); }
size_t malloc_usable_size(void* p)
{
    static size_t (*real)(void*);
    if (p && in_arena(p))
    {
        Hdr* h = hdr_of(p);
        return h ? h->userLen : 0;
    }
    if (!real)
        real = dlsym(RTLD_NEXT, "malloc_usable_size");
    return p && !in_boot(p) ? real(p) : 0;
}
int posix_memalign(void** out, size_t al, size_t n)
{
    void* p = memalign(al, n);
    if (!p)
        return ENOMEM;
    *out = p;
    return 0;
}

__attribute__((constructor)) static void guard_ctor(void) { init(); }
__attribute__((destructor)) static void guard_dtor(void)
{
    if (getenv("GUARD_VERBOSE"))
        fprintf(stderr, "GUARDALLOC: %llu guarded allocations, %llu freed (quarantined), %zu MiB of address range used\n",
                (unsigned long long)nAlloc, (unsigned long long)nFree, (size_t)arenaUsed >> 20);
}
