// aeth_pool.hip -- pinned host buffers: the object pool of the reference (src/pool.rs:43-221) over hipHostMalloc'ed
// elements, the registry of host ranges the runtime may copy from / to asynchronously, and the explicit opt-in
// registration of caller memory.
//
// Why this exists (round 3).  Until round 2 the host pipeline (aeth_fir_stream_host) called hipHostRegister /
// hipHostUnregister on whatever slices the caller passed -- numpy buffers, read-only file mappings, ranges that are
// not page-aligned -- for the length of one call, and threw the return codes away.  A process that did so and later
// made plain pageable hipMemcpyAsync uploads from recycled host addresses died twice with SIGABRT inside the runtime.
// The library no longer registers memory behind the caller's back, ever:
//   * aeth_pool_*        elements are allocated pinned ONCE (hipHostMalloc) and handed out / taken back like the
//                        reference's Pool<T> (take, take_or_make, Elem::drop -> give_back, len, cap).  A caller that
//                        produces its samples into pool elements streams with no registration and no staging copy.
//   * aeth_host_register the explicit opt-in for memory the caller already owns: whole pages only, overlap with any
//                        range this library knows refused, every runtime return code checked and reported.
//   * everything else    is staged through the context's own pool elements by host threads (aeth_pipeline.hip).
#include "aeth_internal.h"
#include "aeth_host.h"

#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

struct aeth_pool {
    int device = 0;                     // of the context it was made on (kept by value: a pool may outlive its context)
    size_t elem_bytes = 0;
    int flags = 0;
    std::mutex mu;                      // Pool<T> = Arc<Mutex<PoolInner<T>>> (pool.rs:71-73)
    std::vector<void *> elems;          // checked-in elements (PoolInner.elems, :163)
    std::vector<void *> owned;          // every element made so far; cap = owned.size() (PoolInner.cap, :171)
};

namespace {

// host ranges known to be page-locked: pool elements and explicit registrations.  [lo, hi) keyed by lo.
struct Range { uintptr_t hi; int kind; };
std::mutex g_mu;
std::map<uintptr_t, Range> g_ranges;

bool overlaps_locked(uintptr_t lo, uintptr_t hi)
{
    auto it = g_ranges.upper_bound(lo);                 // first range starting after lo
    if (it != g_ranges.end() && it->first < hi) return true;
    if (it != g_ranges.begin()) { --it; if (it->second.hi > lo) return true; }
    return false;
}

size_t page_size()
{
    static const size_t p = [] { long v = sysconf(_SC_PAGESIZE); return v > 0 ? (size_t)v : (size_t)4096; }();
    return p;
}

int pool_make_locked(aeth_pool *p, void **out)          // the `maker` (pool.rs:46,117): one pinned element
{
    aeth::DeviceGuard g(p->device);
    void *h = nullptr;
    AETH_HIP(hipHostMalloc(&h, p->elem_bytes, hipHostMallocPortable));
    aeth::pinned_add(h, p->elem_bytes, aeth::PIN_POOL);
    p->owned.push_back(h);
    *out = h;
    return AETH_OK;
}

}  // namespace

namespace aeth {

void pinned_add(const void *p, size_t bytes, int kind)
{
    std::lock_guard<std::mutex> l(g_mu);
    g_ranges[(uintptr_t)p] = Range{(uintptr_t)p + bytes, kind};
}

void pinned_remove(const void *p)
{
    std::lock_guard<std::mutex> l(g_mu);
    g_ranges.erase((uintptr_t)p);
}

bool host_range_pinned(const void *p, size_t bytes)
{
    if (!p || !bytes) return false;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_ranges.upper_bound(lo);
    if (it == g_ranges.begin()) return false;
    --it;
    return it->first <= lo && hi <= it->second.hi;      // wholly inside ONE known range
}

int pool_destroy_forced(aeth_pool *p)
{
    if (!p) return AETH_OK;
    int rc = AETH_OK;
    {
        aeth::DeviceGuard g(p->device);
        std::lock_guard<std::mutex> l(p->mu);
        for (void *h : p->owned) {
            pinned_remove(h);
            const hipError_t e = hipHostFree(h);
            if (e != hipSuccess && rc == AETH_OK) rc = hip_fail(e, "hipHostFree");
        }
        p->owned.clear(); p->elems.clear();
    }
    delete p;
    return rc;
}

}  // namespace aeth

extern "C" {

/* pool::make(initial_len, maker, resetter), src/pool.rs:43-69 */
int aeth_pool_create(aeth_ctx *ctx, size_t elem_bytes, size_t initial_len, int flags, aeth_pool **out)
{
    AETH_REQUIRE(ctx && out, AETH_E_ARG, "null argument");
    *out = nullptr;
    AETH_REQUIRE(elem_bytes > 0, AETH_E_ARG, "elem_bytes is zero");
    AETH_REQUIRE((flags & ~AETH_POOL_ZERO_ON_RETURN) == 0, AETH_E_ARG, "unknown pool flags %d", flags);
    aeth_pool *p = new (std::nothrow) aeth_pool();
    AETH_REQUIRE(p, AETH_E_NOMEM, "out of host memory");
    p->device = ctx->device; p->elem_bytes = elem_bytes; p->flags = flags;
    for (size_t i = 0; i < initial_len; i++) {
        void *h = nullptr;
        const int rc = pool_make_locked(p, &h);
        if (rc) { aeth::pool_destroy_forced(p); return rc; }
        if (flags & AETH_POOL_ZERO_ON_RETURN) memset(h, 0, elem_bytes);      /* the resetter runs on the initial elements too (:53-56) */
        p->elems.push_back(h);
    }
    *out = p;
    return AETH_OK;
}

/* The reference's pool lives as long as any Elem holds its Arc; a C handle cannot, so destroying a pool with
 * elements still checked out is refused. */
int aeth_pool_destroy(aeth_pool *pool)
{
    if (!pool) return AETH_OK;
    {
        std::lock_guard<std::mutex> l(pool->mu);
        AETH_REQUIRE(pool->elems.size() == pool->owned.size(), AETH_E_ARG, "%zu pool element(s) still checked out",
                     pool->owned.size() - pool->elems.size());
    }
    return aeth::pool_destroy_forced(pool);
}

/* Pool::take, :78-97: *buf = NULL when the pool is empty (the reference returns None) */
int aeth_pool_take(aeth_pool *pool, void **buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    std::lock_guard<std::mutex> l(pool->mu);
    if (pool->elems.empty()) { *buf = nullptr; return AETH_OK; }
    *buf = pool->elems.back();
    pool->elems.pop_back();
    return AETH_OK;
}

/* Pool::take_or_make, :115-132: grows the pool by one element when it is empty */
int aeth_pool_take_or_make(aeth_pool *pool, void **buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    *buf = nullptr;
    std::lock_guard<std::mutex> l(pool->mu);
    if (pool->elems.empty()) return pool_make_locked(pool, buf);
    *buf = pool->elems.back();
    pool->elems.pop_back();
    return AETH_OK;
}

/* Elem::drop -> PoolInner::give_back, :175-208: reset, then back into the pool */
int aeth_pool_give_back(aeth_pool *pool, void *buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    std::lock_guard<std::mutex> l(pool->mu);
    AETH_REQUIRE(std::find(pool->owned.begin(), pool->owned.end(), buf) != pool->owned.end(), AETH_E_ARG,
                 "pointer is not an element of this pool");
    AETH_REQUIRE(std::find(pool->elems.begin(), pool->elems.end(), buf) == pool->elems.end(), AETH_E_ARG,
                 "element given back twice");
    if (pool->flags & AETH_POOL_ZERO_ON_RETURN) memset(buf, 0, pool->elem_bytes);
    pool->elems.push_back(buf);
    return AETH_OK;
}

size_t aeth_pool_len(aeth_pool *pool)               /* Pool::len, :138-140 */
{
    if (!pool) return 0;
    std::lock_guard<std::mutex> l(pool->mu);
    return pool->elems.size();
}

size_t aeth_pool_cap(aeth_pool *pool)               /* Pool::cap, :157-159 */
{
    if (!pool) return 0;
    std::lock_guard<std::mutex> l(pool->mu);
    return pool->owned.size();
}

size_t aeth_pool_elem_bytes(const aeth_pool *pool) { return pool ? pool->elem_bytes : 0; }

/* Explicit opt-in: page-lock memory the caller owns so that the host pipeline copies from / to it directly.  Whole
 * pages only, and nothing that touches a range this library already knows; the caller keeps the memory alive and
 * mapped until aeth_host_unregister has returned AETH_OK. */
int aeth_host_register(aeth_ctx *ctx, void *ptr, size_t bytes)
{
    AETH_REQUIRE(ctx && ptr && bytes, AETH_E_ARG, "null / empty range");
    const size_t pg = page_size();
    AETH_REQUIRE(((uintptr_t)ptr % pg) == 0 && (bytes % pg) == 0, AETH_E_ALIGN,
                 "range must start on a page boundary and cover whole pages (page size %zu)", pg);
    {
        std::lock_guard<std::mutex> l(g_mu);
        AETH_REQUIRE(!overlaps_locked((uintptr_t)ptr, (uintptr_t)ptr + bytes), AETH_E_ARG,
                     "range overlaps memory that is already registered or belongs to a pinned pool");
    }
    aeth::DeviceGuard g(ctx->device);
    AETH_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    aeth::pinned_add(ptr, bytes, aeth::PIN_REGISTERED);
    return AETH_OK;
}

int aeth_host_unregister(aeth_ctx *ctx, void *ptr)
{
    AETH_REQUIRE(ctx && ptr, AETH_E_ARG, "null argument");
    {
        std::lock_guard<std::mutex> l(g_mu);
        auto it = g_ranges.find((uintptr_t)ptr);
        AETH_REQUIRE(it != g_ranges.end() && it->second.kind == aeth::PIN_REGISTERED, AETH_E_ARG,
                     "pointer was not registered with aeth_host_register");
    }
    aeth::DeviceGuard g(ctx->device);
    // nothing of this context may still be copying from / to the range
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    AETH_HIP(hipHostUnregister(ptr));
    aeth::pinned_remove(ptr);
    return AETH_OK;
}

/* 1 if [ptr, ptr + bytes) lies inside one pool element or one registered range */
int aeth_host_is_pinned(const void *ptr, size_t bytes) { return aeth::host_range_pinned(ptr, bytes) ? 1 : 0; }

}  // extern "C"
