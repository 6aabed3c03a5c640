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

#include <new>

// The pool, the range registry and their locking live in aeth_hostcore.h (no HIP in there: that header also builds
// under the thread and address sanitizers, tests/test_hostcore_sanitizers.py); this file supplies the pinned allocator
// and the C ABI's argument checks and error texts.
struct aeth_pool {
    int device = 0;                     // of the context it was made on (kept by value: a pool may outlive its context)
    int flags = 0;
    aeth::hostcore::PoolCore *core = nullptr;
};

namespace {

namespace hc = aeth::hostcore;

size_t page_size()
{
    static const size_t p = [] { long v = sysconf(_SC_PAGESIZE); return v > 0 ? (size_t)v : (size_t)4096; }();
    return p;
}

// PinHooks over hipHostMalloc / hipHostFree; `user` is the pool (for its device)
int pin_alloc(void *user, void **out, size_t bytes)
{
    aeth::DeviceGuard g(static_cast<aeth_pool *>(user)->device);
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    return e == hipSuccess ? 0 : aeth::hip_fail(e, "hipHostMalloc");
}

int pin_release(void *user, void *p)
{
    aeth::DeviceGuard g(static_cast<aeth_pool *>(user)->device);
    const hipError_t e = hipHostFree(p);
    return e == hipSuccess ? 0 : aeth::hip_fail(e, "hipHostFree");
}

}  // namespace

namespace aeth {

void pinned_add(const void *p, size_t bytes, int kind) { hc::ranges().add(p, bytes, kind); }
void pinned_remove(const void *p) { hc::ranges().remove(p); }
bool host_range_pinned(const void *p, size_t bytes) { return hc::ranges().contains(p, bytes); }

int pool_destroy_forced(aeth_pool *p)
{
    if (!p) return AETH_OK;
    int rc = p->core ? p->core->release_all() : AETH_OK;
    delete p->core;
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
    p->device = ctx->device; p->flags = flags;
    hc::PinHooks hooks; hooks.alloc = pin_alloc; hooks.release = pin_release; hooks.user = p;
    p->core = new (std::nothrow) hc::PoolCore(elem_bytes, (flags & AETH_POOL_ZERO_ON_RETURN) != 0, hooks);
    if (!p->core) { delete p; return aeth::set_error(AETH_E_NOMEM, "out of host memory"); }
    const int rc = p->core->prefill(initial_len);             /* the resetter runs on the initial elements too (:53-56) */
    if (rc) { aeth::pool_destroy_forced(p); return rc; }
    *out = p;
    return AETH_OK;
}

/* The reference's pool lives as long as any Elem holds its Arc; a C handle cannot, so destroying a pool with
 * elements still checked out is refused. */
int aeth_pool_destroy(aeth_pool *pool)
{
    if (!pool) return AETH_OK;
    const size_t out = pool->core->checked_out();
    AETH_REQUIRE(out == 0, AETH_E_ARG, "%zu pool element(s) still checked out", out);
    return aeth::pool_destroy_forced(pool);
}

/* Pool::take, :78-97: *buf = NULL when the pool is empty (the reference returns None) */
int aeth_pool_take(aeth_pool *pool, void **buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    *buf = pool->core->take();
    return AETH_OK;
}

/* Pool::take_or_make, :115-132: grows the pool by one element when it is empty */
int aeth_pool_take_or_make(aeth_pool *pool, void **buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    return pool->core->take_or_make(buf);
}

/* Elem::drop -> PoolInner::give_back, :175-208: reset, then back into the pool */
int aeth_pool_give_back(aeth_pool *pool, void *buf)
{
    AETH_REQUIRE(pool && buf, AETH_E_ARG, "null argument");
    const int r = pool->core->give_back(buf);
    AETH_REQUIRE(r != hc::PoolCore::NOT_ELEMENT, AETH_E_ARG, "pointer is not an element of this pool");
    AETH_REQUIRE(r != hc::PoolCore::GIVEN_TWICE, AETH_E_ARG, "element given back twice");
    return AETH_OK;
}

size_t aeth_pool_len(aeth_pool *pool) { return pool ? pool->core->len() : 0; }               /* Pool::len, :138-140 */
size_t aeth_pool_cap(aeth_pool *pool) { return pool ? pool->core->cap() : 0; }               /* Pool::cap, :157-159 */
size_t aeth_pool_elem_bytes(const aeth_pool *pool) { return pool ? pool->core->elem_bytes() : 0; }

/* Explicit opt-in: page-lock memory the caller owns so that the host pipeline copies from / to it directly.  Whole
 * pages only, and nothing that touches a range this library already knows; the caller keeps the memory alive and
 * mapped until aeth_host_unregister has returned AETH_OK. */
int aeth_host_register(aeth_ctx *ctx, void *ptr, size_t bytes)
{
    AETH_REQUIRE(ctx && ptr && bytes, AETH_E_ARG, "null / empty range");
    const size_t pg = page_size();
    AETH_REQUIRE(((uintptr_t)ptr % pg) == 0 && (bytes % pg) == 0, AETH_E_ALIGN,
                 "range must start on a page boundary and cover whole pages (page size %zu)", pg);
    // check and claim under ONE lock (two threads cannot both pass the check with overlapping ranges); the claim is
    // pending -- invisible to aeth_host_is_pinned and the pipeline -- until the runtime has locked the pages
    AETH_REQUIRE(hc::ranges().try_claim(ptr, bytes, hc::PIN_REGISTERED), AETH_E_ARG,
                 "range overlaps memory that is already registered or belongs to a pinned pool");
    aeth::DeviceGuard g(ctx->device);
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { hc::ranges().remove(ptr); return aeth::hip_fail(e, "hipHostRegister"); }
    hc::ranges().confirm(ptr);
    return AETH_OK;
}

int aeth_host_unregister(aeth_ctx *ctx, void *ptr)
{
    AETH_REQUIRE(ctx && ptr, AETH_E_ARG, "null argument");
    AETH_REQUIRE(hc::ranges().kind_at(ptr) == hc::PIN_REGISTERED, AETH_E_ARG, "pointer was not registered with aeth_host_register");
    aeth::DeviceGuard g(ctx->device);
    // nothing of this context may still be copying from / to the range
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    AETH_HIP(hipHostUnregister(ptr));
    hc::ranges().remove(ptr);
    return AETH_OK;
}

/* 1 if [ptr, ptr + bytes) lies inside one pool element or one registered range */
int aeth_host_is_pinned(const void *ptr, size_t bytes) { return aeth::host_range_pinned(ptr, bytes) ? 1 : 0; }

}  // extern "C"
