// aeth_runtime.hip -- context, device memory, events, error reporting.
// Host-side runtime of libaether_hip.so; the GPU analogue of what the reference
// keeps implicit in Rust ownership (Vec<cf32>, Cfft.tmp: src/fft.rs:134-159).
#include "aeth_internal.h"
#include "aeth_host.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {
thread_local char g_err[512] = "";
}

namespace aeth {

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return set_error(e == hipErrorOutOfMemory ? AETH_E_NOMEM : AETH_E_HIP, "HIP error %d (%s) in %s",
                     (int)e, hipGetErrorString(e), what);
}

int tuning_int(const char *name, int dflt)
{
    static const bool enabled = [] { const char *e = getenv("AETH_TUNING"); return e && atoi(e) != 0; }();
    if (!enabled) return dflt;
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

hipStream_t ctx_stream(aeth_ctx *ctx)
{
    if (ctx->aux_pending) {
        DeviceGuard g(ctx->device);
        // the main stream continues behind everything the aux lane holds
        if (hipEventRecord(ctx->ev_aux_done, ctx->stream_aux) == hipSuccess)
            (void)hipStreamWaitEvent(ctx->stream_main, ctx->ev_aux_done, 0);
        else (void)hipStreamSynchronize(ctx->stream_aux);
        ctx->aux_pending = false;
    }
    ctx->chain_last = -1;
    ctx->last_chained = false;
    ctx->since_sync++;
    return ctx->stream_main;
}

static inline bool ranges_touch(uintptr_t a_lo, uintptr_t a_hi, uintptr_t b_lo, uintptr_t b_hi)
{
    return a_lo < b_hi && b_lo < a_hi;
}

hipStream_t ctx_fir_lane(aeth_ctx *ctx, uintptr_t in_lo, uintptr_t in_hi, uintptr_t out_lo, uintptr_t out_hi)
{
    if (!ctx->overlap || ctx->stream_shared) return ctx_stream(ctx);
    const int prev = ctx->chain_last;
    bool chained = prev >= 0;
    if (chained) {
        // the only launch this one is NOT ordered behind is its immediate predecessor: their buffers must be disjoint
        const uintptr_t *pi = ctx->last_in, *po = ctx->last_out;
        if (ranges_touch(in_lo, in_hi, po[0], po[1]) || ranges_touch(out_lo, out_hi, pi[0], pi[1]) ||
            ranges_touch(out_lo, out_hi, po[0], po[1]))
            chained = false;
    }
    int lane = 0;
    hipStream_t s;
    // nothing has been handed a stream since the last aeth_ctx_sync: both queues are idle and this launch has no history
    const bool idle = ctx->since_sync == 0 && !ctx->aux_pending;
    if (!chained) {
        s = ctx_stream(ctx);                    // joins; everything enqueued so far is in front of this launch
    } else {
        lane = 1 - prev;
        s = lane ? ctx->stream_aux : ctx->stream_main;
        // behind everything the other lane held BEFORE its latest launch (which itself runs beside this one) -- if it
        // held anything: a launch that started on an idle context has nothing in front of it, and no wait packet then
        // sits in front of this kernel
        if (!ctx->ev_pre_empty[prev] && hipStreamWaitEvent(s, ctx->ev_pre[prev], 0) != hipSuccess) { s = ctx_stream(ctx); lane = 0; }
    }
    if (!chained && idle) {
        // the head of a chain on an idle context: no history to mark, so no event-record packet in front of the kernel
        // (measured: a K = 1 region 68.1 us with the packet, 65.1 us without, tools/k20_lab.py)
        ctx->ev_pre_empty[lane] = true;
    } else {
        ctx->ev_pre_empty[lane] = false;
        if (hipEventRecord(ctx->ev_pre[lane], s) != hipSuccess) {   // this lane's history in front of the launch
            s = ctx_stream(ctx); lane = 0;
            ctx->chain_last = -1;                   // no chain without the event
            return s;
        }
    }
    if (lane == 1) ctx->aux_pending = true;
    ctx->since_sync++;
    ctx->last_chained = chained;
    ctx->chain_last = lane;
    ctx->last_in[0] = in_lo; ctx->last_in[1] = in_hi;
    ctx->last_out[0] = out_lo; ctx->last_out[1] = out_hi;
    return s;
}

int ctx_stage(aeth_ctx *ctx, int i, size_t bytes)
{
    DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) bytes = 16;
    if (ctx->stage_bytes[i] < bytes) {
        size_t want = bytes + bytes / 4;
        if (ctx->stage[i]) { AETH_HIP(hipFree(ctx->stage[i])); ctx->stage[i] = nullptr; ctx->stage_bytes[i] = 0; }
        AETH_HIP(hipMalloc(&ctx->stage[i], want));
        ctx->stage_bytes[i] = want;
    }
    return AETH_OK;
}

int HostIO::open(aeth_ctx *c, size_t bytes0, size_t bytes1)
{
    ctx = c;
    pinned = bytes0 <= kZeroCopyMax && bytes1 <= kZeroCopyMax;
    const size_t want[2] = {bytes0, bytes1};
    for (int i = 0; i < 2; i++) {
        if (!want[i]) continue;
        if (pinned) {
            if (!c->bounce[i]) AETH_HIP(hipHostMalloc(&c->bounce[i], kZeroCopyMax, hipHostMallocDefault));    // coherent: the wait below publishes the kernel's stores
            buf[i] = c->bounce[i];
        } else {
            int rc = ctx_stage(c, i, want[i]); if (rc) return rc;
            buf[i] = c->stage[i];
        }
    }
    return AETH_OK;
}

int HostIO::put(int slot, const void *src, size_t bytes)
{
    if (!bytes) return AETH_OK;
    if (pinned) { memcpy(buf[slot], src, bytes); return AETH_OK; }
    AETH_HIP(hipMemcpyAsync(buf[slot], src, bytes, hipMemcpyHostToDevice, ctx_stream(ctx)));
    return AETH_OK;
}

int HostIO::wait()
{
    AETH_HIP(hipStreamSynchronize(ctx_stream(ctx)));
    return AETH_OK;
}

int HostIO::get(void *dst, int slot, size_t bytes)
{
    if (pinned) {
        int rc = wait(); if (rc) return rc;
        memcpy(dst, buf[slot], bytes);
        return AETH_OK;
    }
    if (bytes) AETH_HIP(hipMemcpyAsync(dst, buf[slot], bytes, hipMemcpyDeviceToHost, ctx_stream(ctx)));
    return wait();
}

}  // namespace aeth

using aeth::set_error;

extern "C" {

const char *aeth_last_error(void) { return g_err; }

int aeth_version(void) { return 0x000100; }

int aeth_device_count(int *count)
{
    AETH_REQUIRE(count, AETH_E_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return aeth::hip_fail(e, "hipGetDeviceCount"); }
    *count = n;
    return AETH_OK;
}

static int ctx_make(int device, hipStream_t borrowed, bool borrow, aeth_ctx **out)
{
    AETH_REQUIRE(out, AETH_E_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    AETH_HIP(hipGetDeviceCount(&n));
    AETH_REQUIRE(device >= 0 && device < n, AETH_E_ARG, "device %d out of range (have %d)", device, n);
    aeth::DeviceGuard dev_guard(device);        // the caller's current device is restored on return
    AETH_REQUIRE(dev_guard.ok, AETH_E_HIP, "hipSetDevice(%d) failed", device);
    aeth_ctx *c = new (std::nothrow) aeth_ctx();
    AETH_REQUIRE(c, AETH_E_NOMEM, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->num_cus = prop.multiProcessorCount;
    if (borrow) {
        c->stream_main = borrowed;
        c->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream_main, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return aeth::hip_fail(e, "hipStreamCreateWithFlags"); }
        c->owns_stream = true;
    }
    *out = c;
    return AETH_OK;
}

static void overlap_release(aeth_ctx *c)
{
    if (c->stream_aux) { (void)hipStreamSynchronize(c->stream_aux); (void)hipStreamDestroy(c->stream_aux); c->stream_aux = nullptr; }
    for (int i = 0; i < 2; i++) if (c->ev_pre[i]) { (void)hipEventDestroy(c->ev_pre[i]); c->ev_pre[i] = nullptr; }
    if (c->ev_aux_done) { (void)hipEventDestroy(c->ev_aux_done); c->ev_aux_done = nullptr; }
    c->overlap = false; c->aux_pending = false; c->chain_last = -1;
}

int aeth_ctx_set_overlap(aeth_ctx *ctx, int enable)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard g(ctx->device);
    (void)aeth::ctx_stream(ctx);                 // join whatever is in flight
    if (!enable) {
        if (ctx->stream_aux) { AETH_HIP(hipStreamSynchronize(ctx->stream_main)); overlap_release(ctx); }
        return AETH_OK;
    }
    // a borrowed stream receives work this library does not see, so nothing can be reordered around it
    AETH_REQUIRE(ctx->owns_stream, AETH_E_UNSUPPORTED, "overlap needs a context that owns its stream");
    ctx->stream_shared = false;                  // (re-)armed by the caller: whoever holds the stream pointer has been told
    if (ctx->overlap) return AETH_OK;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream_aux, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->ev_pre[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_aux_done, hipEventDisableTiming);
    if (e != hipSuccess) { overlap_release(ctx); return aeth::hip_fail(e, "overlap lane set-up"); }
    ctx->overlap = true;
    return AETH_OK;
}

/* 1 only while the lane is in USE: handing the stream out (aeth_ctx_stream) parks it until it is re-armed */
int aeth_ctx_overlap(const aeth_ctx *ctx) { return ctx && ctx->overlap && !ctx->stream_shared ? 1 : 0; }

int aeth_ctx_create(int device, aeth_ctx **out) { return ctx_make(device, nullptr, false, out); }

int aeth_ctx_create_on_stream(int device, void *hip_stream, aeth_ctx **out)
{
    return ctx_make(device, (hipStream_t)hip_stream, true, out);
}

int aeth_ctx_destroy(aeth_ctx *ctx)
{
    if (!ctx) return AETH_OK;
    aeth::DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(aeth::ctx_stream(ctx));
    aeth::fft_cache_release(ctx);
    overlap_release(ctx);
    aeth::pipe_release(ctx);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i]) (void)hipFree(ctx->stage[i]);
        if (ctx->bounce[i]) (void)hipHostFree(ctx->bounce[i]);
    }
    if (ctx->owns_stream && ctx->stream_main) (void)hipStreamDestroy(ctx->stream_main);
    delete ctx;
    return AETH_OK;
}

int aeth_ctx_sync(aeth_ctx *ctx)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard g(ctx->device);
    // With work on the aux lane the host waits for BOTH queues itself instead of first enqueueing a join (event
    // record on one queue, wait on the other) and then waiting for that: the join's two packets only run once the
    // last kernel has ended and sit between it and the moment this call returns.  Once both queues are idle the
    // lanes are trivially ordered, so the chain simply ends here.
    hipStream_t waits[2] = {ctx->stream_main, nullptr};
    int nw = 1;
    if (ctx->aux_pending) { waits[nw++] = ctx->stream_aux; }
    // A batch of launches is waited for by polling: the wake-up latency of a blocking wait is a visible share of a
    // millisecond-long region (K = 20 fused-FIR launches: 985 us polled, 998 us blocked; tools/k20_lab.py).  After
    // AETH_SYNC_SPIN_US (tuning; default 2000) the blocking wait takes over.  One or two short launches on one queue
    // -- the literal "call, then wait" of a device-resident trait call -- are the other way round: the runtime's own
    // wait spins on the completion signal and returns 6 us sooner than a hipStreamQuery loop (10.4 against 16.5 us for
    // an empty kernel, tools/host_latency.hip); equal from 70 us on.
    const bool few = ctx->since_sync <= 2 && !ctx->aux_pending;
    ctx->since_sync = 0;
    const int spin_us = few ? 0 : aeth::tuning_int("AETH_SYNC_SPIN_US", 2000);
    bool done[2] = {false, false};
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            bool all = true;
            for (int i = nw - 1; i >= 0; i--) {
                if (done[i]) continue;
                const hipError_t q = hipStreamQuery(waits[i]);
                if (q == hipSuccess) done[i] = true;
                else if (q == hipErrorNotReady) all = false;
                else return aeth::hip_fail(q, "hipStreamQuery");
            }
            if (all) break;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
        }
        (void)hipGetLastError();                                 // hipErrorNotReady is not an error
    }
    for (int i = nw - 1; i >= 0; i--)
        if (!done[i]) AETH_HIP(hipStreamSynchronize(waits[i]));
    ctx->aux_pending = false;
    ctx->chain_last = -1;
    ctx->last_chained = false;
    return AETH_OK;
}

/* Hands the stream to code this library does not see (torch ops on an ExternalStream, the caller's own copies):
 * joined first, and from here on every launch stays on this one stream -- work the caller enqueues on it is ordered
 * behind everything the library launched before AND after.  The overlap lane comes back only when the caller asks
 * for it again (aeth_ctx_set_overlap(ctx, 1)), i.e. knows that launches may then run beside the stream it holds. */
void *aeth_ctx_stream(aeth_ctx *ctx)
{
    if (!ctx) return nullptr;
    hipStream_t s = aeth::ctx_stream(ctx);
    if (ctx->overlap) ctx->stream_shared = true;
    return (void *)s;
}
int aeth_ctx_device(const aeth_ctx *ctx) { return ctx ? ctx->device : -1; }

int aeth_dev_alloc(aeth_ctx *ctx, size_t bytes, void **dptr)
{
    AETH_REQUIRE(ctx && dptr, AETH_E_ARG, "null argument");
    *dptr = nullptr;
    aeth::DeviceGuard g(ctx->device);
    AETH_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return AETH_OK;
}

int aeth_dev_free(aeth_ctx *ctx, void *dptr)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (!dptr) return AETH_OK;
    aeth::DeviceGuard g(ctx->device);
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    AETH_HIP(hipFree(dptr));
    return AETH_OK;
}

int aeth_upload(aeth_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_dev && src_host, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, aeth::ctx_stream(ctx)));
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    return AETH_OK;
}

int aeth_download(aeth_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_host && src_dev, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, aeth::ctx_stream(ctx)));
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    return AETH_OK;
}

int aeth_copy_dev(aeth_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_dev && src_dev, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, aeth::ctx_stream(ctx)));
    return AETH_OK;
}

/* ---- events -------------------------------------------------------------- */
}  // extern "C"

struct aeth_event {
    aeth_ctx *ctx;
    hipEvent_t ev;
};

extern "C" {

int aeth_event_create(aeth_ctx *ctx, aeth_event **out)
{
    AETH_REQUIRE(ctx && out, AETH_E_ARG, "null argument");
    aeth_event *e = new (std::nothrow) aeth_event();
    AETH_REQUIRE(e, AETH_E_NOMEM, "out of host memory");
    e->ctx = ctx;
    aeth::DeviceGuard g(ctx->device);
    hipError_t r = hipEventCreate(&e->ev);
    if (r != hipSuccess) { delete e; return aeth::hip_fail(r, "hipEventCreate"); }
    *out = e;
    return AETH_OK;
}

int aeth_event_destroy(aeth_event *ev)
{
    if (!ev) return AETH_OK;
    (void)hipEventDestroy(ev->ev);
    delete ev;
    return AETH_OK;
}

int aeth_event_record(aeth_event *ev)
{
    AETH_REQUIRE(ev, AETH_E_ARG, "event is null");
    aeth::DeviceGuard g(ev->ctx->device);
    AETH_HIP(hipEventRecord(ev->ev, aeth::ctx_stream(ev->ctx)));   // behind both lanes
    return AETH_OK;
}

int aeth_event_sync(aeth_event *ev)
{
    AETH_REQUIRE(ev, AETH_E_ARG, "event is null");
    aeth::DeviceGuard g(ev->ctx->device);
    AETH_HIP(hipEventSynchronize(ev->ev));
    return AETH_OK;
}

int aeth_event_elapsed_ms(aeth_event *start, aeth_event *stop, float *ms)
{
    AETH_REQUIRE(start && stop && ms, AETH_E_ARG, "null argument");
    aeth::DeviceGuard g(start->ctx->device);
    AETH_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return AETH_OK;
}

}  // extern "C"
