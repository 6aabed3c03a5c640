// aeth_runtime.hip -- context, device memory, events, error reporting.
// Host-side runtime of libaether_hip.so; the GPU analogue of what the reference
// keeps implicit in Rust ownership (Vec<cf32>, Cfft.tmp: src/fft.rs:134-159).
#include "aeth_internal.h"

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
    if (ctx->pin_bytes[i] < bytes) {
        size_t want = bytes + bytes / 4;
        if (ctx->pin[i]) { AETH_HIP(hipHostFree(ctx->pin[i])); ctx->pin[i] = nullptr; ctx->pin_bytes[i] = 0; }
        AETH_HIP(hipHostMalloc(&ctx->pin[i], want, hipHostMallocDefault));
        ctx->pin_bytes[i] = want;
    }
    return AETH_OK;
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
    AETH_HIP(hipSetDevice(device));
    aeth_ctx *c = new (std::nothrow) aeth_ctx();
    AETH_REQUIRE(c, AETH_E_NOMEM, "out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->num_cus = prop.multiProcessorCount;
    if (borrow) {
        c->stream = borrowed;
        c->owns_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return aeth::hip_fail(e, "hipStreamCreateWithFlags"); }
        c->owns_stream = true;
    }
    *out = c;
    return AETH_OK;
}

int aeth_ctx_create(int device, aeth_ctx **out) { return ctx_make(device, nullptr, false, out); }

int aeth_ctx_create_on_stream(int device, void *hip_stream, aeth_ctx **out)
{
    return ctx_make(device, (hipStream_t)hip_stream, true, out);
}

int aeth_ctx_destroy(aeth_ctx *ctx)
{
    if (!ctx) return AETH_OK;
    aeth::DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i]) (void)hipFree(ctx->stage[i]);
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
    }
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return AETH_OK;
}

int aeth_ctx_sync(aeth_ctx *ctx)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

void *aeth_ctx_stream(aeth_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
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
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    AETH_HIP(hipFree(dptr));
    return AETH_OK;
}

int aeth_upload(aeth_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_dev && src_host, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

int aeth_download(aeth_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_host && src_dev, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AETH_HIP(hipStreamSynchronize(ctx->stream));
    return AETH_OK;
}

int aeth_copy_dev(aeth_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard dev_guard(ctx->device);
    if (bytes == 0) return AETH_OK;
    AETH_REQUIRE(dst_dev && src_dev, AETH_E_ARG, "null pointer");
    AETH_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
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
    AETH_HIP(hipEventRecord(ev->ev, ev->ctx->stream));
    return AETH_OK;
}

int aeth_event_sync(aeth_event *ev)
{
    AETH_REQUIRE(ev, AETH_E_ARG, "event is null");
    AETH_HIP(hipEventSynchronize(ev->ev));
    return AETH_OK;
}

int aeth_event_elapsed_ms(aeth_event *start, aeth_event *stop, float *ms)
{
    AETH_REQUIRE(start && stop && ms, AETH_E_ARG, "null argument");
    AETH_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
    return AETH_OK;
}

}  // extern "C"
