// aeth_internal.h -- shared host-side plumbing for libaether_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "../../include/aether_hip.h"

namespace aeth { struct PipeState; }

struct aeth_ctx {
    int device = 0;
    hipStream_t stream_main = nullptr;   // use aeth::ctx_stream(ctx): it orders the caller behind the overlap lane
    bool owns_stream = false;
    int num_cus = 256;
    // Overlap lane (aeth_ctx_set_overlap): a second HIP queue.  Consecutive fused-FIR launches whose buffers do not
    // touch each other alternate between the two queues, so that the drain of one launch (its last round only
    // stores) runs beside the fill of the next (its first round only loads).  Every launch is ordered behind
    // everything except its immediate predecessor; any other call on the context joins the lane first, so
    // results are those of one in-order stream.
    hipStream_t stream_aux = nullptr;
    hipEvent_t ev_pre[2] = {nullptr, nullptr};   // [lane] recorded on that lane right before its latest FIR launch
    bool ev_pre_empty[2] = {false, false};       // ... unless that launch started on an idle context: nothing was in front of it
    hipEvent_t ev_aux_done = nullptr;
    bool overlap = false;          // feature switch (off for borrowed streams)
    bool stream_shared = false;    // aeth_ctx_stream() handed the main stream to code this library does not see: the
                                   // lane stays out of use until aeth_ctx_set_overlap(ctx, 1) is called again
    bool aux_pending = false;      // the aux lane holds work the main stream is not yet ordered behind
    int chain_last = -1;           // lane of the latest FIR launch while nothing else has been enqueued since; else -1
    bool last_chained = false;     // the latest ctx_fir_lane call put its launch beside its predecessor
    uintptr_t last_in[2] = {0, 0}, last_out[2] = {0, 0};   // [lo, hi) byte ranges of that launch
    unsigned since_sync = 0;       // stream hand-outs (= launches, roughly) since the last aeth_ctx_sync: a short wait blocks, a long one polls
    // host pipeline (aeth_fir_stream_host): stage streams, device slots, events, pinned staging pool, copy threads --
    // created on its first run and kept (aeth_host.h)
    aeth::PipeState *pipe = nullptr;
    // device scratch of the host-slice flavours, grown on demand
    void *stage[2] = {nullptr, nullptr};
    size_t stage_bytes[2] = {0, 0};
    // small host-slice calls: two pinned, device-visible bounce buffers (aeth::HostIO)
    void *bounce[2] = {nullptr, nullptr};
    // plans of the one-shot vec_fft / vec_ifft (aeth_fft.hip: fft_cache_get), most recently used first
    void *fft_cache = nullptr;
};

namespace aeth {

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char *what);

// The context's stream for anything but a chained FIR launch: joins the overlap lane (the main stream waits for
// what the aux lane still holds) and ends the chain.
hipStream_t ctx_stream(aeth_ctx *ctx);
inline hipStream_t ctx_stream(const aeth_ctx *ctx) { return ctx_stream(const_cast<aeth_ctx *>(ctx)); }
// Stream for a fused-FIR launch reading [in_lo, in_hi) and writing [out_lo, out_hi): the other lane when the
// previous call on this context was such a launch and the buffers are disjoint, else the main stream.
hipStream_t ctx_fir_lane(aeth_ctx *ctx, uintptr_t in_lo, uintptr_t in_hi, uintptr_t out_lo, uintptr_t out_hi);

// ensure staging slot `i` holds >= bytes of device memory
int ctx_stage(aeth_ctx *ctx, int i, size_t bytes);
// frees the plans vec_fft / vec_ifft built for this context (aeth_ctx_destroy, aeth_ctx_trim)
void fft_cache_release(aeth_ctx *ctx);

// Buffers of one host-slice call (the literal trait call: host slice in, host slice out, synchronous).
//   small (every buffer <= kZeroCopyMax): the context's two pinned, device-visible bounce buffers -- memcpy in, the kernel
//     reads and writes HOST memory over PCIe, one wait, memcpy out: one launch and one completion round trip per call.
//     Measured (tools/host_latency.hip, 2048 samples): 15.4 us against 31.6 us for pageable H2D -> kernel -> D2H and
//     19.7 us for pinned async copies around a device-resident kernel; the floor (an empty kernel + wait) is 10.4 us.
//   large: device staging and the runtime's copy engines, as before (link-rate-bound, not latency-bound).
constexpr size_t kZeroCopyMax = (size_t)256 << 10;
struct HostIO {
    aeth_ctx *ctx;
    bool pinned;
    void *buf[2] = {nullptr, nullptr};
    // bytes0 / bytes1: sizes of the two buffers the call needs (0: unused)
    int open(aeth_ctx *c, size_t bytes0, size_t bytes1);
    int put(int slot, const void *src, size_t bytes);          // caller memory -> buffer
    int get(void *dst, int slot, size_t bytes);                // after the kernels: wait, buffer -> caller memory
    int wait();                                                // the result stays in the buffer (tfwd / tbwd)
};

// Tuning knobs: environment integers that are consulted ONLY when the process was started with AETH_TUNING=1; a
// normal run never reads them.  libaether_hip.so carries the seven that choose between SHIPPED behaviours, all
// documented in include/aether_hip.h ("Tuning knobs"): AETH_FIR_GRID_FIRST, AETH_FIR_GRID_CHAINED, AETH_FIR_SPREAD,
// AETH_NT, AETH_PIPE_THREADS, AETH_PIPE_MIXED, AETH_SYNC_SPIN_US.
int tuning_int(const char *name, int dflt);
// Lab knobs (tools/tune_*.py, prime_route.py, vs_rocfft.py sweeps): routes and shapes that were measured and not
// kept.  They exist only in the lab build (`make LAB=1` -> lib/libaether_hip_lab.so, -DAETH_LAB=1); in the product
// every lab_int() is its default at compile time, so the paths behind the other values are not in the library.
#if defined(AETH_LAB) && AETH_LAB
#define lab_int tuning_int      /* the lab build reads them like any other knob */
#else
constexpr int lab_int(const char *, int dflt) { return dflt; }
#endif

inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// n / d for 32-bit n by multiply-high and shifts (Granlund-Montgomery round-up form):
// index arithmetic of store-bound kernels must not cost more than their stores.
struct FastDiv {
    uint32_t d, m, sh1, sh2;
};
inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) l++;
    f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f)
{
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// Streamed data (touched exactly once by a kernel) is accessed with the non-temporal hint: no write-allocate,
// no retention in L2 / Infinity Cache.  Measured on MI355X with 256 MiB operands: element-wise kernels 5.9 -> 6.5
// TB/s, batched FFT 5.4 -> 5.8-6.1, interpolate 5.5 -> 6.2, the fused FIR kernel -3 % launch time.  A chain of
// kernels over an operand that FITS the 256 MiB cache wants the opposite (C4's 64 MiB channel: 107 GS/s with
// plain accesses, 96 with the hint), so the hint is a kernel template parameter picked per launch by size.
template <bool NT, typename V> __device__ __forceinline__ V nt_load(const V *p)
{
    static_assert(sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 16, "4-, 8- or 16-byte accesses");
    if constexpr (!NT) return *p;
    else {
        typedef unsigned VT __attribute__((ext_vector_type(sizeof(V) / 4)));
        if constexpr (sizeof(V) == 4) return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p)));
        else return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const VT *>(p)));
    }
}
template <bool NT, typename V> __device__ __forceinline__ void nt_store(V *p, V v)
{
    static_assert(sizeof(V) == 4 || sizeof(V) == 8 || sizeof(V) == 16, "4-, 8- or 16-byte accesses");
    if constexpr (!NT) *p = v;
    else {
        typedef unsigned VT __attribute__((ext_vector_type(sizeof(V) / 4)));
        if constexpr (sizeof(V) == 4) __builtin_nontemporal_store(__builtin_bit_cast(unsigned, v), reinterpret_cast<unsigned *>(p));
        else __builtin_nontemporal_store(__builtin_bit_cast(VT, v), reinterpret_cast<VT *>(p));
    }
}

// does a launch that moves `bytes` stream past the cache?  (half the 256 MiB Infinity Cache; AETH_NT=0/1 forces it
// under AETH_TUNING=1)
inline bool streams_past_cache(size_t bytes)
{
    const int forced = tuning_int("AETH_NT", -1);
    if (forced >= 0) return forced != 0;
    return bytes > ((size_t)128 << 20);
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace aeth

#define AETH_HIP(call)                                              \
    do {                                                            \
        hipError_t e__ = (call);                                    \
        if (e__ != hipSuccess) return aeth::hip_fail(e__, #call);   \
    } while (0)

#define AETH_REQUIRE(cond, code, ...)                               \
    do {                                                            \
        if (!(cond)) return aeth::set_error((code), __VA_ARGS__);   \
    } while (0)

// message texts of the reference's panics (kept verbatim for the binding)
#define AETH_MSG_VEC_LEN "Vectors must have same length"            /* src/vecops.rs:100-104 */
#define AETH_MSG_FFT_LEN "Input and FFT must be the same length"    /* src/fft.rs:163-167   */
#define AETH_MSG_DECIM   "Only even decimations are supported"      /* src/sampling.rs:32-36 */
