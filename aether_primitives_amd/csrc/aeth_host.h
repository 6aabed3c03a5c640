// aeth_host.h -- host-side machinery of the stream pipeline (not installed): the registry of page-locked host
// ranges (aeth_pool.hip), the team of copy threads and the per-context pipeline state (aeth_pipeline.hip).
#pragma once

#include "aeth_internal.h"
#include "aeth_hostcore.h"

struct aeth_pool;
struct aeth_fir;

namespace aeth {

using hostcore::PIN_POOL;
using hostcore::PIN_REGISTERED;
void pinned_add(const void *p, size_t bytes, int kind);
void pinned_remove(const void *p);
// is [p, p + bytes) wholly inside one pool element or one explicitly registered range?
bool host_range_pinned(const void *p, size_t bytes);
// frees every element whether checked out or not (context teardown)
int pool_destroy_forced(aeth_pool *p);

using hostcore::CopyTeam;     // aeth_hostcore.h: the host threads of the copy-in / copy-out stages

constexpr int kPipeSlots = 3;

// what a context keeps between runs of the host pipeline: the three stage streams, the device slots, their events,
// the pinned staging pool and the copy threads -- all created on first use, none per call
struct PipeState {
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};            // upload | kernel | download
    float2 *din[kPipeSlots] = {nullptr, nullptr, nullptr};          // device: [history | chunk]
    float2 *dout[kPipeSlots] = {nullptr, nullptr, nullptr};
    size_t din_bytes = 0, dout_bytes = 0;
    void *mid[kPipeSlots][2] = {};                                   // scratch pair per slot (a chain of ops as the compute stage)
    size_t mid_bytes = 0;
    hipEvent_t up[kPipeSlots] = {}, ran[kPipeSlots] = {}, down[kPipeSlots] = {};
    aeth_pool *pool[2] = {nullptr, nullptr};                         // pinned staging elements per side, in | out (src/pool.rs)
    CopyTeam *team = nullptr;
};
void pipe_release(aeth_ctx *ctx);      // aeth_ctx_destroy, aeth_ctx_trim
void pipe_fail_arm(int n);             // test hook: the n-th staging element taken from now on fails once
bool pipe_fail_after_take();

// aeth_fir.hip: the fused kernel on an explicit stream (no overlap lane)
int fir_exec_on(aeth_fir *f, hipStream_t stream, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out);

}  // namespace aeth
