// aeth_host.h -- host-side machinery of the stream pipeline (not installed): the registry of page-locked host
// ranges (aeth_pool.hip), the team of copy threads and the per-context pipeline state (aeth_pipeline.hip).
#pragma once

#include "aeth_internal.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct aeth_pool;
struct aeth_fir;

namespace aeth {

enum { PIN_POOL = 1, PIN_REGISTERED = 2 };
void pinned_add(const void *p, size_t bytes, int kind);
void pinned_remove(const void *p);
// is [p, p + bytes) wholly inside one pool element or one explicitly registered range?
bool host_range_pinned(const void *p, size_t bytes);
// frees every element whether checked out or not (context teardown)
int pool_destroy_forced(aeth_pool *p);

// Host threads that move slices between caller memory and pinned staging elements: the reference runs one thread
// per pipeline stage (src/pipeline.rs:52-119); a PCIe link outruns one core's memcpy several times over, so each of
// the two host stages (copy-in, copy-out) is served by the whole team, a slice at a time.
class CopyTeam {
public:
    explicit CopyTeam(int nthreads);
    ~CopyTeam();
    CopyTeam(const CopyTeam &) = delete;
    CopyTeam &operator=(const CopyTeam &) = delete;
    // dst <- src in slices; *pending is raised by the number of slices now and lowered (release) as each completes
    void submit(void *dst, const void *src, size_t bytes, std::atomic<int> *pending);
    int threads() const { return (int)th_.size(); }

private:
    struct Job { void *dst; const void *src; size_t bytes; std::atomic<int> *pending; };
    void run();
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> q_;
    bool stop_ = false;
};

constexpr int kPipeSlots = 3;

// what a context keeps between runs of the host pipeline: the three stage streams, the device slots, their events,
// the pinned staging pool and the copy threads -- all created on first use, none per call
struct PipeState {
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};            // upload | kernel | download
    float2 *din[kPipeSlots] = {nullptr, nullptr, nullptr};          // device: [history | chunk]
    float2 *dout[kPipeSlots] = {nullptr, nullptr, nullptr};
    size_t din_bytes = 0, dout_bytes = 0;
    hipEvent_t up[kPipeSlots] = {}, ran[kPipeSlots] = {}, down[kPipeSlots] = {};
    aeth_pool *pool = nullptr;                                       // pinned staging elements (src/pool.rs)
    CopyTeam *team = nullptr;
};
void pipe_release(aeth_ctx *ctx);      // aeth_ctx_destroy

// aeth_fir.hip: the fused kernel on an explicit stream (no overlap lane)
int fir_exec_on(aeth_fir *f, hipStream_t stream, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out);

}  // namespace aeth
