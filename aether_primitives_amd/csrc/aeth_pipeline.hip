// aeth_pipeline.hip -- host-resident stream through the device at link rate (SURVEY 8f "next" #4).
//
// The GPU-side counterpart of the reference's thread-per-stage pipeline over pooled buffers
// (src/pipeline.rs:52-137, src/pool.rs:43-221, examples/pipeline.rs:36-52).  Five stages, chunk by chunk:
//
//     copy-in        caller slice  -> pinned pool element          host threads (CopyTeam)
//     upload         pinned        -> device slot                  HIP stream 0 (H2D copy engine)
//     kernel         fused FFT*H*IFFT on the slot                  HIP stream 1
//     download       device slot   -> pinned pool element          HIP stream 2 (D2H copy engine)
//     copy-out       pinned        -> caller slice                 host threads
//
// Slots are handed from stage to stage by events (device stages) and completion counters (host stages); the
// calling thread only enqueues and polls.  A side whose caller memory is ALREADY page-locked -- it lies inside an
// element of an aeth_pool or a range registered with aeth_host_register -- skips its host stage and is copied
// from / to directly.  Caller memory is never registered here (see aeth_pool.hip for why).
#include "aeth_internal.h"
#include "aeth_host.h"
#include "aeth_fft_plan.h"

#include <sched.h>

#include <chrono>
#include <cstring>
#include <memory>
#include <new>

namespace aeth {

// ---- copy threads ---------------------------------------------------------------------------
CopyTeam::CopyTeam(int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    for (int i = 0; i < nthreads; i++) th_.emplace_back([this] { run(); });
}

CopyTeam::~CopyTeam()
{
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
}

void CopyTeam::submit(void *dst, const void *src, size_t bytes, std::atomic<int> *pending)
{
    if (bytes == 0) return;
    // slices of 1-4 MiB: enough of them for every thread, each long enough to amortise the hand-over
    size_t slice = bytes / (size_t)(2 * th_.size());
    const size_t lo = (size_t)1 << 20, hi = (size_t)4 << 20;
    slice = slice < lo ? lo : (slice > hi ? hi : slice);
    slice = (slice + 4095) & ~(size_t)4095;
    const int n = (int)((bytes + slice - 1) / slice);
    pending->fetch_add(n, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> l(mu_);
        for (size_t off = 0; off < bytes; off += slice)
            q_.push_back(Job{(char *)dst + off, (const char *)src + off, bytes - off < slice ? bytes - off : slice, pending});
    }
    if (n > 1) cv_.notify_all(); else cv_.notify_one();
}

void CopyTeam::run()
{
    for (;;) {
        Job j;
        {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [this] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;             // stop_ and nothing left
            j = q_.front();
            q_.pop_front();
        }
        memcpy(j.dst, j.src, j.bytes);
        j.pending->fetch_sub(1, std::memory_order_release);
    }
}

void pipe_release(aeth_ctx *ctx)
{
    PipeState *p = ctx->pipe;
    if (!p) return;
    delete p->team;                             // joins the threads
    for (int i = 0; i < 3; i++)
        if (p->stream[i]) { (void)hipStreamSynchronize(p->stream[i]); (void)hipStreamDestroy(p->stream[i]); }
    for (int s = 0; s < kPipeSlots; s++) {
        if (p->din[s]) (void)hipFree(p->din[s]);
        if (p->dout[s]) (void)hipFree(p->dout[s]);
        if (p->up[s]) (void)hipEventDestroy(p->up[s]);
        if (p->ran[s]) (void)hipEventDestroy(p->ran[s]);
        if (p->down[s]) (void)hipEventDestroy(p->down[s]);
    }
    (void)pool_destroy_forced(p->pool);
    delete p;
    ctx->pipe = nullptr;
}

}  // namespace aeth

namespace {

using aeth::kPipeSlots;
using aeth::PipeState;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// streams, events, device slots of `din_bytes` / `dout_bytes`, grown (never shrunk) between calls
int pipe_prepare(aeth_ctx *ctx, size_t din_bytes, size_t dout_bytes, int nslots)
{
    if (!ctx->pipe) {
        ctx->pipe = new (std::nothrow) PipeState();
        AETH_REQUIRE(ctx->pipe, AETH_E_NOMEM, "out of host memory");
    }
    PipeState *p = ctx->pipe;
    for (int i = 0; i < 3; i++)
        if (!p->stream[i]) AETH_HIP(hipStreamCreateWithFlags(&p->stream[i], hipStreamNonBlocking));
    for (int s = 0; s < kPipeSlots; s++) {
        if (!p->up[s]) AETH_HIP(hipEventCreateWithFlags(&p->up[s], hipEventDisableTiming));
        if (!p->ran[s]) AETH_HIP(hipEventCreateWithFlags(&p->ran[s], hipEventDisableTiming));
        if (!p->down[s]) AETH_HIP(hipEventCreateWithFlags(&p->down[s], hipEventDisableTiming));
    }
    // every slot has the size on record (the largest asked for so far): a slot allocated later, for a run with more
    // chunks, must not be smaller than the ones a previous run sized
    const size_t want_in = p->din_bytes > din_bytes ? p->din_bytes : din_bytes;
    const size_t want_out = p->dout_bytes > dout_bytes ? p->dout_bytes : dout_bytes;
    if (p->din_bytes < want_in || p->dout_bytes < want_out) {
        for (int i = 0; i < 3; i++) AETH_HIP(hipStreamSynchronize(p->stream[i]));
        for (int s = 0; s < kPipeSlots; s++) {
            if (p->din[s]) { AETH_HIP(hipFree(p->din[s])); p->din[s] = nullptr; }
            if (p->dout[s]) { AETH_HIP(hipFree(p->dout[s])); p->dout[s] = nullptr; }
        }
        p->din_bytes = want_in; p->dout_bytes = want_out;
    }
    for (int s = 0; s < nslots; s++) {
        if (!p->din[s]) AETH_HIP(hipMalloc((void **)&p->din[s], p->din_bytes));
        if (!p->dout[s]) AETH_HIP(hipMalloc((void **)&p->dout[s], p->dout_bytes));
    }
    return AETH_OK;
}

// `count` pinned staging elements of at least `bytes` each from the context's pool (rebuilt when a run needs larger ones)
int pipe_take_staging(aeth_ctx *ctx, size_t bytes, int count, void **out)
{
    PipeState *p = ctx->pipe;
    if (p->pool && aeth_pool_elem_bytes(p->pool) < bytes) {
        int rc = aeth_pool_destroy(p->pool);        // every element is back between runs
        p->pool = nullptr;
        if (rc) return rc;
    }
    if (!p->pool) {
        int rc = aeth_pool_create(ctx, bytes, 0, 0, &p->pool);
        if (rc) return rc;
    }
    for (int i = 0; i < count; i++) {
        int rc = aeth_pool_take_or_make(p->pool, &out[i]);
        if (rc) { for (int j = 0; j < i; j++) (void)aeth_pool_give_back(p->pool, out[j]); return rc; }
    }
    if (!p->team) {
        // half of the cores this process may run on, 2 .. 12 (tuning: AETH_PIPE_THREADS)
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = c; }
        int n = hw / 2; n = n < 2 ? 2 : (n > 12 ? 12 : n);
        n = aeth::tuning_int("AETH_PIPE_THREADS", n);
        p->team = new (std::nothrow) aeth::CopyTeam(n);
        AETH_REQUIRE(p->team, AETH_E_NOMEM, "out of host memory");
    }
    return AETH_OK;
}

// hist: ntaps-1 host samples in front of `in` (null: zeros), as in aeth_fir_exec_host
// util: per-stage active time as well (the device-side counterpart of the reference's per-stage utilisation
// report, src/pipeline.rs:89-114): timed events around every device stage operation, wall clock around the host ones
int fir_stream_host(aeth_fir *f, const aeth_cf32 *hist, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t chunk,
                    aeth_pipe_stats *stats, aeth_pipe_util *util = nullptr)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    if (util) *util = aeth_pipe_util{0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (stats) *stats = aeth_pipe_stats{0, 0, 0, 0};
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(in && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(in != out, AETH_E_ARG, "FIR cannot run in place (blocks overlap)");
    aeth_ctx *ctx = f->ctx;
    aeth::DeviceGuard g(ctx->device);
    const size_t nh = f->ntaps - 1;
    // chunk: 32 MiB of samples per transfer for long streams, an eighth of a short one (so that its copies still
    // overlap), never under 1 MiB; hop-aligned, so the blocks are those of the one-shot run
    if (chunk == 0) {
        chunk = (size_t)4 << 20;
        if (n / 8 < chunk) chunk = n / 8;
        if (chunk < ((size_t)128 << 10)) chunk = (size_t)128 << 10;
    }
    if (chunk > n) chunk = n;
    chunk = ((chunk + f->hop - 1) / f->hop) * f->hop;
    const size_t nchunks = (n + chunk - 1) / chunk;
    const int nslots = (int)(nchunks < (size_t)kPipeSlots ? nchunks : (size_t)kPipeSlots);
    const size_t in_slot_bytes = (chunk + nh) * sizeof(float2), out_slot_bytes = chunk * sizeof(float2);

    // a side that is already page-locked is copied from / to directly; anything else goes through pinned staging
    const bool stage_in = !aeth::host_range_pinned(in, n * sizeof(float2));
    bool stage_out = !aeth::host_range_pinned(out, n * sizeof(float2));
    // A staged input next to a directly written output is the one combination that measured badly: with 256 Mi
    // samples the downloads into the caller's (pinned) 2 GiB slice took 2.4 x as long while the copy threads fed the
    // uploads (2.3-2.9 GS/s, profiles/r03_stream_host.txt; cause not found), where staging BOTH sides runs at the
    // rate of the all-pinned case (5.6 GS/s) -- so an input that needs the host stage takes the output through it too.
    if (stage_in && !stage_out && aeth::tuning_int("AETH_PIPE_MIXED", 0) == 0) stage_out = true;

    int rc = pipe_prepare(ctx, in_slot_bytes, out_slot_bytes, nslots);
    if (rc) return rc;
    PipeState *ps = ctx->pipe;
    void *pin[2 * kPipeSlots] = {};
    const int npin = (stage_in ? nslots : 0) + (stage_out ? nslots : 0);
    if (npin) { rc = pipe_take_staging(ctx, in_slot_bytes, npin, pin); if (rc) return rc; }
    float2 *pin_in[kPipeSlots] = {}, *pin_out[kPipeSlots] = {};
    { int k = 0; if (stage_in) for (int s = 0; s < nslots; s++) pin_in[s] = (float2 *)pin[k++];
                 if (stage_out) for (int s = 0; s < nslots; s++) pin_out[s] = (float2 *)pin[k++]; }

    hipStream_t s_up = ps->stream[0], s_run = ps->stream[1], s_down = ps->stream[2];
    auto fail = [&](hipError_t e, const char *what) { rc = aeth::hip_fail(e, what); };
    auto ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) { fail(e, what); return false; } return true; };

    (void)ok(hipStreamSynchronize(aeth::ctx_stream(ctx)), "hipStreamSynchronize");
    const double w0 = now_s();
    // [chunk][stage][begin, end] -- only when the utilisation report is wanted
    std::vector<hipEvent_t> marks;
    if (util && rc == AETH_OK) {
        marks.assign(nchunks * 6, nullptr);
        for (auto &m : marks) if (!ok(hipEventCreate(&m), "hipEventCreate")) break;
    }
    auto mark = [&](size_t k, int stage, int end, hipStream_t st) {
        if (!marks.empty() && rc == AETH_OK) (void)ok(hipEventRecord(marks[k * 6 + stage * 2 + end], st), "hipEventRecord");
    };

    std::unique_ptr<std::atomic<int>[]> in_pending(new std::atomic<int>[nchunks]), out_pending(new std::atomic<int>[nchunks]);
    for (size_t k = 0; k < nchunks; k++) { in_pending[k].store(0); out_pending[k].store(0); }
    std::vector<double> t_in(stage_in ? nchunks : 0, 0.0), t_out(stage_out ? nchunks : 0, 0.0);
    double act_in = 0, act_out = 0, last_in = 0, last_out = 0;
    size_t next_in = 0, next_sub = 0, next_out = 0, fin_in = 0, fin_out = 0;
    auto geom = [&](size_t k, size_t &o0, size_t &cnt, size_t &h) {
        o0 = k * chunk; cnt = (n - o0 < chunk) ? n - o0 : chunk; h = (o0 >= nh) ? nh : o0;     // h: history samples the source holds
    };

    int idle = 0;
    while (rc == AETH_OK && (next_sub < nchunks || (stage_out && fin_out < nchunks))) {
        bool progress = false;
        // ---- copy-in: chunk k into its slot's pinned element once the upload of chunk k - 3 has left it
        if (stage_in && next_in < nchunks) {
            const size_t k = next_in; const int s = (int)(k % kPipeSlots);
            bool free_ = k < (size_t)kPipeSlots;
            if (!free_ && next_sub + kPipeSlots > k) {            // chunk k - 3 has been enqueued: its `up` event is the slot's
                const hipError_t q = hipEventQuery(ps->up[s]);
                if (q == hipSuccess) free_ = true; else if (q != hipErrorNotReady) { fail(q, "hipEventQuery"); break; }
            }
            if (free_) {
                size_t o0, cnt, h; geom(k, o0, cnt, h);
                if (k == 0 && hist && nh) memcpy(pin_in[s], hist, nh * sizeof(float2));
                t_in[k] = now_s();
                ps->team->submit(pin_in[s] + (nh - h), in + (o0 - h), (h + cnt) * sizeof(float2), &in_pending[k]);
                next_in++; progress = true;
            }
        }
        // busy time of a host stage = the union of its chunks' [hand-over, completion] intervals (several can be queued)
        while (stage_in && fin_in < next_in && in_pending[fin_in].load(std::memory_order_acquire) == 0) {
            const double t = now_s(), from = t_in[fin_in] > last_in ? t_in[fin_in] : last_in;
            if (t > from) act_in += t - from;
            last_in = t; fin_in++; progress = true;
        }
        // ---- device stages of chunk k: its input is in place and its pinned output element has been emptied
        if (next_sub < nchunks) {
            const size_t k = next_sub; const int s = (int)(k % kPipeSlots);
            const bool in_ready = !stage_in || fin_in > k;
            const bool out_free = !stage_out || k < (size_t)kPipeSlots || fin_out + kPipeSlots > k;
            if (in_ready && out_free) {
                size_t o0, cnt, h; geom(k, o0, cnt, h);
                const bool used = k >= (size_t)kPipeSlots;
                float2 *din = ps->din[s], *dout = ps->dout[s];
                // H2D: the slot's input buffer is free once the kernel of chunk k - 3 has run
                if (used && !ok(hipStreamWaitEvent(s_up, ps->ran[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 0, 0, s_up);
                // [zeros | history | chunk] -> device
                const bool have_hist0 = k == 0 && hist && nh;
                if (stage_in) {
                    // the pinned element mirrors the slot: history (copied in by the host stage) in front of the chunk
                    const size_t skip = have_hist0 ? 0 : nh - h;
                    if (skip && !ok(hipMemsetAsync(din, 0, skip * sizeof(float2), s_up), "hipMemsetAsync")) break;
                    if (!ok(hipMemcpyAsync(din + skip, pin_in[s] + skip, (nh - skip + cnt) * sizeof(float2), hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break;
                } else {
                    if (have_hist0) { if (!ok(hipMemcpyAsync(din, hist, nh * sizeof(float2), hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break; }
                    else if (h < nh && !ok(hipMemsetAsync(din, 0, (nh - h) * sizeof(float2), s_up), "hipMemsetAsync")) break;
                    if (!ok(hipMemcpyAsync(din + (nh - h), in + (o0 - h), (h + cnt) * sizeof(float2), hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break;
                }
                mark(k, 0, 1, s_up);
                if (!ok(hipEventRecord(ps->up[s], s_up), "hipEventRecord")) break;
                // kernel: needs the chunk up and the slot's output buffer drained by the D2H of chunk k - 3
                if (!ok(hipStreamWaitEvent(s_run, ps->up[s], 0), "hipStreamWaitEvent")) break;
                if (used && !ok(hipStreamWaitEvent(s_run, ps->down[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 1, 0, s_run);
                rc = aeth::fir_exec_on(f, s_run, (o0 || hist) ? (const aeth_cf32 *)din : nullptr, (const aeth_cf32 *)(din + nh), cnt, (aeth_cf32 *)dout);
                if (rc) break;
                mark(k, 1, 1, s_run);
                if (!ok(hipEventRecord(ps->ran[s], s_run), "hipEventRecord")) break;
                // D2H
                if (!ok(hipStreamWaitEvent(s_down, ps->ran[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 2, 0, s_down);
                float2 *dst = stage_out ? pin_out[s] : (float2 *)out + o0;
                if (!ok(hipMemcpyAsync(dst, dout, cnt * sizeof(float2), hipMemcpyDeviceToHost, s_down), "hipMemcpyAsync D2H")) break;
                mark(k, 2, 1, s_down);
                if (!ok(hipEventRecord(ps->down[s], s_down), "hipEventRecord")) break;
                next_sub++; progress = true;
            }
        }
        // ---- copy-out: chunk k from its pinned element to the caller's slice once its download has landed
        if (stage_out && next_out < next_sub) {
            const size_t k = next_out; const int s = (int)(k % kPipeSlots);
            const hipError_t q = hipEventQuery(ps->down[s]);
            if (q == hipSuccess) {
                size_t o0, cnt, h; geom(k, o0, cnt, h);
                t_out[k] = now_s();
                ps->team->submit(out + o0, pin_out[s], cnt * sizeof(float2), &out_pending[k]);
                next_out++; progress = true;
            } else if (q != hipErrorNotReady) { fail(q, "hipEventQuery"); break; }
        }
        while (stage_out && fin_out < next_out && out_pending[fin_out].load(std::memory_order_acquire) == 0) {
            const double t = now_s(), from = t_out[fin_out] > last_out ? t_out[fin_out] : last_out;
            if (t > from) act_out += t - from;
            last_out = t; fin_out++; progress = true;
        }
        // nothing moved: a chunk takes hundreds of microseconds, so back off instead of hammering the runtime with
        // event queries (its completion handling shares locks with them)
        if (progress) idle = 0;
        else if (++idle < 32) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    (void)hipGetLastError();                                       // hipErrorNotReady from the queries is not an error
    // drain: the device stages, then whatever the copy threads still hold (also on the error path: they write
    // into the caller's memory and read the pinned elements that go back to the pool below)
    { hipError_t e = hipStreamSynchronize(s_up); if (e != hipSuccess && rc == AETH_OK) fail(e, "hipStreamSynchronize"); }
    { hipError_t e = hipStreamSynchronize(s_run); if (e != hipSuccess && rc == AETH_OK) fail(e, "hipStreamSynchronize"); }
    { hipError_t e = hipStreamSynchronize(s_down); if (e != hipSuccess && rc == AETH_OK) fail(e, "hipStreamSynchronize"); }
    for (size_t k = 0; k < next_in; k++) while (in_pending[k].load(std::memory_order_acquire) != 0) std::this_thread::yield();
    for (size_t k = 0; k < next_out; k++) while (out_pending[k].load(std::memory_order_acquire) != 0) std::this_thread::yield();
    const double w1 = now_s();

    if (rc == AETH_OK && (stats || util)) {
        // wall time of the whole run (the host stages end after the last device event); device stage times from events
        const double secs = w1 - w0;
        const double pinned = (stage_in ? 0 : 1) + (stage_out ? 0 : 2);
        if (stats) { stats->seconds = secs; stats->samples = (double)n; stats->chunks = (double)nchunks; stats->pinned = pinned; }
        if (util) {
            util->seconds = secs; util->samples = (double)n; util->chunks = (double)nchunks; util->pinned = pinned;
            double act[3] = {0, 0, 0};
            for (size_t k = 0; k < nchunks && !marks.empty(); k++)
                for (int st = 0; st < 3; st++) {
                    float d = 0;
                    if (hipEventElapsedTime(&d, marks[k * 6 + st * 2], marks[k * 6 + st * 2 + 1]) == hipSuccess) act[st] += d * 1e-3;
                }
            (void)hipGetLastError();
            util->active_upload = act[0]; util->active_kernel = act[1]; util->active_download = act[2];
            util->active_copy_in = act_in; util->active_copy_out = act_out;
        }
    }
    for (auto m : marks) if (m) (void)hipEventDestroy(m);
    for (int i = 0; i < npin; i++) {
        const int r = aeth_pool_give_back(ps->pool, pin[i]);
        if (r && rc == AETH_OK) rc = r;
    }
    return rc;
}

}  // namespace

extern "C" {

int aeth_fir_stream_host(aeth_fir *f, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t chunk, aeth_pipe_stats *stats)
{
    return fir_stream_host(f, nullptr, in, n, out, chunk, stats);
}

int aeth_fir_stream_host_util(aeth_fir *f, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t chunk, aeth_pipe_util *util)
{
    AETH_REQUIRE(util, AETH_E_ARG, "util is null");
    return fir_stream_host(f, nullptr, in, n, out, chunk, nullptr, util);
}

}  // extern "C"
