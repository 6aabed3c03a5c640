// aeth_pipeline.hip -- host-resident stream through the device at link rate (SURVEY 8f "next" #4).
//
// The GPU-side counterpart of the reference's thread-per-stage pipeline over pooled buffers
// (src/pipeline.rs:52-137, src/pool.rs:43-221, examples/pipeline.rs:36-52).  Five stages, chunk by chunk:
//
//     copy-in        caller slice  -> pinned pool element          host threads (CopyTeam)
//     upload         pinned        -> device slot                  HIP stream 0 (H2D copy engine)
//     compute        one of the library's device ops on the slot   HIP stream 1
//     download       device slot   -> pinned pool element          HIP stream 2 (D2H copy engine)
//     copy-out       pinned        -> caller slice                 host threads
//
// The reference's pipeline takes an arbitrary closure per stage (src/pipeline.rs:24-41 `add_stage<F: FnMut(O) -> U>`,
// :123-137 `new`); a closure cannot cross the C ABI, so the compute stage is an op descriptor (aeth_stream_op) or a
// chain of up to eight of them (`add_stage` ... `add_stage`: intermediates stay on the device): the fused FIR (plain or
// with its decimating store), batched FFT frames, the correlator chain, correlate + demod (8 B in, 1-2 B out per sample),
// FFT + interpolate (1 sample in, n_between + 1 out), modulate + AWGN (bit bytes in, symbols out) -- input and output
// chunks differ in element size and count per op.
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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <vector>

namespace aeth {

static std::atomic<int> g_fail_after{0};
void pipe_fail_arm(int n) { g_fail_after.store(n); }
// true exactly once: when the armed count reaches zero
bool pipe_fail_after_take()
{
    int v = g_fail_after.load();
    while (v > 0) { if (g_fail_after.compare_exchange_weak(v, v - 1)) return v == 1; }
    return false;
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
        for (int k = 0; k < 2; k++) if (p->mid[s][k]) (void)hipFree(p->mid[s][k]);
        if (p->up[s]) (void)hipEventDestroy(p->up[s]);
        if (p->ran[s]) (void)hipEventDestroy(p->ran[s]);
        if (p->down[s]) (void)hipEventDestroy(p->down[s]);
    }
    (void)pool_destroy_forced(p->pool[0]);
    (void)pool_destroy_forced(p->pool[1]);
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

// What the compute stage does with one chunk.  Elements are the op's own (cf32 samples in; samples or bit bytes out).
struct StageOp {
    aeth_ctx *ctx = nullptr;
    size_t in_unit = 8, out_unit = 8;      // bytes per input / output element
    size_t hist = 0;                       // input elements of left context in front of every chunk (FIR: ntaps - 1)
    size_t align = 1;                      // chunk granularity in input elements (FIR: hop; frame ops: the frame length)
    size_t out_per_align = 1;              // output elements per `align` input elements (out_count is linear in the input)
    bool inplace = false;                  // the op leaves its result in the input slot (aeth_fft_mul_ifft)
    size_t mid_per_align = 0;              // bytes of device scratch per `align` input elements, twice (a chain of ops: the
                                           // intermediates ping-pong between two scratch buffers of the slot)
    // din_chunk: first input element of the chunk in the device slot (hist elements of context in front of it when
    // have_hist); runs on `s` and must not wait for the host; mid0 / mid1: the slot's scratch pair (chains only)
    // first: index in the stream of the chunk's first input element (position-addressed ops: the noise generator)
    std::function<int(hipStream_t s, const void *din_hist, void *din_chunk, bool have_hist, size_t cnt, void *dout, size_t cnt_out,
                      void *mid0, void *mid1, size_t first)> run;
    size_t out_count(size_t cnt) const { return cnt / align * out_per_align + (cnt % align) * out_per_align / align; }
};

// While an op of the public API runs for the pipeline, the context's stream IS the compute stage's stream: every
// entry point launches on aeth::ctx_stream(ctx).  A context is used by one thread at a time (aether_hip.h), and the
// pipeline has drained the context's own stream before its first chunk, so nothing else can be enqueued meanwhile.
struct StreamSwap {
    aeth_ctx *c; hipStream_t saved; bool overlap;
    StreamSwap(aeth_ctx *ctx, hipStream_t s) : c(ctx), saved(ctx->stream_main), overlap(ctx->overlap) { c->stream_main = s; c->overlap = false; c->chain_last = -1; }
    ~StreamSwap() { c->stream_main = saved; c->overlap = overlap; c->chain_last = -1; }
};

// streams, events, device slots of `din_bytes` / `dout_bytes`, grown (never shrunk; aeth_ctx_trim releases) between calls
int pipe_prepare(aeth_ctx *ctx, size_t din_bytes, size_t dout_bytes, int nslots, size_t mid_bytes = 0)
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
    if (mid_bytes > p->mid_bytes) {                  // scratch pair per slot of a chain of ops
        for (int i = 0; i < 3; i++) AETH_HIP(hipStreamSynchronize(p->stream[i]));
        for (int s = 0; s < kPipeSlots; s++)
            for (int k = 0; k < 2; k++) if (p->mid[s][k]) { AETH_HIP(hipFree(p->mid[s][k])); p->mid[s][k] = nullptr; }
        p->mid_bytes = mid_bytes;
    }
    if (mid_bytes)
        for (int s = 0; s < nslots; s++)
            for (int k = 0; k < 2; k++) if (!p->mid[s][k]) AETH_HIP(hipMalloc((void **)&p->mid[s][k], p->mid_bytes));
    if (!p->team) {
        // three eighths of the cores this process may USE, 2 .. 12 (tuning: AETH_PIPE_THREADS).  Created BEFORE any staging
        // element is taken, so that no error path between the two leaves elements checked out.
        int hw = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = c; }
        // a container's CPU quota counts, not the cores it can see: a one-GPU box shows 256 hardware threads and grants
        // 16 -- twelve copy threads there get throttled and run at HALF the rate of six (profiles/r04_pipe_mixed_lab.txt:
        // 68 ms against 48 ms for 256 Mi samples)
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long q = 0, per = 0; char buf[64] = {0};
            if (fscanf(f, "%63s %lld", buf, &per) == 2 && strcmp(buf, "max") != 0 && per > 0 && (q = atoll(buf)) > 0) {
                const int cpus = (int)((q + per - 1) / per);
                if (cpus >= 1 && cpus < hw) hw = cpus;
            }
            fclose(f);
        }
        int n = hw * 3 / 8; n = n < 2 ? 2 : (n > 12 ? 12 : n);
        n = aeth::tuning_int("AETH_PIPE_THREADS", n);
        p->team = new (std::nothrow) aeth::CopyTeam(n);
        AETH_REQUIRE(p->team, AETH_E_NOMEM, "out of host memory");
    }
    return AETH_OK;
}

// `count` pinned staging elements of at least `bytes` each from the context's pool for side `side` (0 in, 1 out; the
// pool is rebuilt when a run needs larger elements).  On ANY failure nothing stays checked out and the pool a refused
// aeth_pool_destroy leaves behind stays where it is (still owned, still usable for smaller runs).
int pipe_take_staging(aeth_ctx *ctx, int side, size_t bytes, int count, void **out)
{
    PipeState *p = ctx->pipe;
    if (p->pool[side] && aeth_pool_elem_bytes(p->pool[side]) < bytes) {
        int rc = aeth_pool_destroy(p->pool[side]);  // every element is back between runs
        if (rc) return rc;                          // refused: the pool stays
        p->pool[side] = nullptr;
    }
    if (!p->pool[side]) {
        int rc = aeth_pool_create(ctx, bytes, 0, 0, &p->pool[side]);
        if (rc) return rc;
    }
    for (int i = 0; i < count; i++) {
        int rc = aeth_pool_take_or_make(p->pool[side], &out[i]);
        if (rc == AETH_OK && aeth::pipe_fail_after_take() ) rc = aeth::set_error(AETH_E_NOMEM, "staging: forced failure (test hook)");
        if (rc) {
            for (int j = 0; j <= i; j++) if (out[j]) { (void)aeth_pool_give_back(p->pool[side], out[j]); out[j] = nullptr; }
            return rc;
        }
    }
    return AETH_OK;
}

// The largest slot (one side of one chunk) the pipeline sizes by itself or accepts from the caller: a larger
// chunk_samples is split internally.  Keeps what a context retains between runs bounded: 3 device slots per side and,
// for pageable caller memory, 3 pinned elements per side -- at most 6 x 64 MiB each of device and of pinned memory.
constexpr size_t kMaxSlotBytes = (size_t)64 << 20;

// hist: op.hist host elements in front of `in` (null: zeros), as in aeth_fir_exec_host
// util: per-stage active time as well (the device-side counterpart of the reference's per-stage utilisation
// report, src/pipeline.rs:89-114): timed events around every device stage operation, wall clock around the host ones
int stream_host(const StageOp &op, const void *hist, const void *in_, size_t n, void *out_, size_t n_out, size_t chunk,
                aeth_pipe_stats *stats, aeth_pipe_util *util = nullptr)
{
    if (util) *util = aeth_pipe_util{0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (stats) *stats = aeth_pipe_stats{0, 0, 0, 0};
    AETH_REQUIRE(n_out == op.out_count(n), AETH_E_LEN, "output holds %zu elements, the op produces %zu from %zu", n_out, op.out_count(n), n);
    if (n == 0) return AETH_OK;
    AETH_REQUIRE(in_ && out_, AETH_E_ARG, "null pointer");
    const char *in = (const char *)in_;
    char *out = (char *)out_;
    const size_t IU = op.in_unit, OU = op.out_unit, nh = op.hist;
    {   // the stream is read while its head is already being written: the two host ranges must not touch
        const uintptr_t a0 = (uintptr_t)in, a1 = a0 + n * IU, b0 = (uintptr_t)out, b1 = b0 + n_out * OU;
        AETH_REQUIRE(a1 <= b0 || b1 <= a0, AETH_E_ARG, "input and output ranges overlap");
    }
    aeth_ctx *ctx = op.ctx;
    aeth::DeviceGuard g(ctx->device);
    // chunk: 32 MiB on the larger side per transfer for long streams, an eighth of a short one (so that its copies
    // still overlap), never under 1 MiB; a whole number of `align` units, so every chunk runs the blocks / frames the
    // one-shot call would
    const size_t per_in = IU * op.align, per_out = OU * op.out_per_align;             // bytes per align unit
    const size_t per_max = per_in > per_out ? per_in : per_out;
    auto units_for = [&](size_t bytes) { size_t u = bytes / per_max; return u < 1 ? (size_t)1 : u; };
    const size_t n_units = (n + op.align - 1) / op.align;
    size_t cu;                                                                        // chunk in align units
    if (chunk == 0) {
        cu = units_for((size_t)32 << 20);
        if (n_units / 8 < cu) cu = n_units / 8;
        const size_t lo = units_for((size_t)1 << 20);
        if (cu < lo) cu = lo;
    } else cu = (chunk + op.align - 1) / op.align;
    if (cu > units_for(kMaxSlotBytes)) cu = units_for(kMaxSlotBytes);
    if (cu > n_units) cu = n_units;
    chunk = cu * op.align;
    const size_t chunk_out = op.out_count(chunk);
    const size_t nchunks = (n + chunk - 1) / chunk;
    const int nslots = (int)(nchunks < (size_t)kPipeSlots ? nchunks : (size_t)kPipeSlots);
    const size_t in_slot_bytes = (chunk + nh) * IU, out_slot_bytes = chunk_out * OU;

    // a side that is already page-locked is copied from / to directly; anything else goes through pinned staging
    const bool stage_in = !aeth::host_range_pinned(in, n * IU);
    bool stage_out = !aeth::host_range_pinned(out, n_out * OU);
    // A staged input next to a directly written output is the one combination that measured badly: with 256 Mi
    // samples the downloads into the caller's (pinned) 2 GiB slice took 2.4 x as long while the copy threads fed the
    // uploads (2.3-2.9 GS/s, profiles/r03_stream_host.txt), where staging BOTH sides runs at the rate of the
    // all-pinned case (5.6 GS/s) -- so an input that needs the host stage takes the output through it too
    // (AETH_PIPE_MIXED=1 under AETH_TUNING=1 keeps the direct download; include/aether_hip.h, "Tuning knobs").
    if (stage_in && !stage_out && aeth::tuning_int("AETH_PIPE_MIXED", 0) == 0) stage_out = true;

    int rc = pipe_prepare(ctx, in_slot_bytes, op.inplace ? 16 : out_slot_bytes, nslots, op.mid_per_align * cu);
    if (rc) return rc;
    PipeState *ps = ctx->pipe;
    void *pin_in[kPipeSlots] = {}, *pin_out[kPipeSlots] = {};
    if (stage_in) { rc = pipe_take_staging(ctx, 0, in_slot_bytes, nslots, pin_in); if (rc) return rc; }
    if (stage_out) {
        rc = pipe_take_staging(ctx, 1, out_slot_bytes, nslots, pin_out);
        if (rc) { if (stage_in) for (int s = 0; s < nslots; s++) (void)aeth_pool_give_back(ps->pool[0], pin_in[s]); return rc; }
    }

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
        o0 = k * chunk; cnt = (n - o0 < chunk) ? n - o0 : chunk; h = (o0 >= nh) ? nh : o0;     // h: history elements the source holds
    };

    int idle = 0;
    {
    StreamSwap swap(ctx, s_run);
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
                char *pe = (char *)pin_in[s];
                if (k == 0 && hist && nh) memcpy(pe, hist, nh * IU);
                t_in[k] = now_s();
                ps->team->submit(pe + (nh - h) * IU, in + (o0 - h) * IU, (h + cnt) * IU, &in_pending[k]);
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
                const size_t cnt_out = op.out_count(cnt), oo0 = op.out_count(o0);
                const bool used = k >= (size_t)kPipeSlots;
                char *din = (char *)ps->din[s], *dout = (char *)ps->dout[s];
                // H2D: the slot's input buffer is free once the op of chunk k - 3 has run (an in-place op: once its
                // result has been downloaded)
                if (used && !ok(hipStreamWaitEvent(s_up, op.inplace ? ps->down[s] : ps->ran[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 0, 0, s_up);
                // [zeros | history | chunk] -> device
                const bool have_hist0 = k == 0 && hist && nh;
                if (stage_in) {
                    // the pinned element mirrors the slot: history (copied in by the host stage) in front of the chunk
                    const size_t skip = have_hist0 ? 0 : nh - h;
                    if (skip && !ok(hipMemsetAsync(din, 0, skip * IU, s_up), "hipMemsetAsync")) break;
                    if (!ok(hipMemcpyAsync(din + skip * IU, (char *)pin_in[s] + skip * IU, (nh - skip + cnt) * IU, hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break;
                } else {
                    if (have_hist0) { if (!ok(hipMemcpyAsync(din, hist, nh * IU, hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break; }
                    else if (h < nh && !ok(hipMemsetAsync(din, 0, (nh - h) * IU, s_up), "hipMemsetAsync")) break;
                    if (!ok(hipMemcpyAsync(din + (nh - h) * IU, in + (o0 - h) * IU, (h + cnt) * IU, hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break;
                }
                mark(k, 0, 1, s_up);
                if (!ok(hipEventRecord(ps->up[s], s_up), "hipEventRecord")) break;
                // compute: needs the chunk up and the slot's output buffer drained by the D2H of chunk k - 3
                if (!ok(hipStreamWaitEvent(s_run, ps->up[s], 0), "hipStreamWaitEvent")) break;
                if (used && !ok(hipStreamWaitEvent(s_run, ps->down[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 1, 0, s_run);
                rc = op.run(s_run, din, din + nh * IU, (o0 || hist) && nh, cnt, dout, cnt_out, ps->mid[s][0], ps->mid[s][1], o0);
                if (rc) break;
                mark(k, 1, 1, s_run);
                if (!ok(hipEventRecord(ps->ran[s], s_run), "hipEventRecord")) break;
                // D2H
                if (!ok(hipStreamWaitEvent(s_down, ps->ran[s], 0), "hipStreamWaitEvent")) break;
                mark(k, 2, 0, s_down);
                char *dst = stage_out ? (char *)pin_out[s] : out + oo0 * OU;
                if (!ok(hipMemcpyAsync(dst, op.inplace ? din + nh * IU : dout, cnt_out * OU, hipMemcpyDeviceToHost, s_down), "hipMemcpyAsync D2H")) break;
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
                ps->team->submit(out + op.out_count(o0) * OU, pin_out[s], op.out_count(cnt) * OU, &out_pending[k]);
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
    for (int s = 0; s < nslots; s++) {
        if (pin_in[s]) { const int r = aeth_pool_give_back(ps->pool[0], pin_in[s]); if (r && rc == AETH_OK) rc = r; }
        if (pin_out[s]) { const int r = aeth_pool_give_back(ps->pool[1], pin_out[s]); if (r && rc == AETH_OK) rc = r; }
    }
    return rc;
}

// ---- the ops -------------------------------------------------------------------------------------------------
int op_fir(aeth_fir *f, StageOp &op)
{
    AETH_REQUIRE(f, AETH_E_ARG, "fir is null");
    op.ctx = f->ctx; op.hist = f->ntaps - 1; op.align = f->hop; op.out_per_align = f->hop;
    op.run = [f](hipStream_t s, const void *dh, void *dc, bool have_hist, size_t cnt, void *dout, size_t, void *, void *, size_t) {
        return aeth::fir_exec_on(f, s, have_hist ? (const aeth_cf32 *)dh : nullptr, (const aeth_cf32 *)dc, cnt, (aeth_cf32 *)dout);
    };
    return AETH_OK;
}

int make_op(aeth_ctx *ctx, const aeth_stream_op *d, StageOp &op)
{
    AETH_REQUIRE(ctx && d, AETH_E_ARG, "null argument");
    if (d->kind == AETH_STREAM_FIR || d->kind == AETH_STREAM_FIR_DECIM) {
        int rc = op_fir(d->fir, op); if (rc) return rc;
        AETH_REQUIRE(op.ctx == ctx, AETH_E_ARG, "the filter belongs to another context");
        if (d->kind == AETH_STREAM_FIR_DECIM) {
            // the filter followed by sampling::downsample (sampling.rs:28-42) in the kernel's store: out[i] = y[i * dec].
            // Chunks are whole hops, so with dec | hop every chunk starts on a kept sample and keeps hop / dec per hop.
            aeth_fir *f = d->fir;
            const size_t dec = d->n_between;
            AETH_REQUIRE(dec >= 1 && f->hop % dec == 0, AETH_E_ARG, "decimation %zu must divide the filter's hop (%zu)", dec, f->hop);
            op.out_per_align = f->hop / dec;
            op.run = [f, dec](hipStream_t, const void *dh, void *dc, bool have_hist, size_t cnt, void *dout, size_t cnt_out, void *, void *, size_t) {
                if (cnt % dec != 0) return aeth::set_error(AETH_E_ARG, AETH_MSG_DECIM);          /* sampling.rs:32-36 */
                return aeth_fir_exec_decim(f, have_hist ? (const aeth_cf32 *)dh : nullptr, (const aeth_cf32 *)dc, cnt, (aeth_cf32 *)dout, cnt_out);
            };
        }
        return AETH_OK;
    }
    if (d->kind == AETH_STREAM_MODULATE_AWGN) {
        // Modulation::modulate, then Awgn::apply on the fresh symbols (examples/modem.rs:19-26): bit bytes in, symbols out;
        // the noise is addressed by position in the stream, so chunks draw what the one-shot call would
        const aeth_stream_op c = *d;
        AETH_REQUIRE(c.bits_per_symbol == 1 || c.bits_per_symbol == 2, AETH_E_UNSUPPORTED, "bits_per_symbol %d: BPSK (1) or QPSK (2)", c.bits_per_symbol);
        const size_t bps = (size_t)c.bits_per_symbol;
        op.ctx = ctx; op.in_unit = 1; op.out_unit = sizeof(aeth_cf32); op.align = 2 * bps; op.out_per_align = 2;   // whole pairs of symbols: the generator draws per pair
        op.run = [ctx, c, bps](hipStream_t, const void *, void *dc, bool, size_t cnt, void *dout, size_t cnt_out, void *, void *, size_t first) {
            return aeth_modulate_awgn(ctx, (const uint8_t *)dc, cnt, c.bits_per_symbol, c.table_host, (aeth_cf32 *)dout, cnt_out, c.x_fwd, c.seed,
                                      c.offset + first / bps);
        };
        return AETH_OK;
    }
    aeth_fft *p = d->fft;
    AETH_REQUIRE(p, AETH_E_ARG, "plan is null");
    AETH_REQUIRE(p->ctx == ctx, AETH_E_ARG, "the plan belongs to another context");
    const size_t N = p->len;
    op.ctx = ctx; op.align = N; op.out_per_align = N;
    const aeth_stream_op c = *d;                     // by value: the descriptor need not outlive the call, but does anyway
    switch (d->kind) {
    case AETH_STREAM_FFT:                            // Fft::fwd / bwd over chunks_mut(fft_len) (src/util/plot.rs:59-61)
        op.run = [p, c, N](hipStream_t, const void *, void *dc, bool, size_t cnt, void *dout, size_t, void *, void *, size_t) {
            return aeth_fft_exec(p, (const aeth_cf32 *)dc, cnt, (aeth_cf32 *)dout, cnt / N, c.sign, c.scale_kind_fwd, c.x_fwd);
        };
        return AETH_OK;
    case AETH_STREAM_FFT_MUL_IFFT:                   // benches/benches.rs:410-416, in place
        op.inplace = true;
        op.run = [p, c, N](hipStream_t, const void *, void *dc, bool, size_t cnt, void *, size_t, void *, void *, size_t) {
            return aeth_fft_mul_ifft(p, (aeth_cf32 *)dc, cnt, cnt / N, c.sig_dev, c.n_sig, c.scale_kind_fwd, c.x_fwd, c.scale_kind_bwd, c.x_bwd);
        };
        return AETH_OK;
    case AETH_STREAM_FFT_MUL_IFFT_DEMOD:             // ... then Modulation::demod_naive (examples/modem.rs:28-31)
        AETH_REQUIRE(c.bits_per_symbol == 1 || c.bits_per_symbol == 2, AETH_E_UNSUPPORTED, "bits_per_symbol %d: BPSK (1) or QPSK (2)", c.bits_per_symbol);
        op.out_unit = 1; op.out_per_align = N * (size_t)c.bits_per_symbol;
        op.run = [p, c, N](hipStream_t, const void *, void *dc, bool, size_t cnt, void *dout, size_t cnt_out, void *, void *, size_t) {
            return aeth_fft_mul_ifft_demod(p, (const aeth_cf32 *)dc, cnt, cnt / N, c.sig_dev, c.n_sig, c.scale_kind_fwd, c.x_fwd,
                                           c.scale_kind_bwd, c.x_bwd, c.bits_per_symbol, c.table_host, (uint8_t *)dout, cnt_out, c.compat);
        };
        return AETH_OK;
    case AETH_STREAM_FFT_INTERPOLATE:                // the transform, then sampling::interpolate per frame (BASELINE config 5)
        op.out_per_align = N + (N - 1) * c.n_between;
        op.run = [p, c, N](hipStream_t, const void *, void *dc, bool, size_t cnt, void *dout, size_t cnt_out, void *, void *, size_t) {
            size_t wrote = 0;
            int rc = aeth_fft_exec_interpolate(p, (const aeth_cf32 *)dc, cnt, cnt / N, c.sign, c.scale_kind_fwd, c.x_fwd, (aeth_cf32 *)dout,
                                               cnt_out, c.n_between, c.compat, &wrote);
            if (rc == AETH_OK && wrote != cnt_out) rc = aeth::set_error(AETH_E_LEN, "interpolate wrote %zu of %zu", wrote, cnt_out);
            return rc;
        };
        return AETH_OK;
    default:
        return aeth::set_error(AETH_E_ARG, "unknown stream op %d", d->kind);
    }
}

// ---- a chain of ops as ONE compute stage: pipeline::new(..).add_stage(a).add_stage(b) (src/pipeline.rs:24-41) ------------
// Stage i's output is stage i + 1's input on the device; only the first stage sees host data and only the last one's
// output goes back.  The chunk granule is the smallest number of input elements that every stage takes in whole
// hops / frames; intermediates ping-pong between the slot's two scratch buffers (an in-place stage leaves its result
// where its input was).
size_t gcd_sz(size_t a, size_t b) { while (b) { const size_t t = a % b; a = b; b = t; } return a; }

int make_chain(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, StageOp &out)
{
    AETH_REQUIRE(ops && n_ops >= 1 && n_ops <= 8, AETH_E_ARG, "a pipeline has 1 .. 8 compute ops");
    if (n_ops == 1) return make_op(ctx, &ops[0], out);
    auto st = std::make_shared<std::vector<StageOp>>(n_ops);
    for (size_t i = 0; i < n_ops; i++) {
        int rc = make_op(ctx, &ops[i], (*st)[i]); if (rc) return rc;
        AETH_REQUIRE(i == 0 || (*st)[i].hist == 0, AETH_E_UNSUPPORTED, "stage %zu: a filter (left context) can only be the first stage", i);
        AETH_REQUIRE(i + 1 == n_ops || (*st)[i].out_unit == sizeof(aeth_cf32), AETH_E_ARG, "stage %zu emits bits: it has to be the last stage", i);
        AETH_REQUIRE(i == 0 || (*st)[i].in_unit == sizeof(aeth_cf32), AETH_E_ARG, "stage %zu takes bits: it has to be the first stage", i);
    }
    // granule U: every stage's input count is a multiple of its own granule
    size_t U = (*st)[0].align;
    for (int guard = 0; guard < 64; guard++) {
        size_t c = U; bool ok = true;
        for (size_t i = 0; i < n_ops; i++) {
            const size_t g = (*st)[i].align;
            if (c % g) { U *= g / gcd_sz(c, g); ok = false; break; }
            c = (*st)[i].out_count(c);
        }
        if (ok) break;
        AETH_REQUIRE(guard < 63 && U < ((size_t)1 << 40), AETH_E_UNSUPPORTED, "the stages' chunk granules do not fit together");
    }
    out = StageOp();
    out.ctx = ctx; out.hist = (*st)[0].hist; out.align = U; out.in_unit = (*st)[0].in_unit; out.out_unit = (*st)[n_ops - 1].out_unit;
    size_t c = U, mid = 0; bool all_inplace = true;
    for (size_t i = 0; i < n_ops; i++) {
        const size_t co = (*st)[i].out_count(c);
        all_inplace = all_inplace && (*st)[i].inplace;
        if (!(*st)[i].inplace && co * sizeof(aeth_cf32) > mid) mid = co * sizeof(aeth_cf32);
        c = co;
    }
    out.out_per_align = c; out.inplace = all_inplace; out.mid_per_align = all_inplace ? 0 : mid;
    out.run = [st, n_ops](hipStream_t s, const void *dh, void *dc, bool have_hist, size_t cnt, void *dout, size_t cnt_out, void *m0, void *m1, size_t first) -> int {
        void *mids[2] = {m0, m1};
        int which = 0;
        const void *cur_hist = dh; void *cur = dc; size_t c = cnt;
        for (size_t i = 0; i < n_ops; i++) {
            const StageOp &o = (*st)[i];
            const size_t co = o.out_count(c);
            const bool last = i + 1 == n_ops;
            if (o.inplace) {
                int rc = o.run(s, cur_hist, cur, have_hist, c, nullptr, co, nullptr, nullptr, first); if (rc) return rc;
            } else {
                void *dst = last ? dout : mids[which]; which ^= 1;
                int rc = o.run(s, cur_hist, cur, have_hist, c, dst, co, nullptr, nullptr, first); if (rc) return rc;
                cur = dst;
            }
            have_hist = false; cur_hist = cur; first = o.out_count(first); c = co;
        }
        if (c != cnt_out) return aeth::set_error(AETH_E_LEN, "the chain produced %zu of %zu elements", c, cnt_out);
        // an in-place last stage behind an out-of-place one leaves the result in scratch: hand it to the download
        if (cur != dout && dout && cur != dc) {
            const hipError_t e = hipMemcpyAsync(dout, cur, c * (*st)[n_ops - 1].out_unit, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return aeth::hip_fail(e, "hipMemcpyAsync D2D");
        }
        return AETH_OK;
    };
    return AETH_OK;
}

int stream_chain(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, const void *in, size_t n_in, void *out, size_t n_out,
                 size_t chunk, aeth_pipe_stats *stats, aeth_pipe_util *util)
{
    StageOp op;
    int rc = make_chain(ctx, ops, n_ops, op); if (rc) return rc;
    AETH_REQUIRE(n_in % op.align == 0, AETH_E_LEN, "a chain of stages takes whole granules of %zu samples (" AETH_MSG_FFT_LEN ")", op.align);
    return stream_host(op, nullptr, in, n_in, out, n_out, chunk, stats, util);
}

int stream_any(aeth_ctx *ctx, const aeth_stream_op *d, const void *in, size_t n_in, void *out, size_t n_out, size_t chunk,
               aeth_pipe_stats *stats, aeth_pipe_util *util)
{
    StageOp op;
    int rc = make_op(ctx, d, op); if (rc) return rc;
    if (d->kind == AETH_STREAM_FIR_DECIM)
        AETH_REQUIRE(n_in % d->n_between == 0, AETH_E_ARG, AETH_MSG_DECIM);           /* sampling.rs:32-36 */
    else if (d->kind == AETH_STREAM_MODULATE_AWGN)
        AETH_REQUIRE(n_in % (size_t)d->bits_per_symbol == 0, AETH_E_LEN, "bit count %zu is not a multiple of BITS_PER_SYMBOL %d", n_in, d->bits_per_symbol);
    else if (d->kind != AETH_STREAM_FIR)
        AETH_REQUIRE(n_in % op.align == 0, AETH_E_LEN, AETH_MSG_FFT_LEN);             /* fft.rs:163-167: whole frames only */
    return stream_host(op, nullptr, in, n_in, out, n_out, chunk, stats, util);
}

}  // namespace

extern "C" {

int aeth_stream_host(aeth_ctx *ctx, const aeth_stream_op *op, const void *in, size_t n_in, void *out, size_t n_out,
                     size_t chunk, aeth_pipe_stats *stats)
{
    return stream_any(ctx, op, in, n_in, out, n_out, chunk, stats, nullptr);
}

int aeth_stream_host_util(aeth_ctx *ctx, const aeth_stream_op *op, const void *in, size_t n_in, void *out, size_t n_out,
                          size_t chunk, aeth_pipe_util *util)
{
    AETH_REQUIRE(util, AETH_E_ARG, "util is null");
    return stream_any(ctx, op, in, n_in, out, n_out, chunk, nullptr, util);
}

/* pipeline::new(..).add_stage(a).add_stage(b)...: several ops as one compute stage (see make_chain) */
int aeth_stream_host_chain(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, const void *in, size_t n_in, void *out, size_t n_out,
                           size_t chunk, aeth_pipe_stats *stats, aeth_pipe_util *util)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (n_ops == 1 && ops) return stream_any(ctx, ops, in, n_in, out, n_out, chunk, stats, util);
    return stream_chain(ctx, ops, n_ops, in, n_in, out, n_out, chunk, stats, util);
}

size_t aeth_stream_chain_out_count(aeth_ctx *ctx, const aeth_stream_op *ops, size_t n_ops, size_t n_in)
{
    StageOp so;
    if (!ctx || make_chain(ctx, ops, n_ops, so) != AETH_OK) return 0;
    return so.out_count(n_in);
}

size_t aeth_stream_out_count(aeth_ctx *ctx, const aeth_stream_op *op, size_t n_in)
{
    StageOp so;
    if (make_op(ctx, op, so) != AETH_OK) return 0;
    return so.out_count(n_in);
}

int aeth_fir_stream_host(aeth_fir *f, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t chunk, aeth_pipe_stats *stats)
{
    StageOp op;
    int rc = op_fir(f, op); if (rc) return rc;
    return stream_host(op, nullptr, in, n, out, n, chunk, stats);
}

int aeth_fir_stream_host_util(aeth_fir *f, const aeth_cf32 *in, size_t n, aeth_cf32 *out, size_t chunk, aeth_pipe_util *util)
{
    AETH_REQUIRE(util, AETH_E_ARG, "util is null");
    StageOp op;
    int rc = op_fir(f, op); if (rc) return rc;
    return stream_host(op, nullptr, in, n, out, n, chunk, nullptr, util);
}

/* Gives back what a context retains between calls: the host pipeline's stage streams, device slots, pinned staging
 * pools and copy threads, and the device scratch of the host-slice flavours.  The next call that needs them creates
 * them again. */
int aeth_ctx_trim(aeth_ctx *ctx)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    aeth::DeviceGuard g(ctx->device);
    AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(ctx)));
    aeth::fft_cache_release(ctx);
    aeth::pipe_release(ctx);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i]) { AETH_HIP(hipFree(ctx->stage[i])); ctx->stage[i] = nullptr; ctx->stage_bytes[i] = 0; }
        if (ctx->bounce[i]) { AETH_HIP(hipHostFree(ctx->bounce[i])); ctx->bounce[i] = nullptr; }
    }
    return AETH_OK;
}

/* test hook (tests/test_gpu_pool.py): the n-th staging element taken from now on fails after it has been taken */
void aeth_test_fail_staging_after(int n) { aeth::pipe_fail_arm(n); }

}  // extern "C"
