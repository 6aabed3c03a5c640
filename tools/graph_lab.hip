// graph_lab.hip -- launch-bound call sequences as HIP graphs: what would capture-and-replay buy over the C ABI's
// plain calls?  C++ (no interpreter in the loop).  The library's device-flavour calls are captured from the context's own
// stream (hipStreamBeginCapture on aeth_ctx_stream, which parks the overlap lane) and replayed with hipGraphLaunch;
// beside each, the same calls issued directly.  Per sequence: microseconds per replay over back-to-back replays with one
// sync at the end (throughput) and with a sync after every replay (latency).
//   C1    add -> mul -> conj on 4096 samples (three launches; BASELINE config 1), and as one fused chain
//   C2    fft-2048 fwd (copy) + ifwd (in place) on 1 Mi samples (two launches; BASELINE config 2)
//   C4/8  what one GPU of eight runs per C4 step: (modulate_awgn + correlate_demod) x 2 on 2048 frames (four launches)
//   x16   sixteen single-frame fft-2048 calls on sixteen buffers (the reference's per-frame loop, util/plot.rs:59-61)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/graph_lab.hip -o tools/bin/graph_lab \
//         -Laether_primitives_amd/lib -laether_hip -Wl,-rpath,'$ORIGIN/../../aether_primitives_amd/lib'
#include <hip/hip_runtime.h>
#include "aether_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define AK(x) do { int r_ = (x); if (r_) { printf("aeth error %d (%s) line %d\n", r_, aeth_last_error(), __LINE__); exit(1); } } while (0)

static aeth_ctx *ctx;
static hipStream_t s;

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void measure(const char *name, int launches, const std::function<void()> &fn, int reps = 4000)
{
    fn(); AK(aeth_ctx_sync(ctx));                          // plans' scratch and tables exist before the capture
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    fn();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nodes = 0; CK(hipGraphGetNodes(g, nullptr, &nodes));
    double t[2][2];
    for (int mode = 0; mode < 2; mode++) {
        auto go = [&] { if (mode) CK(hipGraphLaunch(ge, s)); else fn(); };
        for (int i = 0; i < 100; i++) go();
        AK(aeth_ctx_sync(ctx));
        double bt = 1e30, bl = 1e30;
        for (int r = 0; r < 5; r++) {
            double t0 = now_us();
            for (int i = 0; i < reps; i++) go();
            AK(aeth_ctx_sync(ctx));
            bt = std::min(bt, (now_us() - t0) / reps);
            t0 = now_us();
            for (int i = 0; i < reps / 4; i++) { go(); AK(aeth_ctx_sync(ctx)); }
            bl = std::min(bl, (now_us() - t0) / (reps / 4));
        }
        t[mode][0] = bt; t[mode][1] = bl;
    }
    printf("%-62s %2d launches, %2zu graph nodes | back to back: direct %6.2f us  graph %6.2f us (%4.2f x) | a sync each: direct %6.2f  graph %6.2f (%4.2f x)\n",
           name, launches, nodes, t[0][0], t[1][0], t[0][0] / t[1][0], t[0][1], t[1][1], t[0][1] / t[1][1]);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

static aeth_cf32 *dev(size_t n, float v = 1.0f)
{
    void *p; AK(aeth_dev_alloc(ctx, n * 8, &p));
    std::vector<aeth_cf32> h(n);
    for (size_t i = 0; i < n; i++) h[i] = aeth_cf32{v + 1e-3f * (float)(i % 97), -v + 2e-3f * (float)(i % 89)};
    AK(aeth_upload(ctx, p, h.data(), n * 8));
    return (aeth_cf32 *)p;
}

int main()
{
    AK(aeth_ctx_create(0, &ctx));
    s = (hipStream_t)aeth_ctx_stream(ctx);
    // C1
    aeth_cf32 *v = dev(4096), *a = dev(4096, 0.5f), *b = dev(4096, 1.0f);
    measure("C1: add -> mul -> conj on 4096 samples, three calls", 3, [&] { AK(aeth_vec_add(ctx, v, 4096, a, 4096)); AK(aeth_vec_mul(ctx, v, 4096, b, 4096)); AK(aeth_vec_conj(ctx, v, 4096)); });
    aeth_vec_step steps[3] = {{AETH_VEC_ADD, a, 4096, 0.f}, {AETH_VEC_MUL, b, 4096, 0.f}, {AETH_VEC_CONJ, nullptr, 0, 0.f}};
    measure("C1: the same as one fused chain (aeth_vec_chain)", 1, [&] { AK(aeth_vec_chain(ctx, v, 4096, steps, 3)); });
    // C2
    aeth_fft *f; AK(aeth_fft_create(ctx, 2048, 512, &f));
    aeth_cf32 *x = dev(1 << 20), *y = dev(1 << 20);
    measure("C2: fft-2048 fwd (copy) + ifwd (in place) on 1 Mi samples", 2, [&] { AK(aeth_fft_exec(f, x, 1 << 20, y, 512, +1, 1, 0.f)); AK(aeth_fft_exec(f, y, 1 << 20, y, 512, +1, 1, 0.f)); });
    // C4 / 8
    const size_t nfr = 2048, nsym = nfr * 2048;
    aeth_fft *fc; AK(aeth_fft_create(ctx, 2048, nfr, &fc));
    uint8_t *bits[2], *rx[2]; aeth_cf32 *tx[2];
    for (int k = 0; k < 2; k++) {
        void *p; AK(aeth_dev_alloc(ctx, 2 * nsym, &p)); bits[k] = (uint8_t *)p; CK(hipMemset(p, 1, 2 * nsym));
        AK(aeth_dev_alloc(ctx, 2 * nsym, &p)); rx[k] = (uint8_t *)p;
        tx[k] = dev(nsym);
    }
    aeth_cf32 *sig = dev(2048, 0.3f);
    measure("C4 / 8: (modulate_awgn + correlate_demod) x 2 on 2048 frames", 4, [&] {
        for (int k = 0; k < 2; k++) {
            AK(aeth_modulate_awgn(ctx, bits[k], 2 * nsym, 2, nullptr, tx[k], nsym, 0.01f, 815, k * nsym));
            AK(aeth_fft_mul_ifft_demod(fc, tx[k], nsym, nfr, sig, 2048, 0, 0.f, 0, 0.f, 2, nullptr, rx[k], 2 * nsym, 1));
        }
    }, 1000);
    // sixteen single frames
    aeth_fft *f1; AK(aeth_fft_create(ctx, 2048, 1, &f1));
    aeth_cf32 *fr[16];
    for (auto &p : fr) p = dev(2048);
    measure("16 single-frame fft-2048 ifwd calls (the per-frame loop)", 16, [&] { for (auto p : fr) AK(aeth_fft_exec(f1, p, 2048, p, 1, +1, 1, 0.f)); });
    return 0;
}
