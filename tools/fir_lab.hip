// fir_lab.hip -- A/B bench of variants of the fused FIR kernel (aeth_fir_kernel.h) on the C3 geometry:
// FFT-2048, 64 taps, 16 Mi samples per launch, buffers rotating over 1.5 GiB.  Interleaved rounds in one process
// (cdna guide rule 24); every variant's output is compared bit for bit with variant 0 and variant 0 with an f64
// direct convolution on sampled outputs.
//
// build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -Iaether_primitives_amd/csrc tools/fir_lab.hip \
//         -o tools/bin/fir_lab -Laether_primitives_amd/lib -laether_hip -Wl,-rpath,'$ORIGIN/../../aether_primitives_amd/lib'
// run:  tools/bin/fir_lab [steps=300] [rounds=5] [name ...]      (no names: every variant)
#define AETH_FIR_LAB 1        // the diagnosis / measurement variants of fmi_kernel (aeth_fir_kernel.h)
#include "aeth_fft_plan.h"
#include "aeth_fir_kernel.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace aeth::firk;
using C2048 = CfgFor<2048>::type;
using C2048x2 = Cfg<2048, 16, 16, 16, 8, 1, 256>;   // two blocks per 256-lane workgroup: one wave on every SIMD

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "aeth error %d (%s) at %s:%d\n", r_, aeth_last_error(), __FILE__, __LINE__); exit(2); } } while (0)

static const size_t NS = (size_t)1 << 24;
static const int NBUF = 6;

struct Variant {
    const char *name;
    int var;          // kernel template VAR
    int mode;         // 0 plain launch, 1 hipExtAnyOrderLaunch, 2 two streams alternating, 3 the same with the library's event protocol,
                      // 4 / 5 / 6: three / four / six streams round-robin (no events: the lab's buffers make the launches independent)
    int grid;         // 0 = default
};

template <int VAR>
static void launch_var2(const FmiArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((fmi_kernel<C2048x2, false, 1, true, false, VAR>), dim3(grid), dim3(C2048x2::WG), 0, s, a);
}

template <int VAR>
static void launch_var(const FmiArgs &a, int grid, hipStream_t s, int any_order)
{
    if (any_order) {
        hipExtLaunchKernelGGL((fmi_kernel<C2048, false, 1, true, false, VAR>), dim3(grid), dim3(C2048::WG), 0, s, nullptr, nullptr,
                              hipExtAnyOrderLaunch, a);
    } else {
        hipLaunchKernelGGL((fmi_kernel<C2048, false, 1, true, false, VAR>), dim3(grid), dim3(C2048::WG), 0, s, a);
    }
}

template <int VAR>
static void launch_dma(const FmiArgs &a, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((fmi_dma_kernel<C2048, true, VAR>), dim3(grid), dim3(C2048::WG), 0, s, a);
}

static void launch(int var, const FmiArgs &a, int grid, hipStream_t s, int any_order)
{
    switch (var) {
    case (1 << 18) + 2052: launch_dma<2052>(a, grid, s); break;
    case (1 << 18) + 18436: launch_dma<18436>(a, grid, s); break;
    case 1000: launch_var2<0>(a, grid, s); break;
    case 1048: launch_var2<48>(a, grid, s); break;
    case 1176: launch_var2<176>(a, grid, s); break;
    case 1432: launch_var2<176 + 256>(a, grid, s); break;
    case 432: launch_var<176 + 256>(a, grid, s, any_order); break;
    case 2052: launch_var<2048 + 4>(a, grid, s, any_order); break;
    case 18436: launch_var<16384 + 2048 + 4>(a, grid, s, any_order); break;
    case 10244: launch_var<8192 + 2048 + 4>(a, grid, s, any_order); break;
    case 6148: launch_var<4096 + 2048 + 4>(a, grid, s, any_order); break;
    case 2048: launch_var<2048>(a, grid, s, any_order); break;
    case 2096: launch_var<2048 + 48>(a, grid, s, any_order); break;
    case 0: launch_var<0>(a, grid, s, any_order); break;
    case 1: launch_var<1>(a, grid, s, any_order); break;
    case 2: launch_var<2>(a, grid, s, any_order); break;
    case 3: launch_var<3>(a, grid, s, any_order); break;
    case 4: launch_var<4>(a, grid, s, any_order); break;
    case 5: launch_var<5>(a, grid, s, any_order); break;
    case 7: launch_var<7>(a, grid, s, any_order); break;
    case 10: launch_var<10>(a, grid, s, any_order); break;
    case 11: launch_var<11>(a, grid, s, any_order); break;
    case 16: launch_var<16>(a, grid, s, any_order); break;
    case 32: launch_var<32>(a, grid, s, any_order); break;
    case 48: launch_var<48>(a, grid, s, any_order); break;
    case 64: launch_var<64>(a, grid, s, any_order); break;
    case 112: launch_var<112>(a, grid, s, any_order); break;
    case 176: launch_var<176>(a, grid, s, any_order); break;
    case 128: launch_var<128>(a, grid, s, any_order); break;
    default: fprintf(stderr, "variant %d not instantiated\n", var); exit(2);
    }
}

int main(int argc, char **argv)
{
    int steps = argc > 1 ? atoi(argv[1]) : 300;
    int rounds = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<Variant> all = {
        {"base", 0, 0, 0},
        {"peel", 1, 0, 0},
        {"touch", 2, 0, 0},
        {"peel+touch", 3, 0, 0},
        {"prio", 4, 0, 0},
        {"peel+prio", 5, 0, 0},
        {"peel+touch+prio", 7, 0, 0},
        {"touch3", 10, 0, 0},
        {"peel+touch3", 11, 0, 0},
        {"base/anyorder", 0, 1, 0},
        {"peel+touch/anyorder", 3, 1, 0},
        {"base/2q", 0, 2, 0},
        {"peel+touch/2q", 3, 2, 0},
        {"base/g960", 0, 0, 960},
        {"peel+touch/g960", 3, 0, 960},
        {"base/g768", 0, 0, 768}, {"base/g512", 0, 0, 512}, {"base/g256", 0, 0, 256},
        {"noload", 16, 0, 0}, {"nostore", 32, 0, 0},
        {"nobar", 64, 0, 0}, {"nobar/2q", 64, 2, 0}, {"nobar+nomem", 112, 0, 0}, {"nobar+nomem/g512", 112, 0, 512}, {"nobar+nomem/g256", 112, 0, 256},
        {"nolds", 128, 0, 0}, {"nolds+nomem", 176, 0, 0}, {"nolds+nomem/g512", 176, 0, 512}, {"nolds+nomem/g256", 176, 0, 256},
        {"x2", 1000, 0, 512}, {"x2/nomem", 1048, 0, 512}, {"x2/nolds+nomem", 1176, 0, 512},
        {"base/g1280", 0, 0, 1280}, {"base/g1536", 0, 0, 1536}, {"base/g2048", 0, 0, 2048}, {"base/g3072", 0, 0, 3072}, {"base/g4229", 0, 0, 4229}, {"base/g8457", 0, 0, 8457},
        {"base/g2048/2q", 0, 2, 2048}, {"base/g4229/2q", 0, 2, 4229},
        {"prio/2q+events", 4, 3, 0}, {"base/2q+events", 0, 3, 0},
        {"xor+prio+unroll2", 6148, 0, 0}, {"xor+prio+unroll2/2q", 6148, 2, 0},
        {"xor", 2048, 0, 0}, {"xor+prio", 2052, 0, 0}, {"xor+prio/2q", 2052, 2, 0}, {"xor+nomem", 2096, 0, 0},
        {"xor+prio+spread", 18436, 0, 0}, {"xor+prio+spread/2q", 18436, 2, 0}, {"xor+prio+spread/g768/2q", 18436, 2, 768},
        {"xor+prio/g768/3q", 2052, 4, 768}, {"xor+prio/g768/4q", 2052, 5, 768}, {"xor+prio/g768/6q", 2052, 6, 768},
        {"xor+prio/g640/3q", 2052, 4, 640}, {"xor+prio/g640/4q", 2052, 5, 640}, {"xor+prio/g512/3q", 2052, 4, 512}, {"xor+prio/g512/4q", 2052, 5, 512},
        {"xor+prio/g896/3q", 2052, 4, 896}, {"xor+prio/g1024/3q", 2052, 4, 0}, {"xor+prio/g1024/4q", 2052, 5, 0},
        {"xor+prio/g384/4q", 2052, 5, 384}, {"xor+prio/g384/6q", 2052, 6, 384}, {"xor+prio/g512/6q", 2052, 6, 512},
        {"xor+prio+xcd", 10244, 0, 0}, {"xor+prio+xcd/2q", 10244, 2, 0},
        {"xor+prio/g768/2q", 2052, 2, 768}, {"xor+prio/g896/2q", 2052, 2, 896}, {"xor+prio/g768", 2052, 0, 768},
        {"xor+prio/g704/2q", 2052, 2, 704}, {"xor+prio/g736/2q", 2052, 2, 736}, {"xor+prio/g800/2q", 2052, 2, 800},
        {"xor+prio/g832/2q", 2052, 2, 832}, {"xor+prio/g960/2q", 2052, 2, 960},
        // round 4: the window through an LDS landing image (V_DMA), burst and spread, one queue and two
        {"dma", (1 << 18) + 2052, 0, 0}, {"dma/2q", (1 << 18) + 2052, 2, 0}, {"dma/g768/2q", (1 << 18) + 2052, 2, 768},
        {"dma+spread", (1 << 18) + 18436, 0, 0}, {"dma+spread/2q", (1 << 18) + 18436, 2, 0}, {"dma+spread/g768/2q", (1 << 18) + 18436, 2, 768},
        // round 4: 8457 blocks = 11 x 768 + 9 -- grids that divide the stream into whole rounds (769: 11 rounds, 705: 12, 846: 10)
        {"xor+prio/g769/2q", 2052, 2, 769}, {"xor+prio/g705/2q", 2052, 2, 705}, {"xor+prio/g846/2q", 2052, 2, 846},
        {"nolds/2q", 128, 2, 0}, {"prio/2q", 4, 2, 0}, {"peel/2q", 1, 2, 0},
        {"nomem", 48, 0, 0}, {"nomem/g768", 48, 0, 768}, {"nomem/g512", 48, 0, 512}, {"nomem/g256", 48, 0, 256},
    };
    std::vector<Variant> vs;
    if (argc > 3) {
        for (int i = 3; i < argc; i++) {
            bool found = false;
            for (auto &v : all) if (!strcmp(v.name, argv[i])) { vs.push_back(v); found = true; }
            if (!found) { fprintf(stderr, "unknown variant %s\n", argv[i]); return 2; }
        }
    } else vs = all;

    aeth_ctx *ctx = nullptr;
    AK(aeth_ctx_create(0, &ctx));
    hipStream_t s0 = (hipStream_t)aeth_ctx_stream(ctx), s1, sx[6];
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    sx[0] = s0; sx[1] = s1;
    for (int i = 2; i < 6; i++) CK(hipStreamCreateWithFlags(&sx[i], hipStreamNonBlocking));

    // taps: 64-tap Hamming-windowed sinc, unit DC gain (bench.py lowpass_taps)
    const int NT_ = 64;
    std::vector<aeth_cf32> taps(NT_);
    {
        std::vector<double> t(NT_);
        double sum = 0;
        for (int k = 0; k < NT_; k++) {
            double m = k - (NT_ - 1) / 2.0;
            double sinc = std::sin(2 * M_PI * 0.25 * m) / (M_PI * m);
            double w = 0.54 - 0.46 * std::cos(2 * M_PI * k / (NT_ - 1));
            t[k] = sinc * w; sum += t[k];
        }
        for (int k = 0; k < NT_; k++) taps[k] = aeth_cf32{(float)(t[k] / sum), 0.f};
    }
    aeth_fir *fir = nullptr;
    AK(aeth_fir_create(ctx, taps.data(), NT_, 2048, &fir));

    // input: deterministic pseudo-random cf32 in [-1, 1)
    std::vector<aeth_cf32> x(NS);
    {
        uint64_t st = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < NS; i++) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            uint32_t a = (uint32_t)(st >> 40), b = (uint32_t)(st >> 16) & 0xFFFFFF;
            x[i] = aeth_cf32{(float)a / 8388608.0f - 1.0f, (float)b / 8388608.0f - 1.0f};
        }
    }
    float2 *in[NBUF], *out[NBUF];
    for (int i = 0; i < NBUF; i++) {
        CK(hipMalloc((void **)&in[i], NS * 8));
        CK(hipMalloc((void **)&out[i], NS * 8));
        CK(hipMemcpy(in[i], x.data(), NS * 8, hipMemcpyHostToDevice));
        CK(hipMemset(out[i], 0xFF, NS * 8));
    }

    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cap = prop.multiProcessorCount * 8 / (C2048::WG / 64);
    auto args_for = [&](int b) {
        FmiArgs a;
        a.in = (const cf *)in[b]; a.out = (cf *)out[b]; a.hist = nullptr; a.Hf = (const cf *)fir->Hf;
        a.twN = (const cf *)fir->fft->tw_dev; a.twL = (const cf *)fir->fft->tw_lane_dev;
        a.n = (long long)NS; a.hop = (int)fir->hop; a.ov = (int)(fir->fft_len - fir->hop); a.nhist = (int)(fir->ntaps - 1);
        a.nblocks = (long long)((NS + fir->hop - 1) / fir->hop);
        a.s_fwd = 1.0f; a.s_bwd = 1.0f; a.frame_n = 2048;
        return a;
    };
    hipEvent_t ev_pre[2];
    CK(hipEventCreateWithFlags(&ev_pre[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_pre[1], hipEventDisableTiming));
    CK(hipEventRecord(ev_pre[0], s0)); CK(hipEventRecord(ev_pre[1], s1));
    auto issue = [&](const Variant &v, int i) {
        const int grid = v.grid ? v.grid : cap;
        const int nq = v.mode == 4 ? 3 : v.mode == 5 ? 4 : v.mode == 6 ? 6 : (v.mode >= 2 ? 2 : 1);
        const int lane = i % nq;
        hipStream_t s = sx[lane];
        if (v.mode == 3) {       // aeth_runtime.hip: ctx_fir_lane -- wait for the other lane's history, record this lane's
            CK(hipStreamWaitEvent(s, ev_pre[1 - lane], 0));
            CK(hipEventRecord(ev_pre[lane], s));
        }
        launch(v.var, args_for(i % NBUF), grid, s, v.mode == 1);
    };
    auto sync_all = [&] { for (int i = 0; i < 6; i++) CK(hipStreamSynchronize(sx[i])); };

    // correctness: variant 0 against an f64 direct convolution on sampled outputs, every variant against variant 0
    std::vector<aeth_cf32> ref(NS), got(NS);
    issue(all[0], 0); sync_all(); CK(hipGetLastError());
    CK(hipMemcpy(ref.data(), out[0], NS * 8, hipMemcpyDeviceToHost));
    {
        double e2 = 0, r2 = 0;
        auto check = [&](size_t n0) {
            double re = 0, im = 0;
            for (int k = 0; k < NT_; k++) {
                if (n0 < (size_t)k) break;
                re += (double)taps[k].re * x[n0 - k].re; im += (double)taps[k].re * x[n0 - k].im;
            }
            double dr = ref[n0].re - re, di = ref[n0].im - im;
            e2 += dr * dr + di * di; r2 += re * re + im * im;
        };
        for (size_t n0 = 0; n0 < 6000; n0++) check(n0);
        for (size_t n0 = NS - 6000; n0 < NS; n0++) check(n0);
        for (size_t j = 0; j < 200000; j++) check((j * 2654435761ull) % NS);
        printf("variant 0 vs f64 direct convolution: EVM %.1f dB\n", 10 * std::log10(e2 / r2));
    }
    for (auto &v : vs) {
        CK(hipMemset(out[0], 0xFF, NS * 8));
        issue(v, 0); sync_all(); CK(hipGetLastError());
        CK(hipMemcpy(got.data(), out[0], NS * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < NS; i++) bad += memcmp(&got[i], &ref[i], 8) != 0;
        printf("%-24s output %s (%zu samples differ)\n", v.name, bad ? "DIFFERS" : "bit-identical", bad);
    }

    if (getenv("LAB_CENSUS")) {
        unsigned *cb; CK(hipMalloc((void **)&cb, 1 << 20));
        for (int mode = 0; mode < 4; mode++) {
            const int grids[4] = {256, 512, 1024, 512};
            const int grid = grids[mode]; const bool x2 = mode == 3;
            CK(hipMemset(cb, 0xFF, 1 << 20));
            FmiArgs a = args_for(0); a.chirp = (const cf *)cb;
            launch(x2 ? 1432 : 432, a, grid, s0, 0); sync_all(); CK(hipGetLastError());
            const int nw = grid * (x2 ? 4 : 2);
            std::vector<unsigned> h(2 * nw); CK(hipMemcpy(h.data(), cb, 8 * nw, hipMemcpyDeviceToHost));
            // waves per (xcc, se, cu, simd)
            std::vector<int> cnt(8 * 8 * 16 * 4, 0);
            for (int w = 0; w < nw; w++) {
                unsigned id = h[2 * w], xcc = h[2 * w + 1] & 15;
                unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, se = (id >> 13) & 7;
                cnt[((xcc * 8 + se) * 16 + cu) * 4 + simd]++;
            }
            int hist[16] = {0}, cus = 0, cuhist[64] = {0};
            for (int c = 0; c < 8 * 8 * 16; c++) {
                int tot = cnt[4 * c] + cnt[4 * c + 1] + cnt[4 * c + 2] + cnt[4 * c + 3];
                if (tot) { cus++; cuhist[tot]++; for (int k = 0; k < 4; k++) hist[cnt[4 * c + k]]++; }
            }
            printf("census %s grid %d: %d CUs used; waves per CU:", x2 ? "x2(256 lanes)" : "128 lanes", grid, cus);
            for (int k = 0; k < 64; k++) if (cuhist[k]) printf(" %d:%d", k, cuhist[k]);
            printf("; waves per SIMD (of used CUs):");
            for (int k = 0; k < 16; k++) if (hist[k]) printf(" %d:%d", k, hist[k]);
            printf("\n  first WGs:");
            for (int w = 0; w < 16; w++) printf(" [x%u s%u c%u simd%u]", h[2*w+1] & 15, (h[2*w] >> 13) & 7, (h[2*w] >> 8) & 15, (h[2*w] >> 4) & 3);
            printf("\n");
        }
    }

    // settle: ~80 ms of launches (load-onset power transient)
    for (int i = 0; i < 1500; i++) issue(vs[0], i);
    sync_all();

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<double>> wall(vs.size()), evt(vs.size());
    for (int r = 0; r < rounds; r++) {
        for (size_t k = 0; k < vs.size(); k++) {
            for (int i = 0; i < 30; i++) issue(vs[k], i);
            sync_all();
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s0));
            for (int i = 0; i < steps; i++) issue(vs[k], i);
            CK(hipEventRecord(e1, s0));
            sync_all();
            auto t1 = std::chrono::steady_clock::now();
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            wall[k].push_back(std::chrono::duration<double>(t1 - t0).count() / steps * 1e6);
            evt[k].push_back(ms * 1e3 / steps);
        }
    }
    printf("%-24s %10s %10s %10s %10s %8s\n", "variant", "wall med", "wall min", "event med", "GS/s", "% 8TB/s");
    for (size_t k = 0; k < vs.size(); k++) {
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        double wm = med(wall[k]), wmin = *std::min_element(wall[k].begin(), wall[k].end()), em = med(evt[k]);
        printf("%-24s %9.2fus %9.2fus %9.2fus %10.1f %8.2f\n", vs[k].name, wm, wmin, em, NS / wm / 1e3, 16.0 * NS / wm / 8e6 * 100);
    }
    return 0;
}
