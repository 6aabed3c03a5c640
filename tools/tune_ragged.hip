// tune_ragged.hip -- times candidate decompositions of the ragged register-resident FFT (aeth_fft_ragged.h), one
// line per candidate: N, T, WG, radices, us, TB/s.  Every candidate's output is checked against the first one's.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I aether_primitives_amd/csrc -I include \
//         -I tools -DAETH_CANDS=\"tune_ragged_cands.inc\" tools/tune_ragged.hip -o tools/bin/tune_ragged && tools/bin/tune_ragged [N ...]
// (candidate lists: tools/gen_ragged_cands.py)
#include "aeth_fft_ragged.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace aeth::fftk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float2 *g_in, *g_out, *g_ref;
static size_t g_total = (size_t)32 << 20;
static int g_cus = 256;

template <class C>
void run(const char *radices, bool first, int pad)
{
    const size_t batch = g_total / C::N;
    std::vector<float2> tw(C::N);
    for (int k = 0; k < C::N; k++) tw[k] = make_float2((float)cos(-2.0 * M_PI * k / C::N), (float)sin(-2.0 * M_PI * k / C::N));
    float2 *twN, *twL;
    CK(hipMalloc(&twN, C::N * 8));
    CK(hipMalloc(&twL, (size_t)C::TW * C::T * 8));
    CK(hipMemcpy(twN, tw.data(), C::N * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((build_ragged_twiddles<C>), dim3(1), dim3((C::T + 63) / 64 * 64), 0, 0, (const cf *)twN, (cf *)twL);
    CK(hipDeviceSynchronize());
    const size_t ngroups = (batch + C::F - 1) / C::F;
    for (int mult : {4096, 2048, 1024}) {
        size_t cap = (size_t)g_cus * (mult / C::WG);
        int grid = (int)(ngroups < cap ? ngroups : cap);
        float2 *dst = first ? g_ref : g_out;
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const int reps = 100;
        double us = 1e30;
        for (int trial = 0; trial < 4; trial++) {            // best of four: the clock wanders by several percent
            for (int i = 0; i < 10; i++)
                hipLaunchKernelGGL((fft_ragged_kernel<C, 1, true>), dim3(grid), dim3(C::WG), 0, 0, (const cf *)g_in, (cf *)dst, (const cf *)twL, batch, 1.0f);
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; i++)
                hipLaunchKernelGGL((fft_ragged_kernel<C, 1, true>), dim3(grid), dim3(C::WG), 0, 0, (const cf *)g_in, (cf *)dst, (const cf *)twL, batch, 1.0f);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms * 1e3 / reps < us) us = ms * 1e3 / reps;
        }
        double err = 0;
        if (!first) {
            const size_t n = batch * C::N < (1u << 20) ? batch * C::N : (1u << 20);
            std::vector<float2> x(n), y(n);
            CK(hipMemcpy(x.data(), g_ref, n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(y.data(), g_out, n * 8, hipMemcpyDeviceToHost));
            double num = 0, den = 0;
            for (size_t i = 0; i < n; i++) {
                num += (double)(x[i].x - y[i].x) * (x[i].x - y[i].x) + (double)(x[i].y - y[i].y) * (x[i].y - y[i].y);
                den += (double)x[i].x * x[i].x + (double)x[i].y * x[i].y;
            }
            err = 10 * log10(num / den + 1e-30);
        }
        printf("N=%5d T=%4d WG=%4d F=%2d idle=%3d radices=%-12s pad=%d staged=%d db=%d lds=%6d grid=%5d  %7.1f us  %5.2f TB/s  diff %.0f dB\n",
               C::N, C::T, C::WG, C::F, C::IDLE, radices, pad, (int)C::STAGED, (int)C::DB, C::LDS_TOTAL * 8, grid, us,
               16.0 * batch * C::N / us / 1e6, err);
        fflush(stdout);
    }
    CK(hipFree(twN)); CK(hipFree(twL));
}

#define CAND(first, N, T, WG, ...) run<RCfg<N, T, WG, 4, __VA_ARGS__>>(#__VA_ARGS__, first, 4)
#define CANDP(first, N, T, WG, PAD, ...) run<RCfg<N, T, WG, PAD, __VA_ARGS__>>(#__VA_ARGS__, first, PAD)

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    CK(hipMalloc(&g_in, g_total * 8)); CK(hipMalloc(&g_out, g_total * 8)); CK(hipMalloc(&g_ref, g_total * 8));
    std::vector<float2> h(g_total);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v.x = (int)(s >> 8) * (1.0f / (1 << 23)) - 1.0f; s = s * 1664525u + 1013904223u; v.y = (int)(s >> 8) * (1.0f / (1 << 23)) - 1.0f; }
    CK(hipMemcpy(g_in, h.data(), g_total * 8, hipMemcpyHostToDevice));
    auto want = [&](int n) { if (argc < 2) return true; for (int i = 1; i < argc; i++) if (atoi(argv[i]) == n) return true; return false; };
#include AETH_CANDS
    return 0;
}
