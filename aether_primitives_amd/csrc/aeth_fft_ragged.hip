// aeth_fft_ragged.hip -- the "stockham_mixed_ragged" path of trait Fft (reference src/fft.rs:48-77): lengths with
// factors 2, 3 and 5 between 100 and 7200, one tuned decomposition each (aeth_fft_ragged.h has the kernel).
#include "aeth_fft_ragged.h"
#include "aeth_fft_plan.h"

using namespace aeth::fftk;

namespace {

// One measured decomposition per length (tools/gen_ragged_cands.py + tools/tune_ragged.hip: up to 40 shapes per
// length, best of four timings of 100 launches over 32 Mi samples each):  N, lanes per frame, workgroup size,
// radices; LANES_PER_CU caps the persistent grid.
template <int N> struct RaggedFor;
template <> struct RaggedFor<100> { using type = RCfg<100, 10, 128, 10, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<120> { using type = RCfg<120, 12, 128, 12, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<200> { using type = RCfg<200, 20, 128, 4, 5, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<240> { using type = RCfg<240, 16, 64, 16, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<300> { using type = RCfg<300, 30, 128, 12, 5, 5>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<360> { using type = RCfg<360, 30, 64, 12, 5, 6>; static constexpr int LANES_PER_CU = 2048; };
template <> struct RaggedFor<400> { using type = RCfg<400, 40, 128, 4, 10, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<480> { using type = RCfg<480, 16, 64, 6, 8, 10>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<500> { using type = RCfg<500, 125, 128, 4, 5, 5, 5>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<600> { using type = RCfg<600, 60, 64, 12, 5, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<625> { using type = RCfg<625, 125, 256, 5, 5, 5, 5>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<640> { using type = RCfg<640, 80, 256, 8, 8, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<720> { using type = RCfg<720, 60, 64, 4, 12, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<800> { using type = RCfg<800, 80, 256, 10, 8, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<900> { using type = RCfg<900, 60, 64, 15, 4, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<960> { using type = RCfg<960, 64, 64, 16, 15, 4>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<1000> { using type = RCfg<1000, 50, 64, 5, 10, 20>; static constexpr int LANES_PER_CU = 2048; };
template <> struct RaggedFor<1080> { using type = RCfg<1080, 60, 64, 6, 6, 6, 5>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<1200> { using type = RCfg<1200, 120, 128, 12, 10, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<1280> { using type = RCfg<1280, 80, 256, 16, 8, 10>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<1440> { using type = RCfg<1440, 96, 192, 16, 15, 6>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<1500> { using type = RCfg<1500, 75, 256, 25, 3, 20>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<1600> { using type = RCfg<1600, 80, 256, 4, 20, 20>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<1800> { using type = RCfg<1800, 120, 256, 8, 15, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<1920> { using type = RCfg<1920, 128, 256, 16, 15, 8>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<2000> { using type = RCfg<2000, 100, 256, 5, 20, 20>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<2160> { using type = RCfg<2160, 180, 192, 12, 12, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<2400> { using type = RCfg<2400, 120, 256, 24, 20, 5>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<2500> { using type = RCfg<2500, 250, 256, 10, 5, 5, 10>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<2560> { using type = RCfg<2560, 256, 256, 16, 10, 16>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<2880> { using type = RCfg<2880, 240, 256, 16, 15, 12>; static constexpr int LANES_PER_CU = 2048; };
template <> struct RaggedFor<3000> { using type = RCfg<3000, 125, 128, 25, 24, 5>; static constexpr int LANES_PER_CU = 2048; };
template <> struct RaggedFor<3072> { using type = RCfg<3072, 256, 256, 16, 16, 12>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<3600> { using type = RCfg<3600, 240, 256, 16, 15, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<3840> { using type = RCfg<3840, 256, 256, 16, 16, 15>; static constexpr int LANES_PER_CU = 4096; };
template <> struct RaggedFor<4000> { using type = RCfg<4000, 250, 256, 8, 20, 25>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<4320> { using type = RCfg<4320, 240, 256, 6, 6, 6, 20>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<4800> { using type = RCfg<4800, 240, 256, 24, 20, 10>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<5000> { using type = RCfg<5000, 500, 512, 10, 10, 10, 5>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<5120> { using type = RCfg<5120, 320, 320, 16, 16, 20>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<5400> { using type = RCfg<5400, 360, 384, 24, 15, 15>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<5760> { using type = RCfg<5760, 240, 256, 10, 24, 24>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<6000> { using type = RCfg<6000, 400, 512, 25, 16, 15>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<6144> { using type = RCfg<6144, 384, 384, 24, 16, 16>; static constexpr int LANES_PER_CU = 1024; };
template <> struct RaggedFor<7200> { using type = RCfg<7200, 480, 512, 24, 20, 15>; static constexpr int LANES_PER_CU = 1024; };

#define AETH_RAGGED_SWITCH(len, BODY, DEFAULT)                                                          \
    switch (len) {                                                                                      \
    case 100: BODY(100); case 120: BODY(120); case 200: BODY(200); case 240: BODY(240);                 \
    case 300: BODY(300); case 360: BODY(360); case 400: BODY(400); case 480: BODY(480);                 \
    case 500: BODY(500); case 600: BODY(600); case 625: BODY(625); case 640: BODY(640);                 \
    case 720: BODY(720); case 800: BODY(800); case 900: BODY(900); case 960: BODY(960);                 \
    case 1000: BODY(1000); case 1080: BODY(1080); case 1200: BODY(1200); case 1280: BODY(1280);         \
    case 1440: BODY(1440); case 1500: BODY(1500); case 1600: BODY(1600); case 1800: BODY(1800);         \
    case 1920: BODY(1920); case 2000: BODY(2000); case 2160: BODY(2160); case 2400: BODY(2400);         \
    case 2500: BODY(2500); case 2560: BODY(2560); case 2880: BODY(2880); case 3000: BODY(3000);         \
    case 3072: BODY(3072); case 3600: BODY(3600); case 3840: BODY(3840); case 4000: BODY(4000);         \
    case 4320: BODY(4320); case 4800: BODY(4800); case 5000: BODY(5000); case 5120: BODY(5120);         \
    case 5400: BODY(5400); case 5760: BODY(5760); case 6000: BODY(6000); case 6144: BODY(6144);         \
    case 7200: BODY(7200);                                                                              \
    default: DEFAULT;                                                                                   \
    }

template <class C, int LANES_PER_CU>
int launch_ragged(const aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const aeth_ctx *ctx = plan->ctx;
    const bool nt = aeth::streams_past_cache(2 * batch * (size_t)C::N * sizeof(float2));
    const size_t ngroups = (batch + C::F - 1) / C::F;
    const size_t cap = (size_t)ctx->num_cus * (LANES_PER_CU / C::WG);
    int grid = (int)(ngroups < cap ? ngroups : cap);
    if (grid < 1) grid = 1;
#define AETH_FFT_RAGGED(SS, NN) hipLaunchKernelGGL((fft_ragged_kernel<C, SS, NN>), dim3(grid), dim3(C::WG), 0, ctx->stream, (const cf *)in, (cf *)out, (const cf *)plan->tw_lane_dev, batch, scale)
    if (sign > 0) { if (nt) AETH_FFT_RAGGED(+1, true); else AETH_FFT_RAGGED(+1, false); }
    else          { if (nt) AETH_FFT_RAGGED(-1, true); else AETH_FFT_RAGGED(-1, false); }
#undef AETH_FFT_RAGGED
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

template <class C>
int build_ragged_table(aeth_fft *plan)
{
    AETH_HIP(hipMalloc((void **)&plan->tw_lane_dev, (size_t)C::TW * C::T * sizeof(float2)));
    hipLaunchKernelGGL((build_ragged_twiddles<C>), dim3(1), dim3((C::T + 63) / 64 * 64), 0, plan->ctx->stream,
                       (const cf *)plan->tw_dev, (cf *)plan->tw_lane_dev);
    AETH_HIP(hipGetLastError());
    AETH_HIP(hipStreamSynchronize(plan->ctx->stream));
    return AETH_OK;
}

}  // namespace

namespace aeth {

bool fft_ragged_supported(size_t len)
{
#define AETH_BODY(NN) return true
    AETH_RAGGED_SWITCH(len, AETH_BODY, return false)
#undef AETH_BODY
}

int fft_plan_ragged(aeth_fft *plan)
{
#define AETH_BODY(NN) return build_ragged_table<typename RaggedFor<NN>::type>(plan)
    AETH_RAGGED_SWITCH(plan->len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "stockham_mixed_ragged: length %zu", plan->len))
#undef AETH_BODY
}

int fft_run_ragged(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
#define AETH_BODY(NN) return launch_ragged<typename RaggedFor<NN>::type, RaggedFor<NN>::LANES_PER_CU>(plan, in, out, batch, sign, scale)
    AETH_RAGGED_SWITCH(plan->len, AETH_BODY, return aeth::set_error(AETH_E_UNSUPPORTED, "stockham_mixed_ragged: length %zu", plan->len))
#undef AETH_BODY
}

}  // namespace aeth
