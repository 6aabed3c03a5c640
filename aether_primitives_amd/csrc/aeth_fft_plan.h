// aeth_fft_plan.h -- the plan object behind `aeth_fft` (replaces Cfft,
// reference src/fft.rs:134-159: two rustfft plans + tmp of 2*len).
#pragma once

#include "aeth_internal.h"

#include <vector>

struct aeth_fft {
    aeth_ctx *ctx = nullptr;
    size_t len = 0;
    int algo = 0;
    const char *algo_name = "";
    float2 *tw_dev = nullptr;        // exp(-2 pi i k / len), k < len
    float2 *tw_lane_dev = nullptr;   // stockham_pow2: per-lane twiddle registers, [slot][lane]
    float2 *tw_pass_dev = nullptr;   // stockham_mixed: per-pass twiddles, contiguous in the butterfly index
    float2 *tmp_dev = nullptr;       // >= 2*len*max_batch (Cfft.tmp, fft.rs:141,155)
    size_t tmp_elems = 0;
    float2 *tmp_host = nullptr;      // pinned 2*len: what tfwd/tbwd lend out
    std::vector<int> factors;        // stockham_mixed radix schedule
    // fourstep_pow2: len = n1 * n2
    size_t n1 = 0, n2 = 0;
    aeth_fft *sub1 = nullptr, *sub2 = nullptr;   // the N1- and N2-point plans (per-lane twiddle tables)
    float2 *work_dev = nullptr;      // intermediate of the two launches (len * batch)
    size_t work_elems = 0;
    // bluestein: convolution length m (power of two), sub-plan, chirps
    size_t blu_m = 0;
    aeth_fft *blu_sub = nullptr;
    float2 *blu_chirp = nullptr;     // exp(-j pi k^2 / len), k < len
    float2 *blu_filt = nullptr;      // FFT_m of the zero-padded conjugate chirp, / m
};

// overlap-save FIR plan (aeth_fir.hip)
struct aeth_fir {
    aeth_ctx *ctx = nullptr;
    size_t ntaps = 0, fft_len = 0, hop = 0;
    aeth_fft *fft = nullptr;      // owns the twiddle table; used once to transform the taps
    float2 *Hf = nullptr;         // fwd(taps || 0) / N  (1/N folded in: exact, N is a power of two)
};

namespace aeth {

enum { FFT_ALGO_POW2 = 1, FFT_ALGO_MIXED = 2, FFT_ALGO_FOURSTEP = 3, FFT_ALGO_BLUESTEIN = 4, FFT_ALGO_RAGGED = 5, FFT_ALGO_FOURSTEP_MIXED = 6 };

int fft_run(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
int fft_ensure_tmp(aeth_fft *plan, size_t elems);

int fft_plan_fourstep(aeth_fft *plan);
int fft_run_fourstep(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
bool fourstep_small_factor(size_t r);             // first factors the transpose-free path handles in registers
int fft_plan_fourstep_mixed(aeth_fft *plan);     // n1, n2 set by the caller
int fft_run_fourstep_mixed(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
int fft_plan_bluestein(aeth_fft *plan);
int fft_run_bluestein(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
void fft_plan_release_children(aeth_fft *plan);
// aeth_fft_ragged.hip: register-resident transforms of the 5-smooth lengths (table compiled in four slices)
enum { kRaggedNotHere = 1 };
int fft_ragged_slice0(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
int fft_ragged_slice1(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
int fft_ragged_slice2(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
int fft_ragged_slice3(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
bool fft_ragged_supported(size_t len);
int fft_plan_ragged(aeth_fft *plan);
int fft_run_ragged(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale);
// aeth_fir.hip: chirp-z frames through the fused transform * filter * inverse kernel in one launch
int fmi_bluestein(aeth_fft *sub, const float2 *in, float2 *out, size_t n, size_t batch, const float2 *chirp,
                  const float2 *filt, int conj, float scale);

}  // namespace aeth
