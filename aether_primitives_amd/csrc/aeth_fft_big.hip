// aeth_fft_big.hip -- transforms that do not fit one workgroup's LDS.
//
// fourstep_pow2 (N = 2^14 .. 2^24; 8192 and 16384 run in one workgroup, aeth_fft_ragged.hip; BASELINE config 5 uses
// 65536 = 256 x 256; 2^23 and 2^24 take the DEEP form -- 128 / 256 columns over rows of 65536 points that are
// four-step transforms of their own, then a tile transpose: four launches, and the sub-plan keeps a second
// full-size work buffer, i.e. 2 x 8 B/sample of scratch, 256 MiB for one 2^24-point frame):
//   x[n1*N2 + n2]  --A-->  a[k1*N2 + n2] = W_N^(n2*k1) * sum_n1 x[n1*N2+n2] W_N1^(n1*k1)
//                  --B-->  X[k1 + N1*k2] = sum_n2 a[k1*N2+n2] W_N2^(n2*k2)
//   A: a workgroup owns G adjacent columns (G*8 B contiguous per row: 128 B for G=16),
//      one register-resident N1-point transform per column, twiddle, store in place
//      layout into the plan's work buffer.
//   B: a workgroup owns G adjacent rows (each row contiguous), one N2-point transform
//      per row, then the N1-strided output is transposed through LDS so that stores
//      again cover G adjacent k1 (128 B segments).
//   HBM traffic 32 B/sample unless the 8 B/sample intermediate stays in the 256 MiB
//   Infinity Cache (it does for batches up to ~16 frames of 65536).
//
// bluestein (any other length): chirp-z through a power-of-two circular convolution
//   of length M >= 2N-1, itself run by the paths above.
#include "aeth_internal.h"
#include "aeth_fft_core.h"
#include "aeth_fft_plan.h"

#include <cmath>
#include <cstdlib>
#include <vector>

using namespace aeth::fftk;

#ifndef AETH_4S_LDS_LIMIT
#define AETH_4S_LDS_LIMIT (144 * 1024)
#endif

namespace {

// single LDS image regardless of size (the column/row kernels hold G frames per workgroup)
template <class C> struct OneImage : C { static constexpr bool DB = false; };

// LDS stride between the frames of a column group.  Adjacent LANES are adjacent COLUMNS
// here, so the frame stride decides the banking: one exchange image is LDS_FRAME elements
// (a multiple of 16 elements = 32 banks, which would put all 16 columns on two bank
// positions: 8-way conflicts on every access); a stride of 1 element modulo 32 elements
// makes the 16 lanes of a write group land on 16 consecutive 8-byte slots.
template <class C> constexpr int col_stride() { return C::LDS_FRAME + ((33 - C::LDS_FRAME % 32) % 32); }

template <class C> constexpr int group_of()
{
    // frames per workgroup: 16 for 128-byte segments, fewer when lanes or LDS run out
    int g = 16;
    while (g > 1 && (g * C::T > 1024 || g * col_stride<C>() * 8 > AETH_4S_LDS_LIMIT || C::N * (g + 1) * 8 > AETH_4S_LDS_LIMIT)) g /= 2;
    return g;
}

// columns per workgroup of step A: GW wanted (16 = 128-byte segments, 32 = 256-byte), fewer when lanes or LDS run out
template <class C, int GW, int MAXL = 1024> constexpr int col_group_of()
{
    int g = GW;
    while (g > 1 && (g * C::T > MAXL || g * col_stride<C>() * 8 > AETH_4S_LDS_LIMIT)) g /= 2;
    return g;
}

// ---- step A: G columns per workgroup ---------------------------------------------
// NT: x is read / X is written with the non-temporal hint, which leaves L2 and the Infinity Cache to the
// intermediate (batch 512 x 65536: 212 -> 159 us); small batches that fit the cache whole do better without
template <class C0, int S, bool NT, int GW = 16, int MAXL = 1024>
__global__ __launch_bounds__((col_group_of<C0, GW, MAXL>()) * C0::T) void fourstep_cols(const cf *in, cf *work,
                                                                          const cf *__restrict__ twL1,
                                                                          const cf *__restrict__ twN, int N2, size_t N)
{
    using C = OneImage<C0>;
    constexpr int G = col_group_of<C0, GW, MAXL>();
    constexpr int CS = col_stride<C0>();
    __shared__ cf lds_all[G * CS > 0 ? G * CS : 1];
    const int col = threadIdx.x % G;
    const int tid = threadIdx.x / G;
    const int groups_per_frame = N2 / G;
    const size_t frame = blockIdx.x / groups_per_frame;
    const int n2 = (blockIdx.x % groups_per_frame) * G + col;
    const cf *src = in + frame * N + n2;
    cf *dst = work + frame * N + n2;
    cf tw[C::TW];
    load_twiddles_lane<C>(tw, twL1, tid);
    cf w[C::P];
#pragma unroll
    for (int m = 0; m < C::P; m++) w[m] = NT ? __builtin_nontemporal_load(src + (size_t)(tid + m * C::T) * N2) : src[(size_t)(tid + m * C::T) * N2];
    fft_in_regs<C, S, 0>(w, tw, lds_all + col * CS, tid);
    // twiddle W_N^(n2*k1), k1 = tid + m*T: a geometric sequence in m for this lane,
    //   W_N^(n2*tid) * (W_N^(T*n2))^m.
    // Three table reads (a, b, b^4) and short product chains (depth <= 5) replace one
    // scattered 8-byte gather per element, which costs more than the arithmetic here.
    {
        const cf a = twN[(size_t)n2 * tid];
        const cf b1 = twN[(size_t)n2 * C::T];
        const cf b4 = twN[(size_t)n2 * C::T * 4];
        const cf b2 = cmul(b1, b1), b3 = cmul(b2, b1);
        cf base = a;
#pragma unroll
        for (int q = 0; q < C::P / 4; q++) {
            w[4 * q + 0] = ctw<S>(w[4 * q + 0], base);
            w[4 * q + 1] = ctw<S>(w[4 * q + 1], cmul(base, b1));
            w[4 * q + 2] = ctw<S>(w[4 * q + 2], cmul(base, b2));
            w[4 * q + 3] = ctw<S>(w[4 * q + 3], cmul(base, b3));
            base = cmul(base, b4);
        }
    }
#pragma unroll
    for (int m = 0; m < C::P; m++) dst[(size_t)(tid + m * C::T) * N2] = w[m];
}

// ---- step B: G rows per workgroup, transposed store --------------------------------
template <class C0, int S, bool NT>
__global__ __launch_bounds__(group_of<C0>() * C0::T) void fourstep_rows(const cf *work, cf *out,
                                                                          const cf *__restrict__ twL2, int N1, size_t N,
                                                                          float scale)
{
    using C = OneImage<C0>;
    constexpr int G = group_of<C0>();
    constexpr int WGS = G * C::T;
    constexpr int XP = C::N * (G + 1);                     // transpose image: [k2][row], one pad per k2
    constexpr int FP = G * C::LDS_FRAME;
    __shared__ cf lds_all[(XP > FP ? XP : FP)];
    const int row = threadIdx.x / C::T;
    const int tid = threadIdx.x % C::T;
    const int groups_per_frame = N1 / G;
    const size_t frame = blockIdx.x / groups_per_frame;
    const int k1_0 = (blockIdx.x % groups_per_frame) * G;
    const cf *src = work + frame * N + (size_t)(k1_0 + row) * C::N + tid;
    cf tw[C::TW];
    load_twiddles_lane<C>(tw, twL2, tid);
    cf w[C::P];
#pragma unroll
    for (int m = 0; m < C::P; m++) w[m] = src[m * C::T];
    fft_in_regs<C, S, 0>(w, tw, lds_all + row * C::LDS_FRAME, tid);
    __syncthreads();                                       // the transform's LDS image is dead
    const cf ss = mk(scale, scale);
#pragma unroll
    for (int m = 0; m < C::P; m++) lds_all[(tid + m * C::T) * (G + 1) + row] = cscale_k(w[m], ss);
    __syncthreads();
    const int r2 = threadIdx.x % G;                        // adjacent lanes -> adjacent k1
    const int kk = threadIdx.x / G;
    cf *dst = out + frame * N + k1_0 + r2;
#pragma unroll
    for (int j = 0; j < C::N * G / WGS; j++) {
        const int k2 = kk + j * (WGS / G);
        if (NT) __builtin_nontemporal_store(lds_all[k2 * (G + 1) + r2], dst + (size_t)k2 * N1);
        else dst[(size_t)k2 * N1] = lds_all[k2 * (G + 1) + r2];
    }
}

// ---- step B fused with sampling::interpolate: built, measured, NOT kept (round 2; profiles/r02_c5.json) -------------
// BASELINE config 5 is `frame.vec_rfft(fft, s)` then `sampling::interpolate(&frame, &mut dst, 9)`.  A kernel that
// transforms G rows plus the next one (X[i+1] of the row group's last element), transposes through LDS as step B does
// and writes the interpolated runs of G*(nb+1) outputs per k2 instead of X was bit-identical to the two calls and
// saves 16 of the chain's 104 B/sample -- but ran the chain (512 x 65536, nb = 9) in 875 us (1058 with per-output index
// divisions, 1231 with descriptor-masked stores) against 653 us for the two kernels: the output phase writes 327 KB
// per workgroup from 272 lanes that first had to transform 17 rows (41 KiB of LDS, 3 workgroups per CU), and a
// store-bound kernel wants the 32 KiB per CU in flight that aeth_sampling.hip's interpolate_kernel32 has (6.4 TB/s).
// aeth_fft_exec_interpolate therefore runs the two steps through the plan's temp.

template <class C, int S>
int launch_cols(aeth_fft *plan, const float2 *in, size_t batch, size_t batch_total, hipStream_t stream = nullptr, size_t work_off = 0)
{
    if (!stream) stream = aeth::ctx_stream(plan->ctx);
    const bool nt = aeth::streams_past_cache(plan->len * batch_total * sizeof(float2) * 4 / 3);   // from 96 MiB: x, a and X together pass the cache
    // 32 adjacent columns per workgroup (256-byte segments) where lanes and LDS allow (n1 <= 256): 512 x 65536 runs in
    // 169 us against 177 us with 16 (tools/tune_4step.py, AETH_4S_COLG), no difference on small batches
    const bool wide = aeth::lab_int("AETH_4S_COLG", 32) >= 32 && col_group_of<C, 32>() == 32 && plan->n2 % 32 == 0;
    // columns of 1024 points and more (64+ lanes each): 16 of them make a 1024-lane workgroup, which caps the kernel at
    // 128 VGPRs and spills 22-28 of them; 8 columns (512 lanes, 256 VGPRs, 64-byte row segments) is the alternative
    const bool half = !wide && C::T >= 64 && aeth::lab_int("AETH_4S_MAXL", 1024) <= 512;
    const int G = wide ? 32 : half ? col_group_of<C, 16, 512>() : col_group_of<C, 16>();
    const size_t grid = batch * (plan->n2 / G);
    auto kern = wide ? (nt ? fourstep_cols<C, S, true, 32> : fourstep_cols<C, S, false, 32>)
                : half ? (nt ? fourstep_cols<C, S, true, 16, 512> : fourstep_cols<C, S, false, 16, 512>)
                       : (nt ? fourstep_cols<C, S, true, 16> : fourstep_cols<C, S, false, 16>);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G * C::T), 0, stream,
                       (const cf *)in, (cf *)(plan->work_dev + work_off), (const cf *)plan->sub1->tw_lane_dev,
                       (const cf *)plan->tw_dev, (int)plan->n2, plan->len);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

template <class C, int S>
int launch_rows(aeth_fft *plan, float2 *out, size_t batch, float scale, size_t batch_total, hipStream_t stream = nullptr, size_t work_off = 0)
{
    if (!stream) stream = aeth::ctx_stream(plan->ctx);
    constexpr int G = group_of<C>();
    const size_t grid = batch * (plan->n1 / G);
    const bool nt = aeth::streams_past_cache(plan->len * batch_total * sizeof(float2) * 4 / 3);
    auto kern = nt ? fourstep_rows<C, S, true> : fourstep_rows<C, S, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G * C::T), 0, stream,
                       (const cf *)(plan->work_dev + work_off), (cf *)out, (const cf *)plan->sub2->tw_lane_dev, (int)plan->n1,
                       plan->len, scale);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int ensure_work(aeth_fft *plan, size_t elems)
{
    if (plan->work_elems >= elems) return AETH_OK;
    aeth::DeviceGuard dev_guard(plan->ctx->device);        // the plan's device, not the caller's current one
    if (plan->work_dev) {
        AETH_HIP(hipStreamSynchronize(aeth::ctx_stream(plan->ctx)));
        AETH_HIP(hipFree(plan->work_dev));
        plan->work_dev = nullptr;
        plan->work_elems = 0;
    }
    AETH_HIP(hipMalloc((void **)&plan->work_dev, elems * sizeof(float2)));
    plan->work_elems = elems;
    return AETH_OK;
}

// ---- bluestein helpers ---------------------------------------------------------------
constexpr int kBlock = 256;

// a[f*M + n] = (n < N) ? x[f*N + n] (conj if CONJ) * chirp[n] : 0
template <bool CONJ>
__global__ __launch_bounds__(kBlock) void blu_pre(const cf *__restrict__ x, const cf *__restrict__ chirp,
                                                  cf *__restrict__ a, size_t N, size_t M, size_t batch)
{
    const size_t total = M * batch;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        size_t f = i / M, n = i - f * M;
        cf v = mk(0.f, 0.f);
        if (n < N) {
            cf xv = x[f * N + n];
            if (CONJ) xv.y = -xv.y;
            v = cmul_plain(xv, chirp[n]);
        }
        a[i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void blu_mul(cf *__restrict__ a, const cf *__restrict__ filt, size_t M, size_t batch)
{
    const size_t total = M * batch;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock)
        a[i] = cmul_plain(a[i], filt[i % M]);
}

template <bool CONJ>
__global__ __launch_bounds__(kBlock) void blu_post(const cf *__restrict__ a, const cf *__restrict__ chirp,
                                                   cf *__restrict__ out, size_t N, size_t M, size_t batch, float scale)
{
    const size_t total = N * batch;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
        size_t f = i / N, k = i - f * N;
        cf v = cmul_plain(a[f * M + k], chirp[k]);
        if (CONJ) v.y = -v.y;
        out[i] = cscale(v, scale);
    }
}

inline int grid_for(const aeth_ctx *ctx, size_t items)
{
    size_t blocks = (items + kBlock - 1) / kBlock;
    size_t cap = (size_t)ctx->num_cus * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

// ---- fourstep_mixed: batched transposes around the batched transforms of the two factors ----
// out[c][r] = in[r][c] (* W_len^(r c) when S != 0) for `batch` frames of R x C; 64 x 64 tiles through LDS, both
// sides in 512-byte rows.  twN = exp(-2 pi i k / len); r*c < len.
template <int S, bool NT>
__global__ __launch_bounds__(256) void transpose_kernel(const cf *in, cf *out, int R, int C, int tr, int tc,
                                                         const cf *__restrict__ twN)
{
    __shared__ cf tile[64][65];
    const size_t t = blockIdx.x;
    const int tcx = (int)(t % tc);
    const size_t t2 = t / tc;
    const int trx = (int)(t2 % tr);
    const size_t f = t2 / tr;
    const int r0 = trx * 64, c0 = tcx * 64;
    const cf *src = in + f * (size_t)R * C;
    cf *dst = out + f * (size_t)R * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    cf v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int r = r0 + ty + 4 * k, c = c0 + tx;
        v[k] = (r < R && c < C) ? aeth::nt_load<NT>(src + (size_t)r * C + c) : mk(0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) tile[ty + 4 * k][tx] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int c = c0 + ty + 4 * k, r = r0 + tx;
        if (r < R && c < C) {
            cf w = tile[tx][ty + 4 * k];
            if constexpr (S != 0) w = ctw<S>(w, twN[(size_t)r * c]);
            aeth::nt_store<NT>(dst + (size_t)c * R + r, w);
        }
    }
}

template <int S>
int launch_transpose(const aeth_fft *plan, const float2 *in, float2 *out, size_t R, size_t C, size_t batch)
{
    const aeth_ctx *ctx = plan->ctx;
    const int tr = (int)((R + 63) / 64), tc = (int)((C + 63) / 64);
    const size_t tiles = batch * (size_t)tr * tc;
    if (tiles > 0x7fffffffull) return aeth::set_error(AETH_E_UNSUPPORTED, "fourstep_mixed: %zu tiles in one launch", tiles);
    const bool nt = aeth::streams_past_cache(2 * batch * R * C * sizeof(float2));
    if (nt) hipLaunchKernelGGL((transpose_kernel<S, true>), dim3((unsigned)tiles), dim3(256), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)out, (int)R, (int)C, tr, tc, (const cf *)plan->tw_dev);
    else    hipLaunchKernelGGL((transpose_kernel<S, false>), dim3((unsigned)tiles), dim3(256), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)out, (int)R, (int)C, tr, tc, (const cf *)plan->tw_dev);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

// ---- fourstep_mixed with a small first factor R: no transposes ----
// step A: the R-point transforms down the columns of the R x M frame, times W_len^(c k1); one lane per column,
// rows M elements apart, so every access is a coalesced row segment
template <int R, int S, bool NT>
__global__ __launch_bounds__(256) void smallcol_kernel(const cf *in, cf *out, size_t M, size_t cols,
                                                        const cf *__restrict__ twN)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;       // frame * M + column
    if (g >= cols) return;
    const size_t f = g / M, c = g - f * M;
    const cf *src = in + f * R * M + c;
    cf *dst = out + f * R * M + c;
    cf u[R];
#pragma unroll
    for (int r = 0; r < R; r++) u[r] = aeth::nt_load<NT>(src + r * M);
    Bfly<R, S>::run(u);
#pragma unroll
    for (int k = 1; k < R; k++) u[k] = ctw<S>(u[k], twN[c * k]);
#pragma unroll
    for (int k = 0; k < R; k++) aeth::nt_store<NT>(dst + k * M, u[k]);
}

// step C: X[k1 + R k2] = b[k1][k2].  Frame f, column c sits at g = f M + c and its R outputs at out[g R ...], so a
// block of 256 columns owns 256 R consecutive outputs: gathered through LDS (one pad slot per 16 elements), stored
// in whole rows
template <int R, bool NT>
__global__ __launch_bounds__(256) void interleave_kernel(const cf *in, cf *out, size_t M, size_t cols)
{
    __shared__ cf tile[256 * R + 256 * R / 16 + 1];
    const size_t g0 = (size_t)blockIdx.x * 256;
    const size_t g = g0 + threadIdx.x;                              // frame * M + k2
    const bool live = g < cols;
    const size_t f = live ? g / M : 0, c = live ? g - f * M : 0;
    const cf *src = in + f * R * M + c;
    cf u[R];
#pragma unroll
    for (int r = 0; r < R; r++) u[r] = live ? aeth::nt_load<NT>(src + r * M) : mk(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int e = (int)threadIdx.x * R + r;
        tile[e + (e >> 4)] = u[r];
    }
    __syncthreads();
    const size_t have = (cols - g0 < 256 ? cols - g0 : 256) * R;   // outputs this block owns
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int e = j * 256 + (int)threadIdx.x;
        if ((size_t)e < have) aeth::nt_store<NT>(out + g0 * R + e, tile[e + (e >> 4)]);
    }
}

template <int R>
int run_small_first_factor(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const aeth_ctx *ctx = plan->ctx;
    const size_t M = plan->n2, cols = batch * M;
    const unsigned blocks = (unsigned)((cols + 255) / 256);
    if ((cols + 255) / 256 > 0x7fffffffull) return aeth::set_error(AETH_E_UNSUPPORTED, "fourstep_mixed: batch too large");
    const bool nt = aeth::streams_past_cache(2 * batch * plan->len * sizeof(float2));
    float2 *a = plan->work_dev;
#define AETH_SC(SS, NN) hipLaunchKernelGGL((smallcol_kernel<R, SS, NN>), dim3(blocks), dim3(256), 0, aeth::ctx_stream(ctx), (const cf *)in, (cf *)a, M, cols, (const cf *)plan->tw_dev)
    if (sign > 0) { if (nt) AETH_SC(+1, true); else AETH_SC(+1, false); }
    else          { if (nt) AETH_SC(-1, true); else AETH_SC(-1, false); }
#undef AETH_SC
    AETH_HIP(hipGetLastError());
    int rc = aeth::fft_run(plan->sub2, a, a, batch * R, sign, scale);
    if (rc) return rc;
    if (nt) hipLaunchKernelGGL((interleave_kernel<R, true>), dim3(blocks), dim3(256), 0, aeth::ctx_stream(ctx), (const cf *)a, (cf *)out, M, cols);
    else    hipLaunchKernelGGL((interleave_kernel<R, false>), dim3(blocks), dim3(256), 0, aeth::ctx_stream(ctx), (const cf *)a, (cf *)out, M, cols);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

}  // namespace

namespace aeth {

// ------------------------------- four-step --------------------------------------------
int fft_plan_fourstep(aeth_fft *plan)
{
    int k = 0;
    while (((size_t)1 << k) < plan->len) k++;
    // The split (tools/vs_rocfft.py under AETH_4S_N1LOG, 32 Mi samples): step B scatters one 8-byte element per row, so
    // it needs many short rows per workgroup -- rows of 4096 points run at 0.6-0.8 TB/s, rows of 2048 and fewer at
    // 2-3 TB/s -- while step A loses its 128-byte row segments once a column needs more than 64 lanes (n1 > 1024).
    // So: 256 columns up to 2^18, rows of 2048 from 2^19 (2^19 as 256 x 2048: 2.18 -> 2.48 TB/s, 2^20 as 512 x 2048:
    // 1.73 -> 2.13 TB/s against the square split, 2^23 as 4096 x 2048: 1052 -> 661 us); 2^24 has no such split and
    // goes through the transposing path instead (aeth_fft_create; 1220 -> 975 us).
    plan->n1 = (size_t)1 << (k / 2);
    if (k >= 16 && k <= 18) plan->n1 = 256;
    else if (k >= 19 && k <= 23) plan->n1 = (size_t)1 << (k - 11);
    // deep form (fft_run_fourstep): n1 <= 256 columns over rows of 65536 points that are four-step transforms themselves
    const int deep_from = aeth::lab_int("AETH_4S_DEEP_FROM", 23);
    if (k >= deep_from && k >= 21) plan->n1 = (size_t)1 << (k - 16);
    const int n1log = aeth::lab_int("AETH_4S_N1LOG", 0);
    if (n1log >= 4 && n1log < k && k - n1log <= 12) plan->n1 = (size_t)1 << n1log;
    plan->n2 = plan->len / plan->n1;
    if (plan->n1 < 16 || (plan->n2 > 4096 && plan->n2 != 65536)) return set_error(AETH_E_UNSUPPORTED, "fourstep_pow2: length %zu", plan->len);
    int rc = aeth_fft_create(plan->ctx, plan->n1, 1, &plan->sub1);
    if (rc) return rc;
    return aeth_fft_create(plan->ctx, plan->n2, 1, &plan->sub2);
}

int fft_run_fourstep(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    // Measured and not kept (tools/tune_4step.py, AETH_4S_GROUP_MIB; profiles/r02_c5.json): running steps A and B
    // group by group through a work buffer of one cache-sized group, so that the intermediate is overwritten in the
    // Infinity Cache instead of travelling to HBM -- 512 x 65536: 176 us as two launches, 185 / 202 / 222 / 308 us
    // with groups of 128 / 64 / 32 / 16 MiB.  Each of the two launches already streams at copy speed (537 MB in
    // 88 us) and part of the intermediate is served from the cache as it is; more launches only add their gaps.
    // The knob stays for the tuning tool: 0 = one group (default).
    const size_t gmib = (size_t)aeth::lab_int("AETH_4S_GROUP_MIB", 0);
    size_t gf = gmib ? gmib * ((size_t)1 << 20) / (plan->len * sizeof(float2)) : batch;
    if (gf < 1) gf = 1;
    if (gf > batch) gf = batch;
    // (round 3 also measured frame halves on two HIP streams -- one half's step B beside the other's step A: 181.9 us
    // against 170.6 for 512 frames, 46.8 against 30.7 for 64, profiles/r03_c5.json; that shape is not in the library)
    int rc = ensure_work(plan, plan->len * gf);
    if (rc) return rc;
    for (size_t g0 = 0; g0 < batch; g0 += gf) {
        const size_t cnt = batch - g0 < gf ? batch - g0 : gf;
        const float2 *gin = in + g0 * plan->len;
        float2 *gout = out + g0 * plan->len;
#define AETH_BODY(NN)                                                                                   \
    return sign > 0 ? launch_cols<typename CfgFor<NN>::type, +1>(plan, gin, cnt, batch)                 \
                    : launch_cols<typename CfgFor<NN>::type, -1>(plan, gin, cnt, batch)
        auto cols = [&]() -> int { AETH_POW2_SWITCH(plan->n1, AETH_BODY, return set_error(AETH_E_UNSUPPORTED, "n1")) };
#undef AETH_BODY
#define AETH_BODY(NN)                                                                                   \
    return sign > 0 ? launch_rows<typename CfgFor<NN>::type, +1>(plan, gout, cnt, scale, batch)         \
                    : launch_rows<typename CfgFor<NN>::type, -1>(plan, gout, cnt, scale, batch)
        auto rows = [&]() -> int { AETH_POW2_SWITCH(plan->n2, AETH_BODY, return set_error(AETH_E_UNSUPPORTED, "n2")) };
#undef AETH_BODY
        rc = cols();
        if (rc) return rc;
        if (plan->n2 > 4096) {
            // deep form: the n1 rows of a frame are n2-point transforms of their own (natural order, in place in the
            // work buffer, two launches), and X[k1 + n1 k2] is their transpose -- four launches of 16 B/sample each,
            // every one with 256-byte row segments, where the two-launch form would store 16-byte segments
            rc = fft_run(plan->sub2, plan->work_dev, plan->work_dev, cnt * plan->n1, sign, scale);
            if (rc) return rc;
            rc = launch_transpose<0>(plan, plan->work_dev, gout, plan->n1, plan->n2, cnt);
            if (rc) return rc;
            continue;
        }
        rc = rows();
        if (rc) return rc;
    }
    return AETH_OK;
}

// ------------------------------- four-step, any two factors ----------------------------
// len = n1 * n2 with both factors served by a single-workgroup kernel (n1, n2 <= 8192).  Five launches:
//   x as n1 x n2 -> transpose -> n2 rows of n1: transform each -> times W_len^(n2 k1), transpose -> n1 rows of n2:
//   transform each (scale fused) -> transpose: X[k1 + n1 k2].
// 80 B/sample of traffic against Bluestein's two power-of-two four-step transforms of 2-4x the length.
bool fourstep_small_factor(size_t r)
{
    switch (r) { case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 12: case 15: case 16: return true; default: return false; }
}

int fft_plan_fourstep_mixed(aeth_fft *plan)
{
    if (!fourstep_small_factor(plan->n1)) {
        int rc = aeth_fft_create(plan->ctx, plan->n1, 1, &plan->sub1);
        if (rc) return rc;
    }
    return aeth_fft_create(plan->ctx, plan->n2, 1, &plan->sub2);
}

int fft_run_fourstep_mixed(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const size_t n1 = plan->n1, n2 = plan->n2, total = plan->len * batch;
    if (fourstep_small_factor(n1)) {
        // small first factor: columns in registers, no transposes -- three launches, 48 B/sample
        int rc = ensure_work(plan, total);
        if (rc) return rc;
        switch (n1) {
        case 2: return run_small_first_factor<2>(plan, in, out, batch, sign, scale);
        case 3: return run_small_first_factor<3>(plan, in, out, batch, sign, scale);
        case 4: return run_small_first_factor<4>(plan, in, out, batch, sign, scale);
        case 5: return run_small_first_factor<5>(plan, in, out, batch, sign, scale);
        case 6: return run_small_first_factor<6>(plan, in, out, batch, sign, scale);
        case 7: return run_small_first_factor<7>(plan, in, out, batch, sign, scale);
        case 8: return run_small_first_factor<8>(plan, in, out, batch, sign, scale);
        case 9: return run_small_first_factor<9>(plan, in, out, batch, sign, scale);
        case 10: return run_small_first_factor<10>(plan, in, out, batch, sign, scale);
        case 12: return run_small_first_factor<12>(plan, in, out, batch, sign, scale);
        case 15: return run_small_first_factor<15>(plan, in, out, batch, sign, scale);
        default: return run_small_first_factor<16>(plan, in, out, batch, sign, scale);
        }
    }
    int rc = ensure_work(plan, 2 * total);
    if (rc) return rc;
    float2 *a = plan->work_dev, *b = plan->work_dev + total;
    rc = launch_transpose<0>(plan, in, a, n1, n2, batch);                       // a[n2][n1]
    if (rc) return rc;
    rc = fft_run(plan->sub1, a, a, batch * n2, sign, 1.0f);                      // a[n2][k1]
    if (rc) return rc;
    rc = sign > 0 ? launch_transpose<+1>(plan, a, b, n2, n1, batch)             // b[k1][n2] = a[n2][k1] W^(n2 k1)
                  : launch_transpose<-1>(plan, a, b, n2, n1, batch);
    if (rc) return rc;
    rc = fft_run(plan->sub2, b, b, batch * n1, sign, scale);                    // b[k1][k2]
    if (rc) return rc;
    return launch_transpose<0>(plan, b, out, n1, n2, batch);                    // out[k2][k1]
}

// ------------------------------- bluestein --------------------------------------------
int fft_plan_bluestein(aeth_fft *plan)
{
    const size_t N = plan->len;
    size_t M = 1;
    while (M < 2 * N - 1) M <<= 1;
    // Past the one-launch kernel (M <= 4096) the convolution length need not be a power of two: the nearest length at
    // or above 2N-1 that one register-resident launch transforms (N = 4099: 8640 = 2^6 3^3 5 instead of 16384, half the
    // bytes through each of the five launches).
    if (M > 4096 && aeth::lab_int("AETH_BLU_ANY_M", 1)) {
        const size_t table_top = 20480;                    // the largest length the ragged table holds
        for (size_t m = 2 * N - 1; m < M && m <= table_top; m++)
            if (aeth::fft_ragged_supported(m)) { M = m; break; }
    }
    plan->blu_m = M;
    int rc = aeth_fft_create(plan->ctx, M, 1, &plan->blu_sub);
    if (rc) return rc;
    // chirp[k] = exp(-j pi k^2 / N), k^2 reduced mod 2N in integers to keep the angle exact
    std::vector<float2> chirp(N), filt(M, make_float2(0.f, 0.f));
    for (size_t k = 0; k < N; k++) {
        unsigned long long q = ((unsigned long long)k * k) % (2ull * N);
        double ang = -M_PI * (double)q / (double)N;
        chirp[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    // filter b[n] = conj(chirp[|n|]) wrapped to length M
    for (size_t k = 0; k < N; k++) {
        float2 c = make_float2(chirp[k].x, -chirp[k].y);
        filt[k] = c;
        if (k) filt[M - k] = c;
    }
    AETH_HIP(hipMalloc((void **)&plan->blu_chirp, N * sizeof(float2)));
    AETH_HIP(hipMalloc((void **)&plan->blu_filt, M * sizeof(float2)));
    rc = aeth_upload(plan->ctx, plan->blu_chirp, chirp.data(), N * sizeof(float2));
    if (rc) return rc;
    rc = aeth_upload(plan->ctx, plan->blu_filt, filt.data(), M * sizeof(float2));
    if (rc) return rc;
    // Bf = DFT-(b) / M : the 1/M of the inverse transform is folded in (exact when M is a power of two)
    rc = fft_run(plan->blu_sub, plan->blu_filt, plan->blu_filt, 1, -1, 1.0f / (float)M);
    if (rc) return rc;
    return aeth_ctx_sync(plan->ctx);
}

int fft_run_bluestein(aeth_fft *plan, const float2 *in, float2 *out, size_t batch, int sign, float scale)
{
    const size_t N = plan->len, M = plan->blu_m;
    if (plan->blu_sub->algo == FFT_ALGO_POW2 && M <= 4096 && N * batch < 0x7fffffffull)
        // the whole chirp-z chain in one launch; DFT+(x) = conj(DFT-(conj x)) by conjugating on the way in and out
        return fmi_bluestein(plan->blu_sub, in, out, N, batch, plan->blu_chirp, plan->blu_filt, sign > 0 ? 1 : 0, scale);
    int rc = ensure_work(plan, M * batch);
    if (rc) return rc;
    aeth_ctx *ctx = plan->ctx;
    cf *a = (cf *)plan->work_dev;
    const dim3 b(kBlock);
    // DFT+(x) = conj(DFT-(conj x))
    if (sign > 0) hipLaunchKernelGGL((blu_pre<true>), dim3(grid_for(ctx, M * batch)), b, 0, aeth::ctx_stream(ctx), (const cf *)in, (const cf *)plan->blu_chirp, a, N, M, batch);
    else          hipLaunchKernelGGL((blu_pre<false>), dim3(grid_for(ctx, M * batch)), b, 0, aeth::ctx_stream(ctx), (const cf *)in, (const cf *)plan->blu_chirp, a, N, M, batch);
    AETH_HIP(hipGetLastError());
    if (plan->blu_sub->algo == FFT_ALGO_POW2 && M <= 4096) {
        // circular convolution in ONE launch of the fused transform * filter * inverse kernel.  It runs +j then
        // -j where the three-launch form runs -j then +j; the filter is symmetric (b[n] = b[M-n]), so its
        // spectrum is the same under both signs and the convolution comes out identical in exact arithmetic.
        rc = aeth_fft_mul_ifft(plan->blu_sub, (aeth_cf32 *)plan->work_dev, M * batch, batch, (const aeth_cf32 *)plan->blu_filt,
                               M, AETH_SCALE_NONE, 0.f, AETH_SCALE_NONE, 0.f);
        if (rc) return rc;
    } else {
        rc = fft_run(plan->blu_sub, plan->work_dev, plan->work_dev, batch, -1, 1.0f);
        if (rc) return rc;
        hipLaunchKernelGGL(blu_mul, dim3(grid_for(ctx, M * batch)), b, 0, aeth::ctx_stream(ctx), a, (const cf *)plan->blu_filt, M, batch);
        AETH_HIP(hipGetLastError());
        rc = fft_run(plan->blu_sub, plan->work_dev, plan->work_dev, batch, +1, 1.0f);
        if (rc) return rc;
    }
    if (sign > 0) hipLaunchKernelGGL((blu_post<true>), dim3(grid_for(ctx, N * batch)), b, 0, aeth::ctx_stream(ctx), (const cf *)a, (const cf *)plan->blu_chirp, (cf *)out, N, M, batch, scale);
    else          hipLaunchKernelGGL((blu_post<false>), dim3(grid_for(ctx, N * batch)), b, 0, aeth::ctx_stream(ctx), (const cf *)a, (const cf *)plan->blu_chirp, (cf *)out, N, M, batch, scale);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

void fft_plan_release_children(aeth_fft *plan)
{
    if (plan->sub1) { aeth_fft_destroy(plan->sub1); plan->sub1 = nullptr; }
    if (plan->sub2) { aeth_fft_destroy(plan->sub2); plan->sub2 = nullptr; }
    if (plan->blu_sub) { aeth_fft_destroy(plan->blu_sub); plan->blu_sub = nullptr; }
    if (plan->work_dev) { (void)hipFree(plan->work_dev); plan->work_dev = nullptr; plan->work_elems = 0; }
    if (plan->blu_chirp) { (void)hipFree(plan->blu_chirp); plan->blu_chirp = nullptr; }
    if (plan->blu_filt) { (void)hipFree(plan->blu_filt); plan->blu_filt = nullptr; }
}

}  // namespace aeth
