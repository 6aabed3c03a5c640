// aeth_vecops.hip -- element-wise cf32 kernels behind trait VecOps
// (reference: src/vecops.rs:94-177).  Compiled with -ffp-contract=off: the
// reference is Rust, which never fuses a*b+c, and these ops must be bit-exact
// against it (the crate's assert_evm! at -80 "dB" is tighter than 1 ulp,
// src/lib.rs:36-47).
//
// HBM-bound streaming kernels: 16-byte (2 x cf32) accesses per lane where
// alignment allows, 8-byte otherwise.  One access per lane and one 4 KiB tile per
// 256-lane workgroup, the grid covering the whole vector: measured on MI355X
// (tools/ew_bw.hip, a[i] += b[i] over 256 MiB operands) this reaches 5.9 TB/s, against
// 5.3-5.5 TB/s for per-workgroup tiles inside a capped persistent grid and 3.7 TB/s for
// a grid-stride loop whose unrolled accesses are a whole grid apart -- the hardware
// dispatcher orders tiles better than a software loop, and HBM pages/TLB entries stay hot.
#include "aeth_internal.h"

namespace {

enum Op { OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_SCALE, OP_CONJ, OP_CLONE, OP_ZERO };

constexpr int kBlock = 256;

// ---- per-element arithmetic, spelled exactly as num-complex 0.2 does ---------
template <int OP>
__device__ __forceinline__ float2 apply2(float2 a, float2 b, float s)
{
    if constexpr (OP == OP_ADD) return make_float2(a.x + b.x, a.y + b.y);          // vecops.rs:140
    if constexpr (OP == OP_SUB) return make_float2(a.x - b.x, a.y - b.y);          // vecops.rs:153
    if constexpr (OP == OP_MUL)                                                     // vecops.rs:110
        return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
    if constexpr (OP == OP_DIV) {                                                   // vecops.rs:123
        float ns = b.x * b.x + b.y * b.y;
        float re = a.x * b.x + a.y * b.y;
        float im = a.y * b.x - a.x * b.y;
        return make_float2(re / ns, im / ns);
    }
    if constexpr (OP == OP_SCALE) return make_float2(a.x * s, a.y * s);             // vecops.rs:95
    if constexpr (OP == OP_CONJ) return make_float2(a.x, -a.y);                     // vecops.rs:128
    if constexpr (OP == OP_CLONE) return b;                                         // vecops.rs:170
    return make_float2(0.0f, 0.0f);                                                 // vecops.rs:175
}

template <int OP>
__device__ __forceinline__ float4 apply4(float4 a, float4 b, float s)
{
    float2 lo = apply2<OP>(make_float2(a.x, a.y), make_float2(b.x, b.y), s);
    float2 hi = apply2<OP>(make_float2(a.z, a.w), make_float2(b.z, b.w), s);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

template <int OP> constexpr bool reads_self() { return OP != OP_CLONE && OP != OP_ZERO; }
template <int OP> constexpr bool reads_other() { return OP == OP_ADD || OP == OP_SUB || OP == OP_MUL || OP == OP_DIV || OP == OP_CLONE; }

// V = float4 (two samples) or float2 (one sample)
template <int OP, typename V, bool NT>
__global__ __launch_bounds__(kBlock) void ew_kernel(V *__restrict__ self, const V *__restrict__ other,
                                                    size_t n, float s)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // NT: every element is touched exactly once and the operands exceed the cache (aeth_internal.h)
    V a, b;
    if constexpr (reads_self<OP>()) a = aeth::nt_load<NT>(self + i);
    if constexpr (reads_other<OP>()) b = aeth::nt_load<NT>(other + i);
    V r;
    if constexpr (sizeof(V) == 16) r = apply4<OP>(a, b, s);
    else r = apply2<OP>(a, b, s);
    aeth::nt_store<NT>(self + i, r);
}

inline unsigned grid_for(const aeth_ctx *, size_t items)
{
    size_t blocks = (items + kBlock - 1) / kBlock;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

template <int OP>
int launch_ew(aeth_ctx *ctx, aeth_cf32 *self, const aeth_cf32 *other, size_t n, float s)
{
    if (n == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    float2 *a = reinterpret_cast<float2 *>(self);
    const float2 *b = reinterpret_cast<const float2 *>(other);
    const bool two = reads_other<OP>();
    const uintptr_t ma = reinterpret_cast<uintptr_t>(a) & 15u;
    const uintptr_t mb = two ? (reinterpret_cast<uintptr_t>(b) & 15u) : ma;
    const bool nt = aeth::streams_past_cache(n * sizeof(float2) * ((reads_self<OP>() ? 2 : 1) + (two ? 1 : 0)));
#define AETH_EW(VV, GRID, A, B, N)                                                                                      \
    do {                                                                                                                \
        if (nt) hipLaunchKernelGGL((ew_kernel<OP, VV, true>), GRID, dim3(kBlock), 0, aeth::ctx_stream(ctx), A, B, N, s);          \
        else hipLaunchKernelGGL((ew_kernel<OP, VV, false>), GRID, dim3(kBlock), 0, aeth::ctx_stream(ctx), A, B, N, s);            \
    } while (0)
    if (ma == mb) {
        // same phase: peel one sample if the base sits on an odd 8-byte slot, then 16-byte body
        size_t head = (ma != 0 && n > 0) ? 1 : 0;
        size_t body = (n - head) / 2;
        size_t tail = (n - head) - 2 * body;
        if (head) AETH_EW(float2, dim3(1), a, b, (size_t)1);
        if (body)
            AETH_EW(float4, dim3(grid_for(ctx, body)), reinterpret_cast<float4 *>(a + head),
                    reinterpret_cast<const float4 *>(two ? b + head : nullptr), body);
        if (tail) AETH_EW(float2, dim3(1), a + head + 2 * body, two ? b + head + 2 * body : nullptr, (size_t)1);
    } else {
        AETH_EW(float2, dim3(grid_for(ctx, n)), a, b, n);
    }
#undef AETH_EW
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

// ---- frames[f][j] *= sig[j]: vec_mul with one operand shared by every frame ------------------------------
// (the middle of `c.vec_rfft(fft, s).vec_mul(&sig).vec_rifft(fft, s)` over chunks_mut(fft_len), benches.rs:410-416
// and util/plot.rs:59-61).  One launch for any number of frames: grid.x covers a frame, grid.y walks the frames.
// One lane per sample (or pair of samples) of the WHOLE batch, the position inside the frame by a multiply-high
// division: full wavefronts whatever the frame length (round 1's form gave every frame its own row of workgroups --
// a 100-sample frame kept 100 of 256 lanes busy: 4.5 TB/s; this form: one 8- or 16-byte access per lane like the
// other element-wise kernels).
template <typename V, bool NT>
__global__ __launch_bounds__(kBlock) void mul_frames_kernel(V *__restrict__ frames, const float2 *__restrict__ sig,
                                                            aeth::FastDiv flen, size_t items)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;             // item = sizeof(V) / 8 samples
    if (i >= items) return;
    constexpr unsigned S = sizeof(V) / sizeof(float2);
    // sample index modulo the frame length; 32-bit arithmetic on a per-chunk basis (the host side launches chunks of
    // at most 2^31 samples that start on a frame boundary)
    const unsigned s0 = (unsigned)i * S;
    const unsigned j = s0 - aeth::fdiv(s0, flen) * flen.d;
    V a = aeth::nt_load<NT>(frames + i);
    if constexpr (S == 2) {
        const float4 b = *reinterpret_cast<const float4 *>(sig + j);      // frame length even, j even: 16-byte aligned, never wraps
        aeth::nt_store<NT>(frames + i, apply4<OP_MUL>(a, b, 0.f));
    } else aeth::nt_store<NT>(frames + i, apply2<OP_MUL>(a, sig[j], 0.f));
}

int launch_mul_frames(aeth_ctx *ctx, aeth_cf32 *frames, size_t frame_len, size_t batch, const aeth_cf32 *sig)
{
    if (frame_len == 0 || batch == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    const bool nt = aeth::streams_past_cache(2 * batch * frame_len * sizeof(float2));
    float2 *f = reinterpret_cast<float2 *>(frames);
    const float2 *sg = reinterpret_cast<const float2 *>(sig);
    const bool wide = frame_len % 2 == 0 && aeth::aligned16(frames) && aeth::aligned16(sig);
    const aeth::FastDiv fd = aeth::make_fastdiv((uint32_t)frame_len);
    AETH_REQUIRE(frame_len < ((size_t)1 << 31), AETH_E_UNSUPPORTED, "frame length %zu", frame_len);
    // chunks of whole frames, under 2^31 samples each
    const size_t per = (((size_t)1 << 31) - 1) / frame_len;
    for (size_t f0 = 0; f0 < batch; f0 += per) {
        const size_t nf = batch - f0 < per ? batch - f0 : per;
        const size_t samples = nf * frame_len, items = wide ? samples / 2 : samples;
        const dim3 grid(grid_for(ctx, items));
        float2 *base = f + f0 * frame_len;
        if (wide) {
            auto k = nt ? mul_frames_kernel<float4, true> : mul_frames_kernel<float4, false>;
            hipLaunchKernelGGL(k, grid, dim3(kBlock), 0, aeth::ctx_stream(ctx), reinterpret_cast<float4 *>(base), sg, fd, items);
        } else {
            auto k = nt ? mul_frames_kernel<float2, true> : mul_frames_kernel<float2, false>;
            hipLaunchKernelGGL(k, grid, dim3(kBlock), 0, aeth::ctx_stream(ctx), base, sg, fd, items);
        }
    }
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

// ---- mirror: swap(x, x+mid), mid = len/2, per frame (vecops.rs:157-161) --------
// K items per lane (a wave owns K * 64 consecutive items, every access of a wave contiguous, all loads first): ONE
// long frame -- two streams half a vector apart -- gains from four (93 -> 86 us for 2^25 samples), many short frames
// lose (frames of 2048: 83 -> 87 us), so the launch picks K by the batch
template <typename V, bool NT, int kMirrorItems>
__global__ __launch_bounds__(kBlock) void mirror_kernel(V *__restrict__ x, size_t frame_stride_v, size_t mid_v,
                                                        size_t batch)
{
    // items = batch * mid_v, item -> (frame, j); kMirrorItems items per lane, a wave's accesses contiguous, loads first
    const size_t i0 = ((size_t)blockIdx.x * kBlock + (threadIdx.x & ~63u)) * kMirrorItems + (threadIdx.x & 63u);
    V lo[kMirrorItems], hi[kMirrorItems];
    V *p[kMirrorItems];
#pragma unroll
    for (int k = 0; k < kMirrorItems; k++) {
        const size_t i = i0 + (size_t)k * 64;
        if (i < batch * mid_v) {
            size_t f = i / mid_v, j = i - f * mid_v;
            p[k] = x + f * frame_stride_v + j;
            lo[k] = aeth::nt_load<NT>(p[k]); hi[k] = aeth::nt_load<NT>(p[k] + mid_v);
        }
    }
#pragma unroll
    for (int k = 0; k < kMirrorItems; k++) {
        const size_t i = i0 + (size_t)k * 64;
        if (i >= batch * mid_v) return;
        aeth::nt_store<NT>(p[k], hi[k]);
        aeth::nt_store<NT>(p[k] + mid_v, lo[k]);
    }
}

int launch_mirror(aeth_ctx *ctx, aeth_cf32 *self, size_t frame_len, size_t batch)
{
    const size_t mid = frame_len / 2;
    if (mid == 0 || batch == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    float2 *x = reinterpret_cast<float2 *>(self);
    const bool vec = aeth::aligned16(x) && (mid % 2 == 0) && (frame_len % 2 == 0);
    const bool nt = aeth::streams_past_cache(2 * batch * frame_len * sizeof(float2));
    const bool four = batch == 1 && mid >= ((size_t)1 << 16);
#define AETH_MIRROR(VV, KK, TOTAL, ...)                                                                                          \
    do {                                                                                                                         \
        auto k = nt ? mirror_kernel<VV, true, KK> : mirror_kernel<VV, false, KK>;                                                \
        hipLaunchKernelGGL(k, dim3(grid_for(ctx, ((TOTAL) + (KK) - 1) / (KK))), dim3(kBlock), 0, aeth::ctx_stream(ctx), __VA_ARGS__); \
    } while (0)
    if (vec) {
        const size_t total = batch * (mid / 2);
        if (four) AETH_MIRROR(float4, 4, total, reinterpret_cast<float4 *>(x), frame_len / 2, mid / 2, batch);
        else AETH_MIRROR(float4, 1, total, reinterpret_cast<float4 *>(x), frame_len / 2, mid / 2, batch);
    } else {
        const size_t total = batch * mid;
        if (four) AETH_MIRROR(float2, 4, total, x, frame_len, mid, batch);
        else AETH_MIRROR(float2, 1, total, x, frame_len, mid, batch);
    }
#undef AETH_MIRROR
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int check_unary(aeth_ctx *ctx, const void *self, size_t n)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(self || n == 0, AETH_E_ARG, "self is null");
    AETH_REQUIRE(aeth::aligned8(self), AETH_E_ALIGN, "self not 8-byte aligned");
    return AETH_OK;
}

int check_binary(aeth_ctx *ctx, const void *self, size_t n, const void *other, size_t n_other)
{
    int rc = check_unary(ctx, self, n);
    if (rc) return rc;
    AETH_REQUIRE(n == n_other, AETH_E_LEN, AETH_MSG_VEC_LEN);
    AETH_REQUIRE(other || n == 0, AETH_E_ARG, "other is null");
    AETH_REQUIRE(aeth::aligned8(other), AETH_E_ALIGN, "other not 8-byte aligned");
    return AETH_OK;
}

// host-slice flavour, synchronous (the literal trait call): small slices through the context's pinned bounce buffers
// (one launch on host memory, one wait), large ones H2D -> kernel -> D2H (aeth::HostIO)
template <typename F>
int host_roundtrip(aeth_ctx *ctx, aeth_cf32 *self, size_t n, const aeth_cf32 *other, bool upload_self, F &&run)
{
    if (n == 0) return AETH_OK;
    const size_t bytes = n * sizeof(aeth_cf32);
    aeth::DeviceGuard dev_guard(ctx->device);
    aeth::HostIO io;
    int rc = io.open(ctx, bytes, other ? bytes : 0);
    if (rc) return rc;
    if (upload_self) { rc = io.put(0, self, bytes); if (rc) return rc; }
    if (other) { rc = io.put(1, other, bytes); if (rc) return rc; }
    rc = run((aeth_cf32 *)io.buf[0], (const aeth_cf32 *)io.buf[1]);
    if (rc) return rc;
    return io.get(self, 0, bytes);
}

// ---- a chain of element-wise steps in ONE pass over memory ------------------------------------------------------
// The reference's VecOps are chainable (`v.vec_add(&a).vec_mul(&b).vec_conj()`, vecops.rs:12-38 and BASELINE config
// 1): on the CPU every link is a pass over cache-resident data, on the device every link is a launch and a round trip
// through HBM (C1: three launches, 24 + 24 + 16 = 64 B per sample).  aeth_vec_chain runs the links per element in
// registers: `self` is read once and written once, every binary link reads its operand once (C1: 8 + 8 + 8 R + 8 W =
// 32 B per sample, one launch).  Each link is the same rounded arithmetic as its own kernel (apply2 above, no
// contraction), so the result is bit-identical to the separate calls.
constexpr int kChainMax = 8;
struct ChainArgs {
    int n_steps;
    int op[kChainMax];
    const float2 *other[kChainMax];
    float scale[kChainMax];
};

__device__ __forceinline__ float2 chain_step(int op, float2 a, float2 b, float s)
{
    switch (op) {                                           // wave-uniform: the program lives in the kernel arguments
    case OP_ADD: return apply2<OP_ADD>(a, b, s);
    case OP_SUB: return apply2<OP_SUB>(a, b, s);
    case OP_MUL: return apply2<OP_MUL>(a, b, s);
    case OP_DIV: return apply2<OP_DIV>(a, b, s);
    case OP_SCALE: return apply2<OP_SCALE>(a, b, s);
    case OP_CONJ: return apply2<OP_CONJ>(a, b, s);
    case OP_CLONE: return b;
    default: return make_float2(0.0f, 0.0f);
    }
}

template <typename V, bool NT>
__global__ __launch_bounds__(kBlock) void chain_kernel(V *__restrict__ self, size_t n, ChainArgs c, int read_self)
{
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    V b[kChainMax];
    V a;
    // every load of the element goes out before the first link runs
    if (read_self) a = aeth::nt_load<NT>(self + i);
    else { if constexpr (sizeof(V) == 16) a = make_float4(0.f, 0.f, 0.f, 0.f); else a = make_float2(0.f, 0.f); }
#pragma unroll
    for (int k = 0; k < kChainMax; k++) {
        if (k < c.n_steps && c.other[k]) b[k] = aeth::nt_load<NT>(reinterpret_cast<const V *>(c.other[k]) + i);
        else { if constexpr (sizeof(V) == 16) b[k] = make_float4(0.f, 0.f, 0.f, 0.f); else b[k] = make_float2(0.f, 0.f); }
    }
#pragma unroll
    for (int k = 0; k < kChainMax; k++) {
        if (k < c.n_steps) {
            if constexpr (sizeof(V) == 16) {
                const float2 lo = chain_step(c.op[k], make_float2(a.x, a.y), make_float2(b[k].x, b[k].y), c.scale[k]);
                const float2 hi = chain_step(c.op[k], make_float2(a.z, a.w), make_float2(b[k].z, b[k].w), c.scale[k]);
                a = make_float4(lo.x, lo.y, hi.x, hi.y);
            } else a = chain_step(c.op[k], a, b[k], c.scale[k]);
        }
    }
    aeth::nt_store<NT>(self + i, a);
}

int launch_chain(aeth_ctx *ctx, float2 *self, size_t n, const ChainArgs &c)
{
    aeth::DeviceGuard dev_guard(ctx->device);
    const int read_self = !(c.op[0] == OP_CLONE || c.op[0] == OP_ZERO);
    size_t operands = 1 + (read_self ? 1 : 0);
    bool same_phase = true;
    const uintptr_t ma = reinterpret_cast<uintptr_t>(self) & 15u;
    for (int k = 0; k < c.n_steps; k++)
        if (c.other[k]) { operands++; same_phase = same_phase && (reinterpret_cast<uintptr_t>(c.other[k]) & 15u) == ma; }
    const bool nt = aeth::streams_past_cache(n * sizeof(float2) * operands);
    auto shifted = [&](size_t off) { ChainArgs d = c; for (int k = 0; k < c.n_steps; k++) if (d.other[k]) d.other[k] += off; return d; };
    auto one = [&](size_t off, size_t cnt) {                                            // 8-byte lanes
        auto kern = nt ? chain_kernel<float2, true> : chain_kernel<float2, false>;
        hipLaunchKernelGGL(kern, dim3(grid_for(ctx, cnt)), dim3(kBlock), 0, aeth::ctx_stream(ctx), self + off, cnt, shifted(off), read_self);
    };
    if (same_phase) {
        const size_t head = (ma != 0) ? 1 : 0, body = (n - head) / 2, tail = (n - head) - 2 * body;
        if (head) one(0, 1);
        if (body) {
            auto kern = nt ? chain_kernel<float4, true> : chain_kernel<float4, false>;
            hipLaunchKernelGGL(kern, dim3(grid_for(ctx, body)), dim3(kBlock), 0, aeth::ctx_stream(ctx), reinterpret_cast<float4 *>(self + head), body,
                               shifted(head), read_self);
        }
        if (tail) one(head + 2 * body, 1);
    } else one(0, n);
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

}  // namespace

extern "C" {

/* `self.vec_a(..).vec_b(..) ...` (vecops.rs:12-38) as one pass: see chain_kernel above */
int aeth_vec_chain(aeth_ctx *ctx, aeth_cf32 *self_, size_t n, const aeth_vec_step *steps, size_t n_steps)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(steps || n_steps == 0, AETH_E_ARG, "steps is null");
    static const int kOps[8] = {OP_SCALE, OP_MUL, OP_DIV, OP_CONJ, OP_ADD, OP_SUB, OP_CLONE, OP_ZERO};      /* AETH_VEC_* order */
    // every link's own checks first (the reference would panic at that link: vecops.rs:100-104 etc.)
    for (size_t k = 0; k < n_steps; k++) {
        const aeth_vec_step &st = steps[k];
        AETH_REQUIRE(st.op >= 0 && st.op < 8, AETH_E_ARG, "step %zu: unknown op %d", k, st.op);
        const int op = kOps[st.op];
        if (op == OP_ADD || op == OP_SUB || op == OP_MUL || op == OP_DIV || op == OP_CLONE) {
            AETH_REQUIRE(st.n_other == n, AETH_E_LEN, AETH_MSG_VEC_LEN);
            if (n) {
                AETH_REQUIRE(st.other_dev, AETH_E_ARG, "step %zu: null operand", k);
                AETH_REQUIRE(aeth::aligned8(st.other_dev), AETH_E_ALIGN, "pointer not 8-byte aligned");
                // a link reads its operand as it was BEFORE the chain: an operand that overlaps `self` would see the
                // original samples where the separate calls see modified ones (Rust cannot express that call anyway)
                const uintptr_t a0 = (uintptr_t)self_, a1 = a0 + n * sizeof(aeth_cf32), b0 = (uintptr_t)st.other_dev, b1 = b0 + n * sizeof(aeth_cf32);
                AETH_REQUIRE(a1 <= b0 || b1 <= a0, AETH_E_ARG, "step %zu: the operand overlaps self", k);
            }
        }
    }
    if (n == 0 || n_steps == 0) return AETH_OK;
    AETH_REQUIRE(self_, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(self_), AETH_E_ALIGN, "pointer not 8-byte aligned");
    for (size_t k0 = 0; k0 < n_steps; k0 += kChainMax) {                  /* longer chains: kChainMax links per pass */
        ChainArgs c;
        c.n_steps = (int)(n_steps - k0 < (size_t)kChainMax ? n_steps - k0 : (size_t)kChainMax);
        for (int k = 0; k < kChainMax; k++) { c.op[k] = OP_ZERO; c.other[k] = nullptr; c.scale[k] = 0.f; }
        for (int k = 0; k < c.n_steps; k++) {
            const aeth_vec_step &st = steps[k0 + k];
            c.op[k] = kOps[st.op];
            const bool bin = c.op[k] == OP_ADD || c.op[k] == OP_SUB || c.op[k] == OP_MUL || c.op[k] == OP_DIV || c.op[k] == OP_CLONE;
            c.other[k] = bin ? reinterpret_cast<const float2 *>(st.other_dev) : nullptr;
            c.scale[k] = st.scale;
        }
        int rc = launch_chain(ctx, reinterpret_cast<float2 *>(self_), n, c);
        if (rc) return rc;
    }
    return AETH_OK;
}

int aeth_vec_scale(aeth_ctx *ctx, aeth_cf32 *x, size_t n, float s)
{
    int rc = check_unary(ctx, x, n); if (rc) return rc;
    return launch_ew<OP_SCALE>(ctx, x, nullptr, n, s);
}
int aeth_vec_conj(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    int rc = check_unary(ctx, x, n); if (rc) return rc;
    return launch_ew<OP_CONJ>(ctx, x, nullptr, n, 0.f);
}
int aeth_vec_zero(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    int rc = check_unary(ctx, x, n); if (rc) return rc;
    return launch_ew<OP_ZERO>(ctx, x, nullptr, n, 0.f);
}
int aeth_vec_mirror(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    int rc = check_unary(ctx, x, n); if (rc) return rc;
    return launch_mirror(ctx, x, n, 1);
}
int aeth_vec_mirror_frames(aeth_ctx *ctx, aeth_cf32 *x, size_t frame_len, size_t batch)
{
    int rc = check_unary(ctx, x, frame_len * batch); if (rc) return rc;
    return launch_mirror(ctx, x, frame_len, batch);
}
int aeth_vec_mul_frames(aeth_ctx *ctx, aeth_cf32 *frames, size_t frame_len, size_t batch, const aeth_cf32 *sig, size_t n_sig)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    AETH_REQUIRE(n_sig == frame_len, AETH_E_LEN, AETH_MSG_VEC_LEN);            /* vecops.rs:100-104, per frame */
    if (frame_len == 0 || batch == 0) return AETH_OK;
    AETH_REQUIRE(frames && sig, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(frames) && aeth::aligned8(sig), AETH_E_ALIGN, "pointer not 8-byte aligned");
    return launch_mul_frames(ctx, frames, frame_len, batch, sig);
}
int aeth_vec_add(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)
{
    int rc = check_binary(ctx, a, n, b, nb); if (rc) return rc;
    return launch_ew<OP_ADD>(ctx, a, b, n, 0.f);
}
int aeth_vec_sub(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)
{
    int rc = check_binary(ctx, a, n, b, nb); if (rc) return rc;
    return launch_ew<OP_SUB>(ctx, a, b, n, 0.f);
}
int aeth_vec_mul(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)
{
    int rc = check_binary(ctx, a, n, b, nb); if (rc) return rc;
    return launch_ew<OP_MUL>(ctx, a, b, n, 0.f);
}
int aeth_vec_div(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)
{
    int rc = check_binary(ctx, a, n, b, nb); if (rc) return rc;
    return launch_ew<OP_DIV>(ctx, a, b, n, 0.f);
}
int aeth_vec_clone(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)
{
    int rc = check_binary(ctx, a, n, b, nb); if (rc) return rc;
    return launch_ew<OP_CLONE>(ctx, a, b, n, 0.f);
}

/* Scale::scale, src/fft.rs:22-37 */
float aeth_scale_factor(int kind, size_t n, float x)
{
    switch (kind) {
    case AETH_SCALE_SN: return 1.0f / sqrtf((float)n);
    case AETH_SCALE_N:  return 1.0f / (float)n;
    case AETH_SCALE_X:  return x;
    default:            return 1.0f;
    }
}

int aeth_scale_apply(aeth_ctx *ctx, int kind, float x, aeth_cf32 *data, size_t n)
{
    AETH_REQUIRE(kind >= AETH_SCALE_NONE && kind <= AETH_SCALE_X, AETH_E_ARG, "bad scale kind %d", kind);
    if (kind == AETH_SCALE_NONE) return check_unary(ctx, data, n);
    return aeth_vec_scale(ctx, data, n, aeth_scale_factor(kind, n, x));
}

/* ---- host-slice flavours ---------------------------------------------------- */
int aeth_host_vec_scale(aeth_ctx *ctx, aeth_cf32 *x, size_t n, float s)
{
    AETH_REQUIRE(ctx && (x || !n), AETH_E_ARG, "null argument");
    return host_roundtrip(ctx, x, n, nullptr, true, [&](aeth_cf32 *d, const aeth_cf32 *) { return aeth_vec_scale(ctx, d, n, s); });
}
int aeth_host_vec_conj(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    AETH_REQUIRE(ctx && (x || !n), AETH_E_ARG, "null argument");
    return host_roundtrip(ctx, x, n, nullptr, true, [&](aeth_cf32 *d, const aeth_cf32 *) { return aeth_vec_conj(ctx, d, n); });
}
int aeth_host_vec_zero(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    AETH_REQUIRE(ctx && (x || !n), AETH_E_ARG, "null argument");
    return host_roundtrip(ctx, x, n, nullptr, false, [&](aeth_cf32 *d, const aeth_cf32 *) { return aeth_vec_zero(ctx, d, n); });
}
int aeth_host_vec_mirror(aeth_ctx *ctx, aeth_cf32 *x, size_t n)
{
    AETH_REQUIRE(ctx && (x || !n), AETH_E_ARG, "null argument");
    return host_roundtrip(ctx, x, n, nullptr, true, [&](aeth_cf32 *d, const aeth_cf32 *) { return aeth_vec_mirror(ctx, d, n); });
}
#define AETH_HOST_BINARY(NAME, UPLOAD_SELF)                                                               \
    int aeth_host_vec_##NAME(aeth_ctx *ctx, aeth_cf32 *a, size_t n, const aeth_cf32 *b, size_t nb)        \
    {                                                                                                     \
        AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");                                                     \
        AETH_REQUIRE(n == nb, AETH_E_LEN, AETH_MSG_VEC_LEN);                                              \
        AETH_REQUIRE((a && b) || !n, AETH_E_ARG, "null argument");                                        \
        return host_roundtrip(ctx, a, n, b, UPLOAD_SELF, [&](aeth_cf32 *d, const aeth_cf32 *o) {          \
            return aeth_vec_##NAME(ctx, d, n, o, n);                                                      \
        });                                                                                               \
    }
AETH_HOST_BINARY(add, true)
AETH_HOST_BINARY(sub, true)
AETH_HOST_BINARY(mul, true)
AETH_HOST_BINARY(div, true)
AETH_HOST_BINARY(clone, false)

}  // extern "C"
