// aeth_modulation.hip -- bit -> symbol map and hard demodulation
// (reference: src/modulation.rs:5-149).  SURVEY 8f "next" row #1: completes BASELINE
// config 4 (QPSK mod -> AWGN -> FFT-2048 correlate -> hard demod) on the device.
// Compiled with -ffp-contract=off: the distance compare must round as the Rust code does.
// HBM-bound: modulate reads 1-2 B and writes 8 B per symbol; demod reads 8 B, writes 1-2 B.
#include "aeth_internal.h"

#define AETH_RNG_FN __host__ __device__ static inline
#include "aeth_rng.h"

namespace {

constexpr int kBlock = 256;
constexpr int kPairs = 2;          // modulate_awgn_kernel: sample pairs per lane
constexpr int kItems = 4;          // modulate_kernel / demod_kernel, vector form: 4-byte bit groups per lane

struct Table4 { float2 s[4]; };

// GENERIC_BPSK_TABLE / GENERIC_QPSK_TABLE, src/modulation.rs:77,87-92
const float2 kBpsk[2] = {{1.f, 1.f}, {-1.f, -1.f}};
const float2 kQpsk[4] = {{1.f, 1.f}, {-1.f, 1.f}, {1.f, -1.f}, {-1.f, -1.f}};

__device__ __forceinline__ unsigned qpsk_index(unsigned b0, unsigned b1) { return ((b1 & 1u) << 1) + (b0 & 1u); }   // modulation.rs:21-24

// table[idx] by selects on the (scalar-register) entries: a runtime index into the by-value table
// would turn into a second, dependent global load from the kernarg segment per symbol
__device__ __forceinline__ float2 pick(const Table4 &t, unsigned idx)
{
    const bool b0 = idx & 1u, b1 = idx & 2u;
    const float2 lo = b0 ? t.s[1] : t.s[0], hi = b0 ? t.s[3] : t.s[2];
    return b1 ? hi : lo;
}

// nearest_rule: the reference folds the candidates with min_by(|d, e| d.partial_cmp(e).unwrap_or(Ordering::Greater))
// (modulation.rs:46, :139).  min_by keeps the running minimum unless the comparison says Greater, so the minimum is
// replaced by a strictly smaller distance AND by any unordered pair (a NaN on either side); of equal distances the
// first stays.  `!(bd <= d)` is exactly that; a NaN sample decodes as the last candidate scanned.
__device__ __forceinline__ unsigned nearest(float2 v, const Table4 &t, int ncand)
{
    unsigned best = 0;
    float bd = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= ncand) break;
        const float dr = v.x - t.s[i].x, di = v.y - t.s[i].y;
        const float d = dr * dr + di * di;                                        // modulation.rs:36-41
        if (i == 0 || !(bd <= d)) { best = (unsigned)i; bd = d; }                 // min_by (:46-49): see nearest_rule below
    }
    return best;
}

// Four input bytes and 16-byte stores per lane where the alignment allows (VEC): these
// kernels move 10 B per symbol and are store- (modulate) or load-bound (demod).
// item index of lane `tid`, slot k: a wave owns kItems * 64 consecutive items and takes them 64 at a time, so every
// access of a wave is one contiguous run and the kItems loads of a lane are in flight together
__device__ __forceinline__ size_t item_of(int k)
{
    return ((size_t)blockIdx.x * kBlock + (threadIdx.x & ~63u)) * kItems + (size_t)k * 64 + (threadIdx.x & 63u);
}

template <int BPS, bool VEC, bool NT>
__global__ __launch_bounds__(kBlock) void modulate_kernel(const uint8_t *__restrict__ bits, float2 *__restrict__ out,
                                                          size_t nsym, Table4 t)
{
    if constexpr (VEC) {
        // kItems items (4 bytes of bits each) per lane, all loads first: a lane with one 4-byte load in flight leaves the
        // kernel bound by the latency of that load, not by its 10 B per symbol
        constexpr int SPL = 4 / BPS;                     // symbols per item
        uchar4 b[kItems];
#pragma unroll
        for (int k = 0; k < kItems; k++) {
            const size_t i = item_of(k);
            if (i * SPL < nsym) b[k] = aeth::nt_load<NT>(reinterpret_cast<const uchar4 *>(bits) + i);
        }
#pragma unroll
        for (int k = 0; k < kItems; k++) {
            const size_t i = item_of(k);
            if (i * SPL >= nsym) return;
            if constexpr (BPS == 2) {
                const float2 a = pick(t, qpsk_index(b[k].x, b[k].y)), c = pick(t, qpsk_index(b[k].z, b[k].w));
                aeth::nt_store<NT>(reinterpret_cast<float4 *>(out) + i, make_float4(a.x, a.y, c.x, c.y));
            } else {
                const float2 a = pick(t, b[k].x & 1u), c = pick(t, b[k].y & 1u), d = pick(t, b[k].z & 1u), e = pick(t, b[k].w & 1u);   // modulation.rs:9-12
                aeth::nt_store<NT>(reinterpret_cast<float4 *>(out) + 2 * i, make_float4(a.x, a.y, c.x, c.y));
                aeth::nt_store<NT>(reinterpret_cast<float4 *>(out) + 2 * i + 1, make_float4(d.x, d.y, e.x, e.y));
            }
        }
    } else {
        const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
        if (i >= nsym) return;
        unsigned idx;
        if constexpr (BPS == 1) idx = bits[i] & 1u;
        else idx = qpsk_index(bits[2 * i], bits[2 * i + 1]);
        out[i] = pick(t, idx);                                                    // modulation.rs:115-121
    }
}

template <int BPS, bool VEC, bool NT>
__global__ __launch_bounds__(kBlock) void demod_kernel(const float2 *__restrict__ sym, uint8_t *__restrict__ bits,
                                                       size_t nsym, Table4 t, int compat)
{
    constexpr int NC = (BPS == 1) ? 2 : 4;              // trait default scans BITS_PER_SYMBOL*2 candidates (:135)
    auto hi = [&](unsigned best) { return (uint8_t)(compat ? (best & 2u) : ((best >> 1) & 1u)); };   // modulation.rs:54
    if constexpr (VEC) {
        constexpr int SPL = 4 / BPS;
        constexpr int LD = (BPS == 2) ? 1 : 2;          // 16-byte loads per item
        float4 v[kItems][LD];
#pragma unroll
        for (int k = 0; k < kItems; k++) {              // kItems items per lane, all loads first (see modulate_kernel)
            const size_t i = item_of(k);
            if (i * SPL < nsym) {
#pragma unroll
                for (int j = 0; j < LD; j++) v[k][j] = aeth::nt_load<NT>(reinterpret_cast<const float4 *>(sym) + LD * i + j);
            }
        }
#pragma unroll
        for (int k = 0; k < kItems; k++) {
            const size_t i = item_of(k);
            if (i * SPL >= nsym) return;
            if constexpr (BPS == 2) {
                const unsigned a = nearest(make_float2(v[k][0].x, v[k][0].y), t, NC), c = nearest(make_float2(v[k][0].z, v[k][0].w), t, NC);
                aeth::nt_store<NT>(reinterpret_cast<uchar4 *>(bits) + i, make_uchar4((uint8_t)(a & 1u), hi(a), (uint8_t)(c & 1u), hi(c)));
            } else {
                reinterpret_cast<uchar4 *>(bits)[i] = make_uchar4((uint8_t)(nearest(make_float2(v[k][0].x, v[k][0].y), t, NC) & 1u),
                                                                  (uint8_t)(nearest(make_float2(v[k][0].z, v[k][0].w), t, NC) & 1u),
                                                                  (uint8_t)(nearest(make_float2(v[k][LD - 1].x, v[k][LD - 1].y), t, NC) & 1u),
                                                                  (uint8_t)(nearest(make_float2(v[k][LD - 1].z, v[k][LD - 1].w), t, NC) & 1u));   // :143
            }
        }
    } else {
        const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
        if (i >= nsym) return;
        const unsigned best = nearest(sym[i], t, NC);
        if constexpr (BPS == 1) bits[i] = (uint8_t)(best & 1u);
        else { bits[2 * i] = (uint8_t)(best & 1u); bits[2 * i + 1] = hi(best); }  // modulation.rs:53-54
    }
}

// ---- any implementor of trait Modulation with a symbol table (3 .. 8 bits per symbol): the trait's DEFAULT methods,
// modulation.rs:94-149.  index() = sum((bit % 2) << i) (:105-111); demod_naive scans candidates
// 0 .. BITS_PER_SYMBOL*2 (:135 -- not 2^BITS_PER_SYMBOL: reproduced when compat != 0, all 2^bps otherwise) and emits
// (idx >> i) & 1 (:143).  The table (<= 256 entries, 2 KiB) travels in the kernel arguments and is staged in LDS.
struct TableN { float2 s[256]; };
inline unsigned grid_for(size_t items);

template <bool NT>
__global__ __launch_bounds__(kBlock) void modulate_generic_kernel(const uint8_t *__restrict__ bits, float2 *__restrict__ out,
                                                                  size_t nsym, int bps, TableN t)
{
    __shared__ float2 tab[256];
    for (int k = threadIdx.x; k < (1 << bps); k += kBlock) tab[k] = t.s[k];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nsym) return;
    unsigned idx = 0;
    for (int k = 0; k < bps; k++) idx += (unsigned)(bits[i * (size_t)bps + k] % 2u) << k;      // :107-110
    aeth::nt_store<NT>(out + i, tab[idx]);                                                      // :115-121
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void demod_generic_kernel(const float2 *__restrict__ sym, uint8_t *__restrict__ bits,
                                                               size_t nsym, int bps, int ncand, TableN t)
{
    __shared__ float2 tab[256];
    for (int k = threadIdx.x; k < ncand; k += kBlock) tab[k] = t.s[k];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nsym) return;
    const float2 v = aeth::nt_load<NT>(sym + i);
    unsigned best = 0;
    float bd = 0.f;
    for (int c = 0; c < ncand; c++) {
        const float dr = v.x - tab[c].x, di = v.y - tab[c].y;                                   // :136
        const float d = dr * dr + di * di;                                                      // :137
        if (c == 0 || !(bd <= d)) { best = (unsigned)c; bd = d; }                               // min_by (:139), same rule
    }
    for (int k = 0; k < bps; k++) bits[i * (size_t)bps + k] = (uint8_t)((best >> k) & 1u);      // :143
}

// modulate, then Awgn::apply on the fresh symbols (examples/modem.rs:19-26: `let mut tx = qpsk().modulate(&bits);
// awgn.apply(&mut tx)`) in one pass: the symbol never goes to memory without its noise.  One lane per PAIR of
// symbols = one Philox call; arithmetic exactly that of modulate_kernel + awgn_apply_kernel (noise.rs:41-42,58).
template <int BPS, bool NT>
__global__ __launch_bounds__(kBlock) void modulate_awgn_kernel(const uint8_t *__restrict__ bits, float2 *__restrict__ out,
                                                               size_t nsym, Table4 t, float scale, uint64_t seed,
                                                               uint64_t offset, int wide)
{
    // the table in LDS: one indexed 8-byte read per symbol where pick() spends two compares and six selects (the kernel
    // is issue-bound on the generator's arithmetic: 70.7 -> 61.5 us per 2^25 symbols)
    __shared__ float2 tab[4];
    if (threadIdx.x < 4) tab[threadIdx.x] = t.s[threadIdx.x];
    __syncthreads();
    // The bit bytes of ALL of the lane's pairs are requested first and looked at last: the noise does not depend on them,
    // so their latency rides under the generator's arithmetic (it used to be waited for at the top of every pair)
    unsigned v[kPairs];
#pragma unroll
    for (int k = 0; k < kPairs; k++) {                               // kPairs pairs per lane, a grid's width apart (aeth_noise.hip)
        const size_t i0 = 2 * ((size_t)blockIdx.x * kBlock + threadIdx.x + (size_t)k * gridDim.x * kBlock);
        v[k] = 0;
        if (i0 >= nsym) break;
        const bool two = i0 + 1 < nsym;
        if ((wide & 2) && two) {                                        // the pair's bit bytes in one aligned load
            if constexpr (BPS == 1) v[k] = *reinterpret_cast<const uint16_t *>(bits + i0);
            else v[k] = *reinterpret_cast<const uint32_t *>(bits + 2 * i0);
        } else if constexpr (BPS == 1) { v[k] = bits[i0]; if (two) v[k] |= (unsigned)bits[i0 + 1] << 8; }
        else {
            v[k] = bits[2 * i0] | ((unsigned)bits[2 * i0 + 1] << 8);
            if (two) v[k] |= ((unsigned)bits[2 * i0 + 2] << 16) | ((unsigned)bits[2 * i0 + 3] << 24);
        }
    }
#pragma unroll
    for (int k = 0; k < kPairs; k++) {
        const size_t i0 = 2 * ((size_t)blockIdx.x * kBlock + threadIdx.x + (size_t)k * gridDim.x * kBlock);
        if (i0 >= nsym) return;
        const bool two = i0 + 1 < nsym;
        float n0r, n0i, n1r = 0.f, n1i = 0.f;
        if ((offset & 1) == 0) {
            uint32_t w[4];
            const uint64_t call = (offset + i0) >> 1;
            aeth_rng_draw(call, seed, w);
            aeth_rng_normal_pair(w[0], w[1], &n0r, &n0i);
            aeth_rng_normal_pair(w[2], w[3], &n1r, &n1i);
        } else {
            aeth_rng_cnormal(seed, offset + i0, &n0r, &n0i);
            if (two) aeth_rng_cnormal(seed, offset + i0 + 1, &n1r, &n1i);
        }
        unsigned idx0, idx1;
        if constexpr (BPS == 1) { idx0 = v[k] & 1u; idx1 = (v[k] >> 8) & 1u; }
        else { idx0 = qpsk_index(v[k], v[k] >> 8); idx1 = qpsk_index(v[k] >> 16, v[k] >> 24); }
        float2 a = tab[idx0], b = tab[idx1];
        a.x = a.x + (n0r * scale) * scale;          // noise.rs:41 then :58
        a.y = a.y + (n0i * scale) * scale;
        b.x = b.x + (n1r * scale) * scale;
        b.y = b.y + (n1i * scale) * scale;
        if ((wide & 1) && two) aeth::nt_store<NT>(reinterpret_cast<float4 *>(out + i0), make_float4(a.x, a.y, b.x, b.y));
        else { out[i0] = a; if (two) out[i0 + 1] = b; }
    }
}

int fill_table_n(TableN &t, int bps, const aeth_cf32 *host)
{
    AETH_REQUIRE(bps >= 3 && bps <= 8, AETH_E_UNSUPPORTED, "bits_per_symbol %d: 1 .. 8 supported", bps);
    AETH_REQUIRE(host, AETH_E_ARG, "bits_per_symbol %d needs a symbol table (the reference only ships BPSK and QPSK)", bps);
    for (int i = 0; i < 256; i++) t.s[i] = make_float2(0.f, 0.f);
    for (int i = 0; i < (1 << bps); i++) t.s[i] = make_float2(host[i].re, host[i].im);
    return AETH_OK;
}

int fill_table(Table4 &t, int bps, const aeth_cf32 *host)
{
    AETH_REQUIRE(bps == 1 || bps == 2, AETH_E_UNSUPPORTED, "bits_per_symbol %d: only BPSK (1) and QPSK (2) tables exist in the reference", bps);
    const int n = bps == 1 ? 2 : 4;
    for (int i = 0; i < 4; i++) t.s[i] = make_float2(0.f, 0.f);
    for (int i = 0; i < n; i++)
        t.s[i] = host ? make_float2(host[i].re, host[i].im) : (bps == 1 ? kBpsk[i] : kQpsk[i]);
    return AETH_OK;
}

inline unsigned grid_for(size_t items) { size_t b = (items + kBlock - 1) / kBlock; return (unsigned)(b < 1 ? 1 : b); }

}  // namespace

extern "C" {

int aeth_modulate(aeth_ctx *ctx, const uint8_t *bits, size_t nbits, int bps, const aeth_cf32 *table, aeth_cf32 *out,
                  size_t n_out)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (bps > 2) {
        TableN tn;
        int rcn = fill_table_n(tn, bps, table); if (rcn) return rcn;
        AETH_REQUIRE(nbits % (size_t)bps == 0, AETH_E_LEN, "bit count %zu is not a multiple of BITS_PER_SYMBOL %d", nbits, bps);
        AETH_REQUIRE(n_out == nbits / (size_t)bps, AETH_E_LEN, "output holds %zu symbols, input gives %zu", n_out, nbits / (size_t)bps);
        if (n_out == 0) return AETH_OK;
        AETH_REQUIRE(bits && out, AETH_E_ARG, "null pointer");
        AETH_REQUIRE(aeth::aligned8(out), AETH_E_ALIGN, "pointer alignment");
        aeth::DeviceGuard dg(ctx->device);
        auto k = aeth::streams_past_cache(n_out * sizeof(float2)) ? modulate_generic_kernel<true> : modulate_generic_kernel<false>;
        hipLaunchKernelGGL(k, dim3(grid_for(n_out)), dim3(kBlock), 0, aeth::ctx_stream(ctx), bits, (float2 *)out, n_out, bps, tn);
        AETH_HIP(hipGetLastError());
        return AETH_OK;
    }
    Table4 t;
    int rc = fill_table(t, bps, table); if (rc) return rc;
    AETH_REQUIRE(nbits % (size_t)bps == 0, AETH_E_LEN, "bit count %zu is not a multiple of BITS_PER_SYMBOL %d", nbits, bps);
    AETH_REQUIRE(n_out == nbits / (size_t)bps, AETH_E_LEN, "output holds %zu symbols, input gives %zu", n_out, nbits / (size_t)bps);
    if (n_out == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    AETH_REQUIRE(bits && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(out) && ((uintptr_t)bits % (size_t)bps) == 0, AETH_E_ALIGN, "pointer alignment");
    const size_t spl = 4 / (size_t)bps;
    const bool vec = aeth::aligned16(out) && ((uintptr_t)bits % 4) == 0 && (n_out % spl) == 0;
    const dim3 gv(grid_for((n_out / spl + kItems - 1) / kItems)), gs(grid_for(n_out)), b(kBlock);
    const bool nt = aeth::streams_past_cache(n_out * sizeof(float2));
#define AETH_MOD(B, V, G)                                                                                                 \
    do {                                                                                                                  \
        if (nt) hipLaunchKernelGGL((modulate_kernel<B, V, true>), G, b, 0, aeth::ctx_stream(ctx), bits, (float2 *)out, n_out, t);   \
        else hipLaunchKernelGGL((modulate_kernel<B, V, false>), G, b, 0, aeth::ctx_stream(ctx), bits, (float2 *)out, n_out, t);     \
    } while (0)
    if (bps == 1 && vec) AETH_MOD(1, true, gv);
    else if (bps == 1)   AETH_MOD(1, false, gs);
    else if (vec)        AETH_MOD(2, true, gv);
    else                 AETH_MOD(2, false, gs);
#undef AETH_MOD
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_modulate_awgn(aeth_ctx *ctx, const uint8_t *bits, size_t nbits, int bps, const aeth_cf32 *table, aeth_cf32 *out,
                       size_t n_out, float power, uint64_t seed, uint64_t offset)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    Table4 t;
    int rc = fill_table(t, bps, table); if (rc) return rc;          // BPSK / QPSK (the reference's modem example)
    AETH_REQUIRE(nbits % (size_t)bps == 0, AETH_E_LEN, "bit count %zu is not a multiple of BITS_PER_SYMBOL %d", nbits, bps);
    AETH_REQUIRE(n_out == nbits / (size_t)bps, AETH_E_LEN, "output holds %zu symbols, input gives %zu", n_out, nbits / (size_t)bps);
    AETH_REQUIRE(power >= 0.0f, AETH_E_ARG, "noise power must be >= 0");
    if (n_out == 0) return AETH_OK;
    AETH_REQUIRE(bits && out, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(out), AETH_E_ALIGN, "pointer alignment");
    aeth::DeviceGuard dev_guard(ctx->device);
    const float scale = sqrtf(power);                               // noise.rs:35
    const size_t pairs = (n_out + 1) / 2;
    const bool nt = aeth::streams_past_cache(n_out * sizeof(float2));
    const int wide = (aeth::aligned16(out) ? 1 : 0) | (((uintptr_t)bits % (size_t)(2 * bps)) == 0 ? 2 : 0);
#define AETH_MA(B)                                                                                                              \
    do {                                                                                                                        \
        if (nt) hipLaunchKernelGGL((modulate_awgn_kernel<B, true>), dim3(grid_for((pairs + kPairs - 1) / kPairs)), dim3(kBlock), 0, aeth::ctx_stream(ctx), \
                                   bits, (float2 *)out, n_out, t, scale, seed, offset, wide);                                  \
        else hipLaunchKernelGGL((modulate_awgn_kernel<B, false>), dim3(grid_for((pairs + kPairs - 1) / kPairs)), dim3(kBlock), 0, aeth::ctx_stream(ctx),   \
                                bits, (float2 *)out, n_out, t, scale, seed, offset, wide);                                     \
    } while (0)
    if (bps == 1) AETH_MA(1); else AETH_MA(2);
#undef AETH_MA
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

int aeth_demod_naive(aeth_ctx *ctx, const aeth_cf32 *sym, size_t nsym, int bps, const aeth_cf32 *table, uint8_t *bits,
                     size_t nbits_out, int compat)
{
    AETH_REQUIRE(ctx, AETH_E_ARG, "ctx is null");
    if (bps > 2) {
        TableN tn;
        int rcn = fill_table_n(tn, bps, table); if (rcn) return rcn;
        AETH_REQUIRE(nbits_out == nsym * (size_t)bps, AETH_E_LEN, "output holds %zu bits, input gives %zu", nbits_out, nsym * (size_t)bps);
        if (nsym == 0) return AETH_OK;
        AETH_REQUIRE(sym && bits, AETH_E_ARG, "null pointer");
        AETH_REQUIRE(aeth::aligned8(sym), AETH_E_ALIGN, "pointer alignment");
        aeth::DeviceGuard dg(ctx->device);
        const int ncand = compat ? ((2 * bps < (1 << bps)) ? 2 * bps : (1 << bps)) : (1 << bps);   // modulation.rs:135
        auto k = aeth::streams_past_cache(nsym * sizeof(float2)) ? demod_generic_kernel<true> : demod_generic_kernel<false>;
        hipLaunchKernelGGL(k, dim3(grid_for(nsym)), dim3(kBlock), 0, aeth::ctx_stream(ctx), (const float2 *)sym, bits, nsym, bps, ncand, tn);
        AETH_HIP(hipGetLastError());
        return AETH_OK;
    }
    Table4 t;
    int rc = fill_table(t, bps, table); if (rc) return rc;
    AETH_REQUIRE(nbits_out == nsym * (size_t)bps, AETH_E_LEN, "output holds %zu bits, input gives %zu", nbits_out, nsym * (size_t)bps);
    if (nsym == 0) return AETH_OK;
    aeth::DeviceGuard dev_guard(ctx->device);
    AETH_REQUIRE(sym && bits, AETH_E_ARG, "null pointer");
    AETH_REQUIRE(aeth::aligned8(sym) && ((uintptr_t)bits % (size_t)bps) == 0, AETH_E_ALIGN, "pointer alignment");
    const size_t spl = 4 / (size_t)bps;
    const bool vec = aeth::aligned16(sym) && ((uintptr_t)bits % 4) == 0 && (nsym % spl) == 0;
    const dim3 gv(grid_for((nsym / spl + kItems - 1) / kItems)), gs(grid_for(nsym)), b(kBlock);
    const bool nt = aeth::streams_past_cache(nsym * sizeof(float2));
#define AETH_DEM(B, V, G)                                                                                                         \
    do {                                                                                                                          \
        if (nt) hipLaunchKernelGGL((demod_kernel<B, V, true>), G, b, 0, aeth::ctx_stream(ctx), (const float2 *)sym, bits, nsym, t, compat);  \
        else hipLaunchKernelGGL((demod_kernel<B, V, false>), G, b, 0, aeth::ctx_stream(ctx), (const float2 *)sym, bits, nsym, t, compat);    \
    } while (0)
    if (bps == 1 && vec) AETH_DEM(1, true, gv);
    else if (bps == 1)   AETH_DEM(1, false, gs);
    else if (vec)        AETH_DEM(2, true, gv);
    else                 AETH_DEM(2, false, gs);
#undef AETH_DEM
    AETH_HIP(hipGetLastError());
    return AETH_OK;
}

}  // extern "C"
