// aeth_fir_kernel.h -- device side of the fused  FFT -> (* H) -> IFFT  kernel (aeth_fir.hip holds the host side
// and the reference citations; tools/fir_lab.hip instantiates the same template for A/B measurements).
//
// One 128-lane workgroup per overlap-save block (N = 2048: 16 points per lane), persistent loop over blocks,
// the next block's window prefetched into registers while the current one is transformed.
#pragma once

#include "aeth_internal.h"
#include "aeth_fft_core.h"

namespace aeth {
namespace firk {

using namespace aeth::fftk;

struct FmiArgs {
    const cf *in;
    cf *out;
    const cf *hist;       // ntaps-1 samples preceding in[0], or null
    const cf *Hf;         // N spectrum multipliers, natural order
    const cf *twN;
    const cf *twL;        // per-lane twiddle table of the plan (null: gather from twN)
    long long n;          // samples in `in` / outputs wanted
    long long nblocks;
    int hop, ov, nhist;
    float s_fwd, s_bwd;
    // chirp-z (Bluestein) mode: a block is one frame of frame_n < N samples, multiplied by chirp[e] on the way in
    // and on the way out, zero beyond frame_n; conj = transform with the other exponent sign
    const cf *chirp = nullptr;
    int frame_n = 0;      // valid samples per window (0: the whole window)
    int conj = 0;
    // V_DECIM: only every dec-th output sample is stored, out[i] = y[i * dec] (fir -> sampling::downsample in one pass)
    aeth::FastDiv dec = {1, 0, 0, 0};
    long long n_out = 0;  // samples in `out` = n / dec
    // V_DEMOD: the chain's output goes through Modulation::demod_naive instead of to memory (hard decisions only)
    unsigned char *bits = nullptr;    // n * bps bytes, one per bit
    cf tab[4] = {};                   // BPSK / QPSK symbol table
    int bps = 0, demod_compat = 0;
    int demod_sep = 0;                // host side only: picks the kernel build (DM_QSEP / DM_QGEN, see demod_block)
};

// Kernel variants (template parameter VAR, a bit set).  0 is the round-1 kernel.
enum : int {
    V_PEEL  = 1,    // first block peeled out of the loop: window 0 is waited for alone (counted vmcnt), the tables
                    // and window 1 land under the first transform instead of in front of it
    V_TOUCH = 2,    // one 4-byte load per 128-byte line of the window two rounds ahead: pulls it into L2 / the
                    // Infinity Cache so that the register prefetch one round later is served on-die
    V_PRIO  = 4,    // s_setprio 1 around every LDS exchange (its latency chain is what a block's time is made of)
    V_TOUCH3 = 8,   // with V_TOUCH: three rounds ahead instead of two
    V_DECIM = 512,  // product variant: decimating store (aeth_fir_exec_decim)
    V_UNROLL2 = 4096, // the block loop unrolled by two with the roles of the two window register sets swapped (no copy)
    V_XOR = 2048,   // XOR-swizzled LDS exchange image instead of the padded one (see aeth_fft_core.h: pidx)
    V_XCD = 8192,   // lab only, measured negative (54.1 -> 55.5 us, two queues 48.5 -> 50.2): workgroups of one XCD
                    // (blockIdx % 8) take ADJACENT blocks of a round, so that the 63-sample halo a block shares with
                    // its neighbour is read through the same L2 -- round-robin over the XCDs spreads every region of
                    // the stream over all eight L2s and wins
    V_SPREAD = 16384, // the next window's 16 loads issued in four groups between the passes of this block's forward
                    // transform instead of one burst in front of it (the TA command FIFO is full 40 % of the time)
    V_DEMOD = 1024, // product variant: hard demodulation instead of the sample store (aeth_fft_mul_ifft_demod)
    V_DM_BPSK = 1 << 16, V_DM_QGEN = 1 << 17,   // with V_DEMOD: the decision's mode (neither: QPSK, separable table)
    V_DMA = 1 << 18, // the next window goes straight into a 16 KiB LDS landing image (buffer_load_dwordx4 ... lds: 8 pieces
                    // of 1 KiB per wave and block instead of 16 register loads) and is read from there at the start of its
                    // own iteration: no prefetch registers, no register copy per block; paid for with ONE exchange image
                    // (two barriers per exchange) so that four workgroups still fit a CU.  N = 2048 (two waves) only.
    V_NOLOAD = 16,  // diagnosis only (wrong output): no window loads inside the loop
    V_NOSTORE = 32, // diagnosis only (wrong output): no output stores inside the loop
    V_CENSUS = 256, // diagnosis only: every wave records HW_ID / XCC_ID in the buffer passed as `chirp`
    V_NOLDS = 128,  // diagnosis only (wrong output): no LDS exchanges at all
    V_NOBAR = 64,   // diagnosis only (wrong output): LDS exchanges without workgroup barriers
};

// What libaether_hip.so may instantiate; everything else is measurement / diagnosis and builds only where
// AETH_FIR_LAB is defined non-zero before this header is included (tools/fir_lab.hip)
constexpr int V_PRODUCT_MASK = V_PRIO | V_XOR | V_SPREAD | V_DECIM | V_DEMOD | V_DM_BPSK | V_DM_QGEN;
#ifndef AETH_FIR_LAB
#define AETH_FIR_LAB 0
#endif

// cache-policy bits of the streamed accesses (aux operand of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1);
// macros so that tools/fir_lab can be built with other choices
#ifndef AETH_FIR_LOAD_AUX
#define AETH_FIR_LOAD_AUX 2
#endif
#ifndef AETH_FIR_STORE_AUX
#define AETH_FIR_STORE_AUX 18
#endif

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Stream accesses carry the non-temporal hint (aux bit 1) when the launch moves more than the cache holds: a
// window is read once and an output block written once, so neither should displace the tables in L2 or take the
// write-allocate path (A/B in one process on 256 MiB: stores alone -3 %, loads alone +3 %, both -5 % of the launch
// time).  NT is a kernel template parameter: short chains over cache-sized operands (C4) keep plain accesses.

__device__ __forceinline__ cf as_cf(u32x2 v) { return __builtin_bit_cast(cf, v); }
__device__ __forceinline__ u32x2 as_u32x2(cf v) { return __builtin_bit_cast(u32x2, v); }

// one block's input window -> registers (slot m = window element tid + m*T)
template <class C, bool NT>
__device__ __forceinline__ void load_window(cf (&x)[C::P], const FmiArgs &a, long long blk, int tid)
{
    const long long win0 = blk * a.hop - a.ov;              // first input sample of the window
    if (blk >= a.nblocks) {
#pragma unroll
        for (int m = 0; m < C::P; m++) x[m] = mk(0.f, 0.f);
        return;
    }
    if constexpr (C::F == 1) {
        if (win0 >= 0) {
            // wave-uniform window: buffer loads, the descriptor's range check zero-fills past the end
            long long left = a.n - win0;
            int bytes = (int)(left < a.frame_n ? left : a.frame_n) * 8;
            auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.in + win0), 0, bytes, 0x00020000);
#pragma unroll
            for (int m = 0; m < C::P; m++)
                x[m] = as_cf(__builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * C::T) * 8, 0, NT ? AETH_FIR_LOAD_AUX : 0));
            // (the window's oldest ov-nhist samples are zeroed when the window is consumed: doing it
            // here would put a wait for the load right behind its issue)
            return;
        }
    }
    if constexpr (C::F > 1) {
        // several windows per workgroup: when the whole group lies inside the stream (workgroup-uniform test)
        // the loads need no per-element range logic, only the zeroing of the samples older than the history
        const long long first = (blk - (long long)(threadIdx.x / C::T)) * a.hop - a.ov;
        const long long last_end = first + (long long)(C::F - 1) * a.hop + C::N;
        if (a.frame_n == C::N && first >= 0 && last_end <= a.n && blk - (long long)(threadIdx.x / C::T) + C::F <= a.nblocks) {
#pragma unroll
            for (int m = 0; m < C::P; m++) {
                const cf v = a.in[win0 + tid + m * C::T];
                x[m] = (tid + m * C::T >= a.ov - a.nhist) ? v : mk(0.f, 0.f);
            }
            return;
        }
        if (a.frame_n < C::N && a.ov == 0 && blk - (long long)(threadIdx.x / C::T) + C::F <= a.nblocks) {
            // chirp-z frames (hop = frame_n samples each, zero beyond): every frame of the group exists
#pragma unroll
            for (int m = 0; m < C::P; m++) {
                const int e = tid + m * C::T;
                x[m] = e < a.frame_n ? a.in[win0 + e] : mk(0.f, 0.f);
            }
            return;
        }
    }
#pragma unroll
    for (int m = 0; m < C::P; m++) {
        const long long gi = win0 + tid + m * C::T;
        cf v = mk(0.f, 0.f);
        if (tid + m * C::T >= a.ov - a.nhist && tid + m * C::T < a.frame_n) {
            if (gi >= 0) { if (gi < a.n) v = a.in[gi]; }
            else if (a.hist && gi >= -(long long)a.nhist) v = a.hist[a.nhist + gi];
        }
        x[m] = v;
    }
}

// Branch-free form for the steady state of one-frame workgroups (win0 >= 0 guaranteed by
// the caller): a block past the end gets a zero-length descriptor, so the loads still
// issue -- and return zeros without touching memory.  No divergent path means hipcc can
// COUNT the loads in flight (vmcnt(N)) instead of falling back to vmcnt(0).
template <class C, bool NT>
__device__ __forceinline__ void load_window_srd(cf (&x)[C::P], const FmiArgs &a, long long blk, int tid)
{
    const bool active = blk < a.nblocks;
    const long long win0 = active ? blk * a.hop - a.ov : 0;
    long long left = a.n - win0;
    const int bytes = active ? (int)(left < a.frame_n ? left : a.frame_n) * 8 : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.in + win0), 0, bytes, 0x00020000);
#pragma unroll
    for (int m = 0; m < C::P; m++)
        x[m] = as_cf(__builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * C::T) * 8, 0, NT ? AETH_FIR_LOAD_AUX : 0));
}

// the same, slots [M0, M1) only (V_SPREAD: the window arrives in four instalments)
template <class C, bool NT, int M0, int M1>
__device__ __forceinline__ void load_window_srd_part(cf (&x)[C::P], const FmiArgs &a, long long blk, int tid)
{
    const bool active = blk < a.nblocks;
    const long long win0 = active ? blk * a.hop - a.ov : 0;
    long long left = a.n - win0;
    const int bytes = active ? (int)(left < a.frame_n ? left : a.frame_n) * 8 : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.in + win0), 0, bytes, 0x00020000);
#pragma unroll
    for (int m = M0; m < M1; m++)
        x[m] = as_cf(__builtin_amdgcn_raw_buffer_load_b64(rs, (tid + m * C::T) * 8, 0, NT ? AETH_FIR_LOAD_AUX : 0));
    __builtin_amdgcn_sched_barrier(0);
}

// ---- V_DMA: the window through an LDS landing image ---------------------------------------------------------
// The image is lane-linear per wave-instruction (LDS dest = base + lane * 16), so the permutation sits on the SOURCE
// side: piece j of wave wv fetches the 512-byte runs of window rows m = 2j and 2j+1 (a row = T elements = 1 KiB) that
// this very wave reads in pass 0 (element tid + m*T).  No wave reads what another wave landed, so the issuing wave's
// own vmcnt orders its ds_reads behind the DMA and no barrier is needed.  A lane past the descriptor's range lands
// zeros (tools/shape_ab.hip, oob_probe), so the ragged end of the stream and a non-existent block need no branch.
typedef __attribute__((address_space(3))) void *lds_vptr_t;
template <bool NT>
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rs, cf *dst, int voff, int soff)
{
#if __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)dst, 16, voff, soff, 0, NT ? AETH_FIR_LOAD_AUX : 0);
#endif
}
template <class C, bool NT, int J0, int J1>
__device__ __forceinline__ void dma_window_part(cf *land, const FmiArgs &a, long long blk, int tid)
{
    static_assert(C::T == 128 && C::P == 16, "landing image: two waves, sixteen rows of 1 KiB");
    const bool active = blk < a.nblocks;
    const long long win0 = active ? blk * a.hop - a.ov : 0;
    long long left = a.n - win0;
    const int bytes = active ? (int)(left < a.frame_n ? left : a.frame_n) * 8 : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.in + win0), 0, bytes, 0x00020000);
    const int wv = tid >> 6, l = tid & 63;
    const int voff = (l >> 5) * 1024 + wv * 512 + (l & 31) * 16;
#pragma unroll
    for (int j = J0; j < J1; j++) dma_piece<NT>(rs, land + wv * 1024 + j * 128, voff, j * 2048);
    __builtin_amdgcn_sched_barrier(0);
}
// slot of window element tid + m*T in the landing image (elements)
template <class C> __device__ __forceinline__ int land_slot(int tid, int m) { return (tid >> 6) * 1024 + (m >> 1) * 128 + (m & 1) * 64 + (tid & 63); }

// V_TOUCH: one dword per 128-byte line of block `blk`'s window (plain cache policy: the line is meant to stay on
// the die until the register prefetch reads it); the value is never used, the caller only keeps it alive.
template <class C>
__device__ __forceinline__ unsigned touch_window(const FmiArgs &a, long long blk, int tid)
{
    const bool active = blk < a.nblocks;
    const long long win0 = active ? blk * a.hop - a.ov : 0;
    long long left = a.n - win0;
    const int bytes = active ? (int)(left < a.frame_n ? left : a.frame_n) * 8 : 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.in + win0), 0, bytes, 0x00020000);
    unsigned acc = 0;
#pragma unroll
    for (int off = 0; off < C::N * 8; off += C::T * 128)
        acc |= __builtin_amdgcn_raw_buffer_load_b32(rs, off + tid * 128, 0, 0);
    return acc;
}

// Modulation::demod_naive on the block's output samples (modulation.rs:33-56 for [cf32; 4], :133-144 for [cf32; 2]):
// nearest table symbol by squared distance, d = rn(rn(dr*dr) + rn(di*di)) -- the products are rounded before the sum
// (packed multiplies and adds as asm statements: nothing for -ffp-contract=fast to fuse), so the decisions are those
// of aeth_demod_naive on the stored samples -- folded as the reference's min_by does: the running minimum is replaced
// by a strictly smaller distance and by an unordered pair (`v_cmp_nle`: !(best <= d)), equal distances keep the first.
//
// Same skeleton as store_block<CHECK = false>: no branch anywhere.  The mode (DM) is a template parameter, the bit
// bytes leave through a per-block buffer descriptor whose range check drops what must not be written (a block past
// the end: zero-length descriptor; samples in front of the valid part: offset out of range), so hipcc can count the
// memory operations in flight and the loop's only wait stays `vmcnt(16)`.
//   DM_BPSK  two candidates, one byte per sample
//   DM_QSEP  QPSK table of the form {(a,c), (b,c), (a,d), (b,d)} (the generic one is): the four distances are built
//            from four squares instead of eight -- six packed instructions per sample, bit for bit the values of DM_QGEN
//   DM_QGEN  any other four-entry table
enum : int { DM_BPSK = 1, DM_QSEP = 2, DM_QGEN = 3 };

// (a.x - t.x, a.x - t.y) / (a.y - t.x, a.y - t.y): one sample component against two table components (t wave-uniform)
__device__ __forceinline__ cf diff_re(cf a, cf t) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "s"(t)); return d; }
__device__ __forceinline__ cf diff_im(cf a, cf t) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "s"(t)); return d; }
__device__ __forceinline__ cf pk_sq(cf a) { cf d; asm("v_pk_mul_f32 %0, %1, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ cf pk_sum(cf a, cf b) { cf d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// (a.x + b.x, a.y + b.x) and (a.x + b.y, a.y + b.y)
__device__ __forceinline__ cf pk_sum_lo(cf a, cf b) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ cf pk_sum_hi(cf a, cf b) { cf d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }

// wave-uniform operands of the decision, built once per launch from the table in the kernel arguments
struct DemodK {
    cf x01, y01, x23, y23;      // (t0.x, t1.x), (t0.y, t1.y), (t2.x, t3.x), (t2.y, t3.y); DM_QSEP uses x01 and (t0.y, t2.y) in y01
    unsigned hi;                // what bit 1 of the index contributes to the stored pair of bytes: 0x100, or 0x200 for `idx & 2` (modulation.rs:54)
};

template <int DM>
__device__ __forceinline__ DemodK demod_consts(const FmiArgs &a)
{
    DemodK k;
    k.x01 = mk(a.tab[0].x, a.tab[1].x); k.y01 = mk(a.tab[0].y, DM == DM_QSEP ? a.tab[2].y : a.tab[1].y);
    k.x23 = mk(a.tab[2].x, a.tab[3].x); k.y23 = mk(a.tab[2].y, a.tab[3].y);
    k.hi = a.demod_compat ? 0x200u : 0x100u;
    return k;
}

// the bytes demod_naive emits for one sample: bit 0 of the index in byte 0, bit 1 (or idx & 2) in byte 1
template <int DM>
__device__ __forceinline__ unsigned demod_decide(cf v, const DemodK &k)
{
    if constexpr (DM == DM_BPSK) {
        const cf d = pk_sum(pk_sq(diff_re(v, k.x01)), pk_sq(diff_im(v, k.y01)));     // (d0, d1)
        return !(d.x <= d.y) ? 1u : 0u;
    } else {
        cf d01, d23;
        if constexpr (DM == DM_QSEP) {
            const cf px = pk_sq(diff_re(v, k.x01)), py = pk_sq(diff_im(v, k.y01));   // (px0, px1), (py0, py1)
            d01 = pk_sum_lo(px, py); d23 = pk_sum_hi(px, py);
        } else {
            d01 = pk_sum(pk_sq(diff_re(v, k.x01)), pk_sq(diff_im(v, k.y01)));
            d23 = pk_sum(pk_sq(diff_re(v, k.x23)), pk_sq(diff_im(v, k.y23)));
        }
        const bool c1 = !(d01.x <= d01.y);
        const float b1 = c1 ? d01.y : d01.x;
        const bool c2 = !(b1 <= d23.x);
        const float b2 = c2 ? d23.x : b1;
        const bool c3 = !(b2 <= d23.y);
        const bool bit0 = c3 | (c1 & !c2), bit1 = c3 | c2;                           // index = c3 ? 3 : c2 ? 2 : c1 ? 1 : 0
        return (bit0 ? 1u : 0u) | (bit1 ? k.hi : 0u);
    }
}

template <class C, bool SCALED, bool NT, int DM>
__device__ __forceinline__ void demod_block(const cf (&w)[C::P], const FmiArgs &a, const DemodK &k, long long blk, int tid)
{
    static_assert(C::F == 1, "demodulating store: one block per workgroup");
    constexpr int B = DM == DM_BPSK ? 1 : 2;                        // bytes per sample
    const long long base = blk * a.hop - a.ov;
    long long left = a.n - base;
    int bytes = (int)(left < a.frame_n ? left : a.frame_n) * B;
    if (blk >= a.nblocks) bytes = 0;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(a.bits + base * B, 0, bytes, 0x00020000);
    const cf ss = mk(a.s_bwd, a.s_bwd);
#pragma unroll
    for (int m = 0; m < C::P; m++) {
        const int e = tid + m * C::T;
        const int off = (e >= a.ov) ? e * B : 0x7ffffff0;
        const cf v = SCALED ? cscale_k(w[m], ss) : w[m];
        const unsigned o = demod_decide<DM>(v, k);
        if constexpr (B == 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)o, rs, off, 0, NT ? AETH_FIR_STORE_AUX : 0);
        else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)o, rs, off, 0, NT ? AETH_FIR_STORE_AUX : 0);
    }
}

// CHECK = false: the caller knows that the block exists (no branch around the stores, so that hipcc keeps
// counting the memory operations in flight across them)
template <class C, bool SCALED, bool NT, bool CHECK = true, bool DECIM = false>
__device__ __forceinline__ void store_block(const cf (&w)[C::P], const FmiArgs &a, long long blk, int tid)
{
    if constexpr (CHECK) { if (blk >= a.nblocks) return; }
    const long long base = blk * a.hop - a.ov;              // output index of window element 0
    if constexpr (C::F == 1 && DECIM) {
        // sampling::downsample behind the filter (sampling.rs:28-42: dst[i] = src[i * dec]) folded into the store:
        // output sample o of the stream is kept iff dec divides it, and lands at out[o / dec].  32-bit index
        // arithmetic (the host side checks n < 2^31); q0 = first kept index of this block, descriptor from there.
        const unsigned o0 = (unsigned)(blk * a.hop);                              // first output sample of the block
        const unsigned q0 = aeth::fdiv(o0 + a.dec.d - 1, a.dec);
        const long long left = a.n_out - (long long)q0;
        const int bytes = left > 0 ? (int)(left < (long long)C::N ? left : (long long)C::N) * 8 : 0;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(a.out + q0, 0, bytes, 0x00020000);
        const cf ss = mk(a.s_bwd, a.s_bwd);
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            const int e = tid + m * C::T;
            const unsigned o = o0 + (unsigned)(e - a.ov);                         // meaningful for e >= ov only
            const unsigned q = aeth::fdiv(o, a.dec);
            const bool keep = e >= a.ov && q * a.dec.d == o && (long long)o < a.n;
            const int off = keep ? (int)(q - q0) * 8 : 0x7ffffff0;
            cf v = SCALED ? cscale_k(w[m], ss) : w[m];
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v), rs, off, 0, NT ? AETH_FIR_STORE_AUX : 0);
        }
    } else if constexpr (C::F == 1) {
        long long left = a.n - base;
        int bytes = (int)(left < a.frame_n ? left : a.frame_n) * 8;   // stores past the end (of the stream, of the frame) are dropped by the range check
        if constexpr (!CHECK) { if (blk >= a.nblocks) bytes = 0; }    // a block past the end: every store dropped, no branch
        auto rs = __builtin_amdgcn_make_buffer_rsrc(a.out + base, 0, bytes, 0x00020000);
        const cf ss = mk(a.s_bwd, a.s_bwd);
        // no branch around the stores either: window elements in front of the valid part
        // (e < ov) get an offset past the descriptor's range and are dropped by its check
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            const int e = tid + m * C::T;
            const int off = (e >= a.ov) ? e * 8 : 0x7ffffff0;
            cf v = SCALED ? cscale_k(w[m], ss) : w[m];
            __builtin_amdgcn_raw_buffer_store_b64(as_u32x2(v), rs, off, 0, NT ? AETH_FIR_STORE_AUX : 0);   // nt + sc1: streamed stores (tools/nt_modes.hip: 6.55 vs 6.43 TB/s for nt alone)
        }
    } else {
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            const int e = tid + m * C::T;
            const long long o = base + e;
            if (e >= a.ov && o < a.n && e < a.frame_n) a.out[o] = SCALED ? cscale(w[m], a.s_bwd) : w[m];
        }
    }
}

// the chain on one block held in w[]: vec_rfft -> vec_mul -> vec_rifft (benches/benches.rs:410-416)
// V_SPREAD form for three-pass configurations: nx[] = the window of block `gn`, loaded in four instalments
template <class C, bool NT, int VAR>
__device__ __forceinline__ void transform_block_spread(cf (&w)[C::P], cf (&nx)[C::P], const cf (&tw)[C::TW], const cf (&H)[C::P],
                                                       cf *__restrict__ lds, const FmiArgs &a, long long gn, int tid)
{
    static_assert(C::NPASS == 3 && C::P % 4 == 0, "spread loads: three passes");
    constexpr int XP = ((VAR & V_PRIO) ? 1 : 0) | ((VAR & V_XOR) ? 8 : 0);
    constexpr int Q = C::P / 4;
    load_window_srd_part<C, NT, 0, Q>(nx, a, gn, tid);
    run_pass<C, 0, +1, 0, XP>(w, tw, lds, tid);
    load_window_srd_part<C, NT, Q, 2 * Q>(nx, a, gn, tid);
    run_pass<C, 1, +1, 0, XP>(w, tw, lds, tid);
    load_window_srd_part<C, NT, 2 * Q, 3 * Q>(nx, a, gn, tid);
    run_pass<C, 2, +1, 0, XP>(w, tw, lds, tid);
#pragma unroll
    for (int m = 0; m < C::P; m++) w[m] = cmul(w[m], H[m]);
    load_window_srd_part<C, NT, 3 * Q, 4 * Q>(nx, a, gn, tid);
    fft_in_regs<C, -1, fft_next_par<C>(0), XP>(w, tw, lds, tid);
}

template <class C, bool SCALED, bool BLU, int VAR>
__device__ __forceinline__ void transform_block(cf (&w)[C::P], const cf (&tw)[C::TW], const cf (&H)[C::P],
                                                cf *__restrict__ lds, const FmiArgs &a, int tid)
{
    constexpr int XP = ((VAR & V_PRIO) ? 1 : 0) | ((VAR & V_NOBAR) ? 2 : 0) | ((VAR & V_NOLDS) ? 4 : 0) | ((VAR & V_XOR) ? 8 : 0);
    if constexpr (BLU) {
        // x[n] (conjugated for the other exponent sign) * chirp[n]; chirp is 0 beyond the frame (descriptor range)
        auto cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.chirp), 0, a.frame_n * 8, 0x00020000);
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            const cf ch = as_cf(__builtin_amdgcn_raw_buffer_load_b64(cr, (tid + m * C::T) * 8, 0, 0));
            cf v = w[m];
            if (a.conj) v.y = -v.y;
            w[m] = cmul(v, ch);
        }
    }
    fft_in_regs<C, +1, 0, XP>(w, tw, lds, tid);             // vec_rfft: the reference's fwd (+j exponent)
    if constexpr (SCALED) {
        const cf ss = mk(a.s_fwd, a.s_fwd);
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = cscale_k(w[m], ss);           // Scale of vec_rfft
    }
#pragma unroll
    for (int m = 0; m < C::P; m++) w[m] = cmul(w[m], H[m]);                 // vec_mul (vecops.rs:99-112)
    fft_in_regs<C, -1, fft_next_par<C>(0), XP>(w, tw, lds, tid);   // vec_rifft: bwd (-j); two transforms leave the parity even
    if constexpr (BLU) {
        auto cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(a.chirp), 0, a.frame_n * 8, 0x00020000);
#pragma unroll
        for (int m = 0; m < C::P; m++) {
            const cf ch = as_cf(__builtin_amdgcn_raw_buffer_load_b64(cr, (tid + m * C::T) * 8, 0, 0));
            cf v = cmul(w[m], ch);
            if (a.conj) v.y = -v.y;
            w[m] = v;
        }
    }
}

// SCALED = false: both Scale factors are 1 (FIR: 1/N is folded into H)
// BLU: chirp-z frames (see FmiArgs): the frame is multiplied by the chirp behind the load and in front of the store
template <class C, bool SCALED, int MINW, bool NT, bool BLU, int VAR = 0>
__global__ __launch_bounds__(C::WG, MINW) void fmi_kernel(FmiArgs a)
{
    static_assert(AETH_FIR_LAB || (VAR & ~V_PRODUCT_MASK) == 0, "lab-only kernel variant in a product build");
    __shared__ cf lds_all[C::LDS_TOTAL];
    // F == 1: the whole workgroup is one frame, so the block index stays provably wave-uniform
    const int tid = (C::F == 1) ? (int)threadIdx.x : (int)(threadIdx.x % C::T);
    const int fl = (C::F == 1) ? 0 : (int)(threadIdx.x / C::T);
    cf *lds = lds_all + fl * C::LDS_FRAME;

    if constexpr (VAR & V_CENSUS) {
        if ((threadIdx.x & 63) == 0) {
            unsigned *cb = reinterpret_cast<unsigned *>(const_cast<cf *>(a.chirp));
            const unsigned slot = blockIdx.x * (C::WG / 64) + threadIdx.x / 64;
            cb[2 * slot] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
            cb[2 * slot + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
        }
    }
    const long long ngroups = (a.nblocks + C::F - 1) / C::F;
    constexpr bool PEEL = (VAR & V_PEEL) && C::F == 1 && !BLU;
    constexpr bool TOUCH = (VAR & V_TOUCH) && C::F == 1 && !BLU;
    constexpr int TOUCH_AHEAD = (VAR & V_TOUCH3) ? 3 : 2;
    cf nx[C::P], tw[C::TW], H[C::P];
    constexpr int DM = (VAR & V_DM_BPSK) ? DM_BPSK : (VAR & V_DM_QGEN) ? DM_QGEN : DM_QSEP;
    static_assert(!(VAR & (V_DM_BPSK | V_DM_QGEN)) || (VAR & V_DEMOD), "decision mode without V_DEMOD");
    [[maybe_unused]] DemodK dk;
    if constexpr (VAR & V_DEMOD) dk = demod_consts<DM>(a);
    unsigned bid = blockIdx.x;
    if constexpr ((VAR & V_XCD) != 0) {
        if ((gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    }
    long long g0 = bid;
    unsigned tprev = 0;     // V_TOUCH: the previous round's touch result, kept alive until the next round (never used)

    if constexpr (PEEL) {
        // Block 0 of this workgroup outside the loop.  Issue order = return order: window 0, tables, window 1.  The
        // first pass needs window 0 only, so its wait leaves the tables and the next window in flight; they land
        // while pass 0 and the first exchange run.
        cf w[C::P];
        if (bid == 0) load_window<C, NT>(w, a, 0, tid);     // history / zero initial state: predicated path
        else load_window_srd<C, NT>(w, a, g0, tid);
        if (a.twL) load_twiddles_lane<C>(tw, a.twL, tid);
        else load_twiddles<C>(tw, a.twN, tid);
#pragma unroll
        for (int m = 0; m < C::P; m++) H[m] = a.Hf[tid + m * C::T];
        load_window_srd<C, NT>(nx, a, g0 + gridDim.x, tid);
        if constexpr (TOUCH) {
#pragma unroll
            for (int k = 2; k <= TOUCH_AHEAD; k++) tprev |= touch_window<C>(a, g0 + (long long)k * gridDim.x, tid);
        }
        if (tid < a.ov - a.nhist) w[0] = mk(0.f, 0.f);
        transform_block<C, SCALED, BLU, VAR>(w, tw, H, lds, a, tid);
        if constexpr (VAR & V_DEMOD) demod_block<C, SCALED, NT, DM>(w, a, dk, g0, tid);
        else store_block<C, SCALED, NT, false, (VAR & V_DECIM) != 0>(w, a, g0, tid);     // grid <= ngroups: the block exists
        g0 += gridDim.x;
    } else {
        // software pipeline: the next block's window is in flight while this one is transformed.
        // The first window goes out before the (L2-resident) tables so the HBM fetch starts at once.
        load_window<C, NT>(nx, a, (long long)bid * C::F + fl, tid);
        if (a.twL) load_twiddles_lane<C>(tw, a.twL, tid);
        else load_twiddles<C>(tw, a.twN, tid);
#pragma unroll
        for (int m = 0; m < C::P; m++) H[m] = a.Hf[tid + m * C::T];
        // Drain the table loads HERE, once.  Otherwise hipcc places their counted waits at the first
        // uses inside the loop body, where they run every iteration and end in vmcnt(0) halfway
        // through each block -- forcing the prefetched window AND the previous block's stores to
        // complete there instead of riding under the whole block.
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) only
    }
    if constexpr ((VAR & V_UNROLL2) && C::F == 1 && !BLU && !(VAR & (V_DEMOD | V_DECIM | V_TOUCH | V_NOLOAD | V_NOSTORE))) {
        // two blocks per trip: window registers A (= nx) and B swap roles, so no block starts with a 32-register copy.
        // Everything stays branch-free (blocks past the end load and store through zero-length descriptors).
        cf wb[C::P];
        auto one = [&](cf (&cur)[C::P], cf (&nxt)[C::P], long long g) {
            if (tid < a.ov - a.nhist) cur[0] = mk(0.f, 0.f);
            load_window_srd<C, NT>(nxt, a, g + gridDim.x, tid);
            transform_block<C, SCALED, BLU, VAR>(cur, tw, H, lds, a, tid);
            store_block<C, SCALED, NT, false, false>(cur, a, g, tid);
        };
#pragma unroll 1
        for (long long g = g0; g < ngroups; g += 2 * (long long)gridDim.x) {
            one(nx, wb, g);
            one(wb, nx, g + gridDim.x);
        }
        return;
    }
#pragma unroll 1
    for (long long g = g0; g < ngroups; g += gridDim.x) {
        const long long blk = g * C::F + fl;
        cf w[C::P];
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = nx[m];
        // window samples older than the ntaps-1 the outputs depend on are forced to zero, so a
        // block is a function of exactly x[out0-(ntaps-1) .. out0+hop): shards of one stream
        // (history = ntaps-1 samples) then reproduce the unsharded run bit for bit
        if constexpr (C::F == 1) { if (tid < a.ov - a.nhist) w[0] = mk(0.f, 0.f); }
        const long long gn = g + gridDim.x;
        if constexpr (VAR & V_NOLOAD) {
#pragma unroll
            for (int m = 0; m < C::P; m++) asm volatile("" : "+v"(nx[m]));     // opaque, so that nothing folds
        } else if constexpr (C::F == 1 && (VAR & V_SPREAD) && !SCALED && !BLU) {
            // (loads issued inside transform_block_spread)
        } else if constexpr (C::F == 1) {
            // gn >= gridDim.x >= 1, so the window never starts before the stream: descriptor path
            load_window_srd<C, NT>(nx, a, gn, tid);
        } else {
            if (gn < ngroups) load_window<C, NT>(nx, a, gn * C::F + fl, tid);
        }
        if constexpr (TOUCH) {
            asm volatile("" ::"v"(tprev));      // issued a whole round ago: no wait
            tprev = touch_window<C>(a, g + (long long)TOUCH_AHEAD * gridDim.x, tid);
        }
        if constexpr (C::F == 1 && (VAR & V_SPREAD) && !SCALED && !BLU) transform_block_spread<C, NT, VAR>(w, nx, tw, H, lds, a, gn, tid);
        else transform_block<C, SCALED, BLU, VAR>(w, tw, H, lds, a, tid);
        if constexpr (VAR & V_NOSTORE) {
#pragma unroll
            for (int m = 0; m < C::P; m++) asm volatile("" ::"v"(w[m]));
        } else if constexpr (VAR & V_DEMOD) demod_block<C, SCALED, NT, DM>(w, a, dk, blk, tid);
        else store_block<C, SCALED, NT, true, (VAR & V_DECIM) != 0>(w, a, blk, tid);
    }
    if constexpr (TOUCH) asm volatile("" ::"v"(tprev));
}

// ---- V_DMA build of the FIR kernel (N = 2048: two waves, sixteen 1 KiB window rows; unscaled, no chirp) ----------
template <class C> struct OneImg : C { static constexpr bool DB = false; static constexpr int LDS_TOTAL = C::LDS_ELEMS; };

template <class C0, bool NT, int VAR>
__global__ __launch_bounds__(C0::WG, 1) void fmi_dma_kernel(FmiArgs a)
{
    using C = OneImg<C0>;
    static_assert(C::F == 1 && C::T == 128 && C::P == 16 && C::NPASS == 3, "landing image layout: N = 2048");
    static_assert(AETH_FIR_LAB || (VAR & ~V_PRODUCT_MASK) == 0, "lab-only kernel variant in a product build");
    __shared__ __attribute__((aligned(1024))) cf lds_all[C::N + C::LDS_TOTAL];
    cf *land = lds_all, *lds = lds_all + C::N;              // [landing image | one exchange image]
    const int tid = (int)threadIdx.x;
    constexpr int XP = ((VAR & V_PRIO) ? 1 : 0) | ((VAR & V_XOR) ? 8 : 0) | 16;     // raw barriers: a DMA is always in flight
    cf tw[C::TW], H[C::P];
    const long long g0 = blockIdx.x;
    if (blockIdx.x == 0) {
        // history / zero initial state: predicated register path, parked in the landing image in the DMA's layout
        cf w0[C::P];
        load_window<C, NT>(w0, a, 0, tid);
#pragma unroll
        for (int m = 0; m < C::P; m++) land[land_slot<C>(tid, m)] = w0[m];
    } else dma_window_part<C, NT, 0, 8>(land, a, g0, tid);
    if (a.twL) load_twiddles_lane<C>(tw, a.twL, tid);
    else load_twiddles<C>(tw, a.twN, tid);
#pragma unroll
    for (int m = 0; m < C::P; m++) H[m] = a.Hf[tid + m * C::T];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // tables and the first window are in
#pragma unroll 1
    for (long long g = g0; g < a.nblocks; g += gridDim.x) {
        cf w[C::P];
#pragma unroll
        for (int m = 0; m < C::P; m++) w[m] = land[land_slot<C>(tid, m)];
        if (tid < a.ov - a.nhist) w[0] = mk(0.f, 0.f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the image is read out before the next window lands in it
        __builtin_amdgcn_sched_barrier(0);
        const long long gn = g + gridDim.x;
        if constexpr (VAR & V_SPREAD) {
            dma_window_part<C, NT, 0, 2>(land, a, gn, tid);
            run_pass<C, 0, +1, 0, XP>(w, tw, lds, tid);
            dma_window_part<C, NT, 2, 4>(land, a, gn, tid);
            run_pass<C, 1, +1, 0, XP>(w, tw, lds, tid);
            dma_window_part<C, NT, 4, 6>(land, a, gn, tid);
            run_pass<C, 2, +1, 0, XP>(w, tw, lds, tid);
#pragma unroll
            for (int m = 0; m < C::P; m++) w[m] = cmul(w[m], H[m]);
            dma_window_part<C, NT, 6, 8>(land, a, gn, tid);
            fft_in_regs<C, -1, 0, XP>(w, tw, lds, tid);
        } else {
            dma_window_part<C, NT, 0, 8>(land, a, gn, tid);
            fft_in_regs<C, +1, 0, XP>(w, tw, lds, tid);
#pragma unroll
            for (int m = 0; m < C::P; m++) w[m] = cmul(w[m], H[m]);
            fft_in_regs<C, -1, 0, XP>(w, tw, lds, tid);
        }
        store_block<C, false, NT, false, false>(w, a, g, tid);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // the next window has landed; this block's 16 stores stay in flight
    }
}

}  // namespace firk
}  // namespace aeth
