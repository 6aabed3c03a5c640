// aether_hip.hpp -- header-only C++17 host wrapper over the C ABI (aether_hip.h).
//
// The reference is Rust; its toolchain is absent from this pipeline, so the host
// side above the C ABI is mirrored here in C++ with the reference's own names,
// argument meaning and error behaviour:
//   aether::Scale            <-> enum Scale              (src/fft.rs:6-38)
//   aether::Fft / HipFft     <-> trait Fft / struct Cfft (src/fft.rs:48-77, :134-235)
//   aether::DeviceVec        <-> trait VecOps on [cf32]  (src/vecops.rs:39-89), device-resident
//   aether::HostVec          <-> the same on a host slice (one H2D + D2H per call)
//   aether::interpolate / downsample                     (src/sampling.rs:7-62)
//   aether::assert_evm                                   (src/lib.rs:26-49)
// The reference panics on misuse (assert_eq!); here the same conditions throw
// aether::Panic carrying the reference's message text.
#pragma once

#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "aether_hip.h"

namespace aether {

using cf32 = std::complex<float>;     // layout-compatible with aeth_cf32 / Complex<f32>
static_assert(sizeof(cf32) == sizeof(aeth_cf32), "cf32 layout");

struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc)
{
    if (rc != AETH_OK) throw Panic(rc, aeth_last_error());
}

inline aeth_cf32 *raw(cf32 *p) { return reinterpret_cast<aeth_cf32 *>(p); }
inline const aeth_cf32 *raw(const cf32 *p) { return reinterpret_cast<const aeth_cf32 *>(p); }

// ---- enum Scale (src/fft.rs:6-18) ------------------------------------------------
struct Scale {
    int kind;
    float x;
    static Scale None() { return {AETH_SCALE_NONE, 0.f}; }
    static Scale SN() { return {AETH_SCALE_SN, 0.f}; }
    static Scale N() { return {AETH_SCALE_N, 0.f}; }
    static Scale X(float v) { return {AETH_SCALE_X, v}; }
    float factor(size_t n) const { return aeth_scale_factor(kind, n, x); }
};

class Context {
public:
    explicit Context(int device = 0) { check(aeth_ctx_create(device, &h_)); }
    ~Context() { aeth_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    aeth_ctx *get() const { return h_; }
    void sync() const { check(aeth_ctx_sync(h_)); }
    // consecutive independent Fir::filter launches alternate between two HIP queues (include/aether_hip.h)
    void set_overlap(bool enable = true) const { check(aeth_ctx_set_overlap(h_, enable ? 1 : 0)); }

private:
    aeth_ctx *h_ = nullptr;
};

class HipFft;

// ---- trait VecOps, device-resident receiver (src/vecops.rs:39-89) ------------------
class DeviceVec {
public:
    DeviceVec(Context &ctx, size_t n) : ctx_(&ctx), n_(n), own_(true)
    {
        void *p = nullptr;
        check(aeth_dev_alloc(ctx.get(), (n ? n : 1) * sizeof(aeth_cf32), &p));
        p_ = static_cast<aeth_cf32 *>(p);
    }
    DeviceVec(Context &ctx, const std::vector<cf32> &host) : DeviceVec(ctx, host.size())
    {
        check(aeth_upload(ctx.get(), p_, host.data(), host.size() * sizeof(cf32)));
    }
    // borrowed view `self[start..stop]`
    DeviceVec(DeviceVec &parent, size_t start, size_t stop)
        : ctx_(parent.ctx_), p_(parent.p_ + start), n_(stop - start), own_(false)
    {
        if (start > stop || stop > parent.n_) throw Panic(AETH_E_LEN, "slice index out of range");
    }
    ~DeviceVec() { if (own_ && p_) aeth_dev_free(ctx_->get(), p_); }
    DeviceVec(const DeviceVec &) = delete;
    DeviceVec &operator=(const DeviceVec &) = delete;
    DeviceVec(DeviceVec &&o) noexcept : ctx_(o.ctx_), p_(o.p_), n_(o.n_), own_(o.own_) { o.p_ = nullptr; o.own_ = false; }

    size_t len() const { return n_; }
    aeth_cf32 *ptr() const { return p_; }
    Context &ctx() const { return *ctx_; }
    std::vector<cf32> to_host() const
    {
        std::vector<cf32> h(n_);
        check(aeth_download(ctx_->get(), h.data(), p_, n_ * sizeof(cf32)));
        return h;
    }

    DeviceVec &vec_scale(float s) { check(aeth_vec_scale(c(), p_, n_, s)); return *this; }
    DeviceVec &vec_mul(const DeviceVec &o) { check(aeth_vec_mul(c(), p_, n_, o.p_, o.n_)); return *this; }
    DeviceVec &vec_div(const DeviceVec &o) { check(aeth_vec_div(c(), p_, n_, o.p_, o.n_)); return *this; }
    DeviceVec &vec_conj() { check(aeth_vec_conj(c(), p_, n_)); return *this; }
    DeviceVec &vec_mirror() { check(aeth_vec_mirror(c(), p_, n_)); return *this; }
    DeviceVec &vec_clone(const DeviceVec &o) { check(aeth_vec_clone(c(), p_, n_, o.p_, o.n_)); return *this; }
    DeviceVec &vec_zero() { check(aeth_vec_zero(c(), p_, n_)); return *this; }
    DeviceVec &vec_add(const DeviceVec &o) { check(aeth_vec_add(c(), p_, n_, o.p_, o.n_)); return *this; }
    DeviceVec &vec_sub(const DeviceVec &o) { check(aeth_vec_sub(c(), p_, n_, o.p_, o.n_)); return *this; }
    // closure per element, in order (src/vecops.rs:179-182): cannot cross the FFI, so
    // the data takes a round trip through the host -- slow by design.
    DeviceVec &vec_mutate(const std::function<void(cf32 &)> &f)
    {
        auto h = to_host();
        for (auto &z : h) f(z);
        check(aeth_upload(c(), p_, h.data(), n_ * sizeof(cf32)));
        return *this;
    }
    inline DeviceVec &vec_fft(Scale s);                    // fresh plan (src/vecops.rs:185-189)
    inline DeviceVec &vec_ifft(Scale s);
    inline DeviceVec &vec_rfft(HipFft &fft, Scale s);      // reused plan (src/vecops.rs:198-207)
    inline DeviceVec &vec_rifft(HipFft &fft, Scale s);

    // A chain of the element-wise methods in ONE pass over memory (aeth_vec_chain): v.fused().vec_add(a).vec_mul(b).vec_conj().run()
    class Chain {
    public:
        explicit Chain(DeviceVec &v) : v_(v) {}
        Chain &vec_scale(float s) { steps_.push_back({AETH_VEC_SCALE, nullptr, 0, s}); return *this; }
        Chain &vec_mul(const DeviceVec &o) { return bin(AETH_VEC_MUL, o); }
        Chain &vec_div(const DeviceVec &o) { return bin(AETH_VEC_DIV, o); }
        Chain &vec_conj() { steps_.push_back({AETH_VEC_CONJ, nullptr, 0, 0.f}); return *this; }
        Chain &vec_add(const DeviceVec &o) { return bin(AETH_VEC_ADD, o); }
        Chain &vec_sub(const DeviceVec &o) { return bin(AETH_VEC_SUB, o); }
        Chain &vec_clone(const DeviceVec &o) { return bin(AETH_VEC_CLONE, o); }
        Chain &vec_zero() { steps_.push_back({AETH_VEC_ZERO, nullptr, 0, 0.f}); return *this; }
        DeviceVec &run() { check(aeth_vec_chain(v_.c(), v_.ptr(), v_.len(), steps_.data(), steps_.size())); steps_.clear(); return v_; }
    private:
        Chain &bin(int op, const DeviceVec &o) { steps_.push_back({op, o.ptr(), o.len(), 0.f}); return *this; }
        DeviceVec &v_;
        std::vector<aeth_vec_step> steps_;
    };
    Chain fused() { return Chain(*this); }
private:
    aeth_ctx *c() const { return ctx_->get(); }
    Context *ctx_;
    aeth_cf32 *p_ = nullptr;
    size_t n_ = 0;
    bool own_ = false;
};

// ---- trait Fft (src/fft.rs:48-77) ---------------------------------------------------
struct Fft {
    virtual ~Fft() = default;
    virtual void fwd(const cf32 *input, size_t n_in, cf32 *output, size_t n_out, Scale s) = 0;
    virtual void bwd(const cf32 *input, size_t n_in, cf32 *output, size_t n_out, Scale s) = 0;
    virtual void ifwd(cf32 *input, size_t n, Scale s) = 0;
    virtual void ibwd(cf32 *input, size_t n, Scale s) = 0;
    virtual const cf32 *tfwd(const cf32 *input, size_t n, Scale s) = 0;
    virtual const cf32 *tbwd(const cf32 *input, size_t n, Scale s) = 0;
    virtual size_t len() const = 0;
};

// `impl Fft for HipFft`: plugs in where the reference's Cfft does (src/fft.rs:134-235).
// The exponent sign is bound to the method names HERE and nowhere else.
class HipFft final : public Fft {
public:
    static constexpr int kFwdSign = AETH_SIGN_REF_FWD;     // Cfft plans fwd with inverse=true (src/fft.rs:148)
    static constexpr int kBwdSign = AETH_SIGN_REF_BWD;
    HipFft(Context &ctx, size_t len, size_t max_batch = 1) { check(aeth_fft_create(ctx.get(), len, max_batch, &h_)); }
    static HipFft with_len(Context &ctx, size_t len) { return HipFft(ctx, len); }      // Cfft::with_len
    ~HipFft() override { aeth_fft_destroy(h_); }
    HipFft(const HipFft &) = delete;
    HipFft(HipFft &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }

    // host slices: the literal trait methods
    void fwd(const cf32 *in, size_t n_in, cf32 *out, size_t n_out, Scale s) override { check(aeth_fft_exec_host(h_, raw(in), n_in, raw(out), n_out, kFwdSign, s.kind, s.x)); }
    void bwd(const cf32 *in, size_t n_in, cf32 *out, size_t n_out, Scale s) override { check(aeth_fft_exec_host(h_, raw(in), n_in, raw(out), n_out, kBwdSign, s.kind, s.x)); }
    void ifwd(cf32 *io, size_t n, Scale s) override { check(aeth_fft_exec_host(h_, raw(io), n, raw(io), n, kFwdSign, s.kind, s.x)); }
    void ibwd(cf32 *io, size_t n, Scale s) override { check(aeth_fft_exec_host(h_, raw(io), n, raw(io), n, kBwdSign, s.kind, s.x)); }
    const cf32 *tfwd(const cf32 *in, size_t n, Scale s) override { return tmp(in, n, kFwdSign, s); }
    const cf32 *tbwd(const cf32 *in, size_t n, Scale s) override { return tmp(in, n, kBwdSign, s); }
    size_t len() const override { return aeth_fft_len(h_); }

    // device-resident frames (len(v) = batch * len()), stream-ordered
    void fwd(const DeviceVec &in, DeviceVec &out, Scale s) { exec(in, out, kFwdSign, s); }
    void bwd(const DeviceVec &in, DeviceVec &out, Scale s) { exec(in, out, kBwdSign, s); }
    void ifwd(DeviceVec &io, Scale s) { exec(io, io, kFwdSign, s); }
    void ibwd(DeviceVec &io, Scale s) { exec(io, io, kBwdSign, s); }
    // frames.vec_rfft(s).vec_mul(sig).vec_rifft(s) fused (benches/benches.rs:410-416)
    void mul_chain(DeviceVec &frames, const DeviceVec &sig, Scale s_fwd, Scale s_bwd)
    {
        size_t n = len();
        check(aeth_fft_mul_ifft(h_, frames.ptr(), frames.len(), n ? frames.len() / n : 0, sig.ptr(), sig.len(),
                                s_fwd.kind, s_fwd.x, s_bwd.kind, s_bwd.x));
    }
    aeth_fft *get() const { return h_; }

private:
    void exec(const DeviceVec &in, DeviceVec &out, int sign, Scale s)
    {
        if (out.len() != in.len()) throw Panic(AETH_E_LEN, "Output and FFT must be the same length");
        size_t n = len();
        check(aeth_fft_exec(h_, in.ptr(), in.len(), out.ptr(), n ? in.len() / n : 0, sign, s.kind, s.x));
    }
    const cf32 *tmp(const cf32 *in, size_t n, int sign, Scale s)
    {
        const aeth_cf32 *view = nullptr;
        check(aeth_fft_exec_tmp_host(h_, raw(in), n, sign, s.kind, s.x, &view));
        return reinterpret_cast<const cf32 *>(view);
    }
    aeth_fft *h_ = nullptr;
};

// the reference plans per call (vecops.rs:185-189); the context keeps the plans these calls built
inline DeviceVec &DeviceVec::vec_fft(Scale s) { check(aeth_vec_fft(c(), p_, n_, HipFft::kFwdSign, s.kind, s.x)); return *this; }
inline DeviceVec &DeviceVec::vec_ifft(Scale s) { check(aeth_vec_fft(c(), p_, n_, HipFft::kBwdSign, s.kind, s.x)); return *this; }
inline DeviceVec &DeviceVec::vec_rfft(HipFft &fft, Scale s) { fft.ifwd(*this, s); return *this; }
inline DeviceVec &DeviceVec::vec_rifft(HipFft &fft, Scale s) { fft.ibwd(*this, s); return *this; }

// ---- VecOps on a host slice: the literal drop-in receiver ----------------------------
class HostVec {
public:
    HostVec(Context &ctx, cf32 *data, size_t n) : ctx_(&ctx), p_(data), n_(n) {}
    HostVec(Context &ctx, std::vector<cf32> &v) : HostVec(ctx, v.data(), v.size()) {}
    HostVec &vec_scale(float s) { check(aeth_host_vec_scale(c(), raw(p_), n_, s)); return *this; }
    HostVec &vec_mul(const std::vector<cf32> &o) { check(aeth_host_vec_mul(c(), raw(p_), n_, raw(o.data()), o.size())); return *this; }
    HostVec &vec_div(const std::vector<cf32> &o) { check(aeth_host_vec_div(c(), raw(p_), n_, raw(o.data()), o.size())); return *this; }
    HostVec &vec_conj() { check(aeth_host_vec_conj(c(), raw(p_), n_)); return *this; }
    HostVec &vec_mirror() { check(aeth_host_vec_mirror(c(), raw(p_), n_)); return *this; }
    HostVec &vec_clone(const std::vector<cf32> &o) { check(aeth_host_vec_clone(c(), raw(p_), n_, raw(o.data()), o.size())); return *this; }
    HostVec &vec_zero() { check(aeth_host_vec_zero(c(), raw(p_), n_)); return *this; }
    HostVec &vec_add(const std::vector<cf32> &o) { check(aeth_host_vec_add(c(), raw(p_), n_, raw(o.data()), o.size())); return *this; }
    HostVec &vec_sub(const std::vector<cf32> &o) { check(aeth_host_vec_sub(c(), raw(p_), n_, raw(o.data()), o.size())); return *this; }
    HostVec &vec_mutate(const std::function<void(cf32 &)> &f) { for (size_t i = 0; i < n_; i++) f(p_[i]); return *this; }
    HostVec &vec_fft(Scale s) { check(aeth_host_vec_fft(c(), raw(p_), n_, HipFft::kFwdSign, s.kind, s.x)); return *this; }
    HostVec &vec_ifft(Scale s) { check(aeth_host_vec_fft(c(), raw(p_), n_, HipFft::kBwdSign, s.kind, s.x)); return *this; }
    HostVec &vec_rfft(Fft &fft, Scale s) { fft.ifwd(p_, n_, s); return *this; }
    HostVec &vec_rifft(Fft &fft, Scale s) { fft.ibwd(p_, n_, s); return *this; }

private:
    aeth_ctx *c() const { return ctx_->get(); }
    Context *ctx_;
    cf32 *p_;
    size_t n_;
};

// ---- pinned host buffers: pool::Pool<T> / Elem<T> (src/pool.rs:43-221) ------------------------------
// Pool::make(ctx, n_samples, initial_len): elements are pinned buffers of n_samples cf32 each (the maker), optionally
// zeroed on return (the resetter).  take() is empty (null Elem) when the pool is; take_or_make() grows it.  An Elem
// derefs into its samples and returns itself to the pool when it goes out of scope (Elem::drop, :196-208).
class Pool {
public:
    class Elem {
    public:
        Elem() = default;
        Elem(aeth_pool *pool, cf32 *p, size_t n) : pool_(pool), p_(p), n_(n) {}
        Elem(Elem &&o) noexcept : pool_(o.pool_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
        Elem &operator=(Elem &&o) noexcept { reset(); pool_ = o.pool_; p_ = o.p_; n_ = o.n_; o.p_ = nullptr; return *this; }
        Elem(const Elem &) = delete;
        Elem &operator=(const Elem &) = delete;
        ~Elem() { reset(); }
        explicit operator bool() const { return p_ != nullptr; }          // Option<Elem<T>>::is_some()
        cf32 *data() { return p_; }
        const cf32 *data() const { return p_; }
        size_t size() const { return n_; }
        cf32 &operator[](size_t i) { return p_[i]; }
        void reset() { if (p_) { aeth_pool_give_back(pool_, p_); p_ = nullptr; } }
    private:
        aeth_pool *pool_ = nullptr;
        cf32 *p_ = nullptr;
        size_t n_ = 0;
    };
    static Pool make(Context &ctx, size_t n_samples, size_t initial_len, bool zero_on_return = false)
    {
        Pool p; p.n_ = n_samples;
        check(aeth_pool_create(ctx.get(), n_samples * sizeof(cf32), initial_len, zero_on_return ? AETH_POOL_ZERO_ON_RETURN : 0, &p.h_));
        return p;
    }
    Pool(Pool &&o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = nullptr; }
    Pool(const Pool &) = delete;
    Pool &operator=(const Pool &) = delete;
    ~Pool() { if (h_) aeth_pool_destroy(h_); }
    Elem take() { void *b = nullptr; check(aeth_pool_take(h_, &b)); return Elem(h_, static_cast<cf32 *>(b), b ? n_ : 0); }
    Elem take_or_make() { void *b = nullptr; check(aeth_pool_take_or_make(h_, &b)); return Elem(h_, static_cast<cf32 *>(b), n_); }
    size_t len() const { return aeth_pool_len(h_); }
    size_t cap() const { return aeth_pool_cap(h_); }
    bool is_empty() const { return len() == 0; }
private:
    Pool() = default;
    aeth_pool *h_ = nullptr;
    size_t n_ = 0;
};

// ---- FIR (src/fir.rs:3-22 has the struct, not the filter) ------------------------------
class Fir {
public:
    Fir(Context &ctx, const std::vector<cf32> &taps, size_t fft_len = 2048) { check(aeth_fir_create(ctx.get(), raw(taps.data()), taps.size(), fft_len, &h_)); }
    ~Fir() { aeth_fir_destroy(h_); }
    Fir(const Fir &) = delete;
    size_t hop() const { return aeth_fir_hop(h_); }
    void filter(const DeviceVec &x, DeviceVec &y, const DeviceVec *hist = nullptr)
    {
        if (y.len() != x.len()) throw Panic(AETH_E_LEN, "Vectors must have same length");
        check(aeth_fir_exec(h_, hist ? hist->ptr() : nullptr, x.ptr(), x.len(), y.ptr()));
    }
    void filter(const std::vector<cf32> &x, std::vector<cf32> &y)
    {
        y.resize(x.size());
        check(aeth_fir_exec_host(h_, nullptr, raw(x.data()), x.size(), raw(y.data())));
    }
    // host-resident stream through the upload | kernel | download pipeline (src/pipeline.rs counterpart); with report =
    // true the reference pipeline's per-stage lines (pipeline.rs:101-108) go to stdout
    aeth_pipe_util filter_stream(const std::vector<cf32> &x, std::vector<cf32> &y, size_t chunk = 0, bool report = false)
    {
        y.resize(x.size());
        aeth_pipe_util u{};
        check(aeth_fir_stream_host_util(h_, raw(x.data()), x.size(), raw(y.data()), chunk, &u));
        if (report) print_report(u);
        return u;
    }
    // the same on raw slices: memory inside a pinned Pool element (or a registered range) is copied from / to directly
    aeth_pipe_util filter_stream(const cf32 *x, size_t n, cf32 *y, size_t chunk = 0, bool report = false)
    {
        aeth_pipe_util u{};
        check(aeth_fir_stream_host_util(h_, raw(x), n, raw(y), chunk, &u));
        if (report) print_report(u);
        return u;
    }
    static void print_report(const aeth_pipe_util &u)
    {
        if (!(u.seconds > 0)) return;
        const char *names[5] = {"copy-in", "upload", "kernel", "download", "copy-out"};
        const double act[5] = {u.active_copy_in, u.active_upload, u.active_kernel, u.active_download, u.active_copy_out};
        for (int s = 0; s < 5; s++) {
            if ((s == 0 || s == 4) && act[s] == 0) continue;       // a side that was copied directly has no host stage
            std::printf("Stage: %-15s : Processed %llu in %3.3fs (%9.2f/s); Utilisation: %3.2f%%\n", names[s],
                        (unsigned long long)u.chunks, u.seconds, u.chunks / u.seconds, act[s] / u.seconds * 100.0);
        }
    }
    // the filter followed by sampling::downsample (sampling.rs:28-42) in one pass: y[i] = fir(x)[i * (x.len / y.len)]
    void filter_decim(const DeviceVec &x, DeviceVec &y, const DeviceVec *hist = nullptr)
    {
        check(aeth_fir_exec_decim(h_, hist ? hist->ptr() : nullptr, x.ptr(), x.len(), y.ptr(), y.len()));
    }

    aeth_fir *get() const { return h_; }

private:
    aeth_fir *h_ = nullptr;
};

// ---- pipeline (src/pipeline.rs:24-41 add_stage, :123-137 new, :89-114 the per-stage report) ------------------
// The reference chains closures over channels; the device pipeline has five fixed stages (copy-in | upload | compute |
// download | copy-out) and the compute stage is one of the library's ops:
//     auto y = pipeline::stage_fft(fft, Scale::SN()).run(ctx, x);            // Vec<cf32> -> Vec<cf32>
//     auto b = pipeline::stage_correlate_demod(fft, sig, 2).run_bits(ctx, x);  // Vec<cf32> -> Vec<u8>
namespace pipeline {
struct Stage {
    aeth_stream_op op{};
    size_t out_count(Context &ctx, size_t n_in) const { return aeth_stream_out_count(ctx.get(), &op, n_in); }
    // host slices; with report = true the reference's per-stage lines go to stdout
    aeth_pipe_util run(Context &ctx, const void *in, size_t n_in, void *out, size_t n_out, size_t chunk = 0, bool report = false) const
    {
        aeth_pipe_util u{};
        check(aeth_stream_host_util(ctx.get(), &op, in, n_in, out, n_out, chunk, &u));
        if (report) Fir::print_report(u);
        return u;
    }
    std::vector<cf32> run(Context &ctx, const std::vector<cf32> &x, size_t chunk = 0) const
    {
        std::vector<cf32> y(out_count(ctx, x.size()));
        run(ctx, x.data(), x.size(), y.data(), y.size(), chunk);
        return y;
    }
    std::vector<uint8_t> run_bits(Context &ctx, const std::vector<cf32> &x, size_t chunk = 0) const
    {
        std::vector<uint8_t> y(out_count(ctx, x.size()));
        run(ctx, x.data(), x.size(), y.data(), y.size(), chunk);
        return y;
    }
};
inline Stage stage_fir(const Fir &f) { Stage s; s.op.kind = AETH_STREAM_FIR; s.op.fir = f.get(); return s; }
inline Stage stage_fir_decim(const Fir &f, size_t dec) { Stage s; s.op.kind = AETH_STREAM_FIR_DECIM; s.op.fir = f.get(); s.op.n_between = dec; return s; }
inline Stage stage_fft(const HipFft &f, Scale sc, int sign = HipFft::kFwdSign)
{
    Stage s; s.op.kind = AETH_STREAM_FFT; s.op.fft = f.get(); s.op.sign = sign; s.op.scale_kind_fwd = sc.kind; s.op.x_fwd = sc.x; return s;
}
inline Stage stage_mul_chain(const HipFft &f, const DeviceVec &sig, Scale s_fwd, Scale s_bwd)
{
    Stage s; s.op.kind = AETH_STREAM_FFT_MUL_IFFT; s.op.fft = f.get(); s.op.sig_dev = sig.ptr(); s.op.n_sig = sig.len();
    s.op.scale_kind_fwd = s_fwd.kind; s.op.x_fwd = s_fwd.x; s.op.scale_kind_bwd = s_bwd.kind; s.op.x_bwd = s_bwd.x; return s;
}
inline Stage stage_correlate_demod(const HipFft &f, const DeviceVec &sig, int bits_per_symbol, bool compat = true)
{
    Stage s; s.op.kind = AETH_STREAM_FFT_MUL_IFFT_DEMOD; s.op.fft = f.get(); s.op.sig_dev = sig.ptr(); s.op.n_sig = sig.len();
    s.op.bits_per_symbol = bits_per_symbol; s.op.compat = compat ? 1 : 0; return s;
}
inline Stage stage_fft_interpolate(const HipFft &f, size_t n_between, Scale sc, bool compat_im = true)
{
    Stage s; s.op.kind = AETH_STREAM_FFT_INTERPOLATE; s.op.fft = f.get(); s.op.sign = HipFft::kFwdSign; s.op.scale_kind_fwd = sc.kind;
    s.op.x_fwd = sc.x; s.op.n_between = n_between; s.op.compat = compat_im ? 1 : 0; return s;
}
}  // namespace pipeline

// ---- sampling (src/sampling.rs) ---------------------------------------------------------
// appends to dst, as the reference does (sampling.rs:17,23)
inline void interpolate(Context &ctx, const std::vector<cf32> &src, std::vector<cf32> &dst, size_t n_between,
                        bool compat_im = true)
{
    if (src.empty()) throw Panic(AETH_E_LEN, "interpolate on an empty src (the reference panics: sampling.rs:23)");
    const size_t add = src.size() + (src.size() - 1) * n_between, old = dst.size();
    dst.resize(old + add);
    size_t written = 0;
    check(aeth_host_interpolate(ctx.get(), raw(src.data()), src.size(), raw(dst.data() + old), add, n_between,
                                compat_im ? 1 : 0, &written));
    dst.resize(old + written);
}

// The reference checks divisibility with a debug_assert_eq! (sampling.rs:32-36): AETHER_REF_RELEASE_BUILD selects the
// semantics of its release build (assert compiled out, ratio floors: what `cargo bench` runs, benches/benches.rs:113,130)
// the way NDEBUG-less / --release selects them for the crate itself; the default mirrors `cargo test` (debug).
#ifdef AETHER_REF_RELEASE_BUILD
constexpr bool kRefReleaseBuild = true;
#else
constexpr bool kRefReleaseBuild = false;
#endif
template <typename T>
inline void downsample(Context &ctx, const std::vector<T> &src, std::vector<T> &dst, bool release = kRefReleaseBuild)
{
    if (release) check(aeth_host_downsample_release(ctx.get(), src.data(), src.size(), dst.data(), dst.size(), sizeof(T), 0));
    else check(aeth_host_downsample(ctx.get(), src.data(), src.size(), dst.data(), dst.size(), sizeof(T)));
}
template <typename T>
inline void downsample_sb(Context &ctx, const std::vector<T> &src, std::vector<T> &dst, bool release = kRefReleaseBuild)
{
    if (release) check(aeth_host_downsample_release(ctx.get(), src.data(), src.size(), dst.data(), dst.size(), sizeof(T), 1));
    else check(aeth_host_downsample(ctx.get(), src.data(), src.size(), dst.data(), dst.size(), sizeof(T)));
}

// ---- assert_evm! (src/lib.rs:26-49), literal, plus a NaN reject ---------------------------
inline void assert_evm(const std::vector<cf32> &actual, const std::vector<cf32> &ref, double evm_limit_db = -80.0)
{
    if (actual.size() != ref.size()) throw Panic(AETH_E_LEN, "Input slices/vectors must be same length");
    if (!(evm_limit_db < 0.0)) throw Panic(AETH_E_ARG, "The EVM threshold must be negative");
    const float fac = (float)std::pow(10.0, evm_limit_db / 10.0);
    for (size_t i = 0; i < actual.size(); i++) {
        const float evm = std::abs(actual[i] - ref[i]);
        const float limit = std::abs(ref[i]) * fac;
        if (evm > limit || std::isnan(actual[i].real()) || std::isnan(actual[i].imag()))
            throw Panic(AETH_E_ARG, "EVM limit exceeded for element " + std::to_string(i));
    }
}

}  // namespace aether
