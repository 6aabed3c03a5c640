// C++ host-API test: the reference's own unit tests and doctests, replayed through
// include/aether_hip.hpp (the C++ mirror of trait VecOps / trait Fft) on the GPU.
// Each block cites the reference test it restates.  Exit code 0 = all passed.
#include <cstdio>
#include <cstring>

#include "../../include/aether_hip.hpp"

using namespace aether;
static int failures = 0;
#define RUN(name, ...)                                                          \
    do {                                                                        \
        try { __VA_ARGS__; std::printf("ok   %s\n", name); }                    \
        catch (const std::exception &e) { failures++; std::printf("FAIL %s: %s\n", name, e.what()); } \
    } while (0)
#define EXPECT_PANIC(name, msg, ...)                                            \
    do {                                                                        \
        bool got = false;                                                       \
        try { __VA_ARGS__; } catch (const Panic &p) { got = std::strstr(p.what(), msg) != nullptr; } \
        if (got) std::printf("ok   %s\n", name); else { failures++; std::printf("FAIL %s: expected panic '%s'\n", name, msg); } \
    } while (0)

int main()
{
    Context ctx(0);
    auto vec = [](size_t n, float re, float im) { return std::vector<cf32>(n, cf32(re, im)); };

    // src/vecops.rs:339-424 -- one test per op, device-resident receiver
    RUN("vec_scale", { DeviceVec v(ctx, vec(100, .5f, .5f)); v.vec_scale(2.0f); assert_evm(v.to_host(), vec(100, 1, 1)); });
    RUN("vec_mul", { DeviceVec a(ctx, vec(100, 1, 1)), b(ctx, vec(100, 0, 2)); a.vec_mul(b); assert_evm(a.to_host(), vec(100, -2, 2)); });
    RUN("vec_div", { DeviceVec a(ctx, vec(100, 2, 2)), b(ctx, vec(100, 2, 0)); a.vec_div(b); assert_evm(a.to_host(), vec(100, 1, 1)); });
    RUN("vec_conj", { DeviceVec a(ctx, vec(100, 1, 1)); a.vec_conj(); assert_evm(a.to_host(), vec(100, 1, -1)); });
    RUN("vec_add", { DeviceVec a(ctx, vec(100, 1, 1)), b(ctx, vec(100, 1, 1)); a.vec_add(b); assert_evm(a.to_host(), vec(100, 2, 2)); });
    RUN("vec_sub", { DeviceVec a(ctx, vec(100, 2, 2)), b(ctx, vec(100, 1, 1)); a.vec_sub(b); assert_evm(a.to_host(), vec(100, 1, 1)); });
    RUN("vec_mirror", {
        std::vector<cf32> e = {{0, 0}, {1, 0}, {2, 0}, {3, 0}};
        DeviceVec a(ctx, e); a.vec_mirror();
        assert_evm(a.to_host(), std::vector<cf32>{{2, 0}, {3, 0}, {0, 0}, {1, 0}});
    });
    RUN("vec_clone", { DeviceVec a(ctx, vec(100, 2, 2)), b(ctx, vec(100, 1, 1)); a.vec_clone(b); assert_evm(a.to_host(), vec(100, 1, 1)); });
    RUN("vec_zero", { DeviceVec a(ctx, vec(100, 2, 2)); a.vec_zero(); assert_evm(a.to_host(), vec(100, 0, 0)); });
    // src/vecops.rs:426-441 -- stateful closure
    RUN("vec_mutate", {
        DeviceVec a(ctx, vec(100, 1, 1)); int x = 0;
        a.vec_mutate([&](cf32 &c) { c *= (float)x; x++; });
        std::vector<cf32> lin(100); for (int i = 0; i < 100; i++) lin[i] = cf32((float)i, (float)i);
        assert_evm(a.to_host(), lin);
    });
    // src/vecops.rs:12-38 -- the chained doctest
    RUN("vecops doctest chain", {
        DeviceVec v(ctx, vec(100, 2, 2)), twos(ctx, vec(100, 2, 2)), ones(ctx, vec(100, 1, 1));
        v.vec_div(twos).vec_mul(twos).vec_zero().vec_add(ones).vec_sub(twos).vec_clone(ones)
            .vec_mutate([](cf32 &c) { c.imag(-1.0f); }).vec_conj().vec_mirror();
        assert_evm(v.to_host(), vec(100, 1, 1), -80.0);
    });
    // the same chain on a host slice (the literal `impl VecOps for Vec<cf32>`)
    RUN("vecops doctest chain (host slice)", {
        auto v = vec(100, 2, 2); auto twos = v; auto ones = vec(100, 1, 1);
        HostVec(ctx, v).vec_div(twos).vec_mul(twos).vec_zero().vec_add(ones).vec_sub(twos).vec_clone(ones)
            .vec_mutate([](cf32 &c) { c.imag(-1.0f); }).vec_conj().vec_mirror();
        assert_evm(v, ones, -80.0);
    });
    EXPECT_PANIC("vec_mul length assert", "Vectors must have same length",
                 { DeviceVec a(ctx, vec(10, 1, 1)), b(ctx, vec(9, 1, 1)); a.vec_mul(b); });

    // src/fft.rs:243-269 -- Scale
    RUN("scale", {
        if (Scale::SN().factor(4) != 0.5f || Scale::N().factor(4) != 0.25f || Scale::X(2).factor(4) != 2.0f ||
            Scale::None().factor(4) != 1.0f) throw Panic(0, "scale factors");
    });
    // src/fft.rs:85-120 -- the Cfft doctest through the trait object
    RUN("fft doctest (128 ones)", {
        auto data = vec(128, 1, 0);
        HostVec(ctx, data).vec_fft(Scale::None());
        auto right = vec(128, 0, 0); right[0] = cf32(128, 0);
        assert_evm(data, right);                                   // off-DC bins exactly zero
        HipFft f = HipFft::with_len(ctx, 128);
        Fft &dyn = f;
        dyn.ibwd(data.data(), data.size(), Scale::N());
        assert_evm(data, vec(128, 1, 0));
        HostVec(ctx, data).vec_rfft(f, Scale::SN()).vec_scale(2.0f).vec_rifft(f, Scale::SN());
        assert_evm(data, vec(128, 2, 0), -72);
        if (dyn.len() != 128) throw Panic(0, "len");
    });
    // src/vecops.rs:443-463 -- round trips at N = 100
    RUN("vec_fft / vec_rfft round trip N=100", {
        auto v = vec(100, 1, 1);
        DeviceVec c(ctx, v); c.vec_fft(Scale::SN()).vec_ifft(Scale::SN()); assert_evm(c.to_host(), v);
        DeviceVec d(ctx, v); HipFft f(ctx, 100); d.vec_rfft(f, Scale::SN()).vec_rifft(f, Scale::SN()); assert_evm(d.to_host(), v);
    });
    RUN("tfwd lends the plan's temp", {
        auto x = vec(64, 1, 0); HipFft f(ctx, 64);
        const cf32 *t = f.tfwd(x.data(), x.size(), Scale::None());
        if (t[0] != cf32(64, 0) || t[1] != cf32(0, 0)) throw Panic(0, "tfwd values");
    });
    EXPECT_PANIC("fft length assert", "Input and FFT must be the same length",
                 { HipFft f(ctx, 128); auto x = vec(127, 1, 0); f.ifwd(x.data(), x.size(), Scale::None()); });

    // src/sampling.rs:72-169
    RUN("interpolate 2 between (appends)", {
        std::vector<cf32> src = {{0, 0}, {3, 3}, {6, 6}, {9, 9}}, dst = {{7, 7}};
        interpolate(ctx, src, dst, 2);
        if (dst.size() != 11) throw Panic(0, "len");
        for (int i = 0; i < 10; i++) if (dst[i + 1] != cf32((float)i, (float)i)) throw Panic(0, "value");
    });
    RUN("downsample 21 -> 7 (generic T)", {
        std::vector<int> src(21), dst(7); for (int i = 0; i < 21; i++) src[i] = i;
        downsample(ctx, src, dst);
        for (int i = 0; i < 7; i++) if (dst[i] != 3 * i) throw Panic(0, "value");
        downsample_sb(ctx, src, dst);
    });
    EXPECT_PANIC("downsample 7 -> 3", "Only even decimations are supported",
                 { std::vector<int> s(7), d(3); downsample(ctx, s, d); });
    // the same call as the reference's RELEASE build runs it (debug_assert compiled out): benches/benches.rs:113,130
    RUN("downsample 8096 -> 512, release-build semantics (dec = 15)", {
        std::vector<cf32> src(8096), dst(512);
        for (int i = 0; i < 8096; i++) src[i] = cf32((float)i, -(float)i);
        downsample(ctx, src, dst, true);
        for (int i = 0; i < 512; i++) if (dst[i] != src[15 * i]) throw Panic(0, "value");
        std::fill(dst.begin(), dst.end(), cf32(0, 0));
        downsample_sb(ctx, src, dst, true);
        for (int i = 0; i < 512; i++) if (dst[i] != src[15 * i]) throw Panic(0, "value (step_by)");
    });

    // FIR: impulse in, taps out
    RUN("fir impulse response", {
        std::vector<cf32> taps(64); for (int k = 0; k < 64; k++) taps[k] = cf32(1.0f / (k + 1), 0.01f * k);
        std::vector<cf32> x(5000, cf32(0, 0)), y; x[0] = cf32(1, 0);
        Fir fir(ctx, taps, 2048); fir.filter(x, y);
        for (int k = 0; k < 64; k++) if (std::abs(y[k] - taps[k]) > 1e-6f) throw Panic(0, "taps");
        for (int k = 64; k < 5000; k++) if (std::abs(y[k]) > 1e-6f) throw Panic(0, "tail");
    });

    // host-resident stream through the three-stage pipeline: the bits of the one-call flavour, and the stage report
    RUN("fir stream pipeline with the per-stage report", {
        std::vector<cf32> taps(64); for (int k = 0; k < 64; k++) taps[k] = cf32(1.0f / (k + 1), 0.01f * k);
        const size_t n = 1984 * 700 + 17;
        std::vector<cf32> x(n), y, z;
        for (size_t i = 0; i < n; i++) x[i] = cf32(std::sin(0.01f * i), std::cos(0.013f * i));
        Fir fir(ctx, taps, 2048);
        fir.filter(x, y);
        const aeth_pipe_util u = fir.filter_stream(x, z, 1984 * 100, true);
        if (z.size() != n || std::memcmp(y.data(), z.data(), n * sizeof(cf32)) != 0) throw Panic(0, "stream != one call");
        if (u.chunks != 8 || u.samples != (double)n || !(u.seconds > 0)) throw Panic(0, "stats");
        if (!(u.active_upload > 0 && u.active_kernel > 0 && u.active_download > 0)) throw Panic(0, "stage times");
    });

    // src/pool.rs:228-296 on pinned elements, then a stream produced into pool elements (copied directly, no staging)
    RUN("pool: taking, taking_or_making, stream from pool elements", {
        Pool pool = Pool::make(ctx, 1984 * 50, 1);
        if (pool.len() != 1 || pool.cap() != 1) throw Panic(0, "make");
        {
            Pool::Elem c1 = pool.take();
            if (!c1 || pool.len() != 0 || pool.cap() != 1) throw Panic(0, "First time checkout failed");
            Pool::Elem c2 = pool.take();
            if (c2) throw Panic(0, "Third checkout succeeded when it should have failed");
        }
        if (pool.len() != 1 || pool.cap() != 1) throw Panic(0, "drop returns the element");
        Pool::Elem a = pool.take_or_make(), b = pool.take_or_make();
        if (pool.len() != 0 || pool.cap() != 2) throw Panic(0, "take_or_make grows the pool");
        std::vector<cf32> taps(64); for (int k = 0; k < 64; k++) taps[k] = cf32(1.0f / (k + 1), 0.01f * k);
        const size_t n = a.size();
        std::vector<cf32> x(n), y;
        for (size_t i = 0; i < n; i++) { x[i] = cf32(std::sin(0.01f * i), std::cos(0.013f * i)); a[i] = x[i]; }
        Fir fir(ctx, taps, 2048);
        fir.filter(x, y);
        const aeth_pipe_util u = fir.filter_stream(a.data(), n, b.data(), 1984 * 10);
        if (u.pinned != 3 || u.active_copy_in != 0 || u.active_copy_out != 0) throw Panic(0, "pool elements must be copied directly");
        if (std::memcmp(y.data(), b.data(), n * sizeof(cf32)) != 0) throw Panic(0, "stream != one call");
    });

    // overlap lane + decimating store: consecutive independent launches on two queues give the bits of the plain run
    RUN("fir on the overlap lane, and with a decimating store", {
        std::vector<cf32> taps(64); for (int k = 0; k < 64; k++) taps[k] = cf32(1.0f / (k + 1), 0.01f * k);
        const size_t n = 1984 * 40;
        std::vector<cf32> a(n), b(n);
        for (size_t i = 0; i < n; i++) { a[i] = cf32(std::sin(0.01f * i), std::cos(0.013f * i)); b[i] = cf32(std::cos(0.02f * i), 0.5f); }
        Fir fir(ctx, taps, 2048);
        DeviceVec da(ctx, a), db(ctx, b), ya(ctx, n), yb(ctx, n), ra(ctx, n), rb(ctx, n);
        fir.filter(da, ra); fir.filter(db, rb);
        const std::vector<cf32> wa = ra.to_host(), wb = rb.to_host();
        ctx.set_overlap(true);
        for (int rep = 0; rep < 4; rep++) { fir.filter(da, ya); fir.filter(db, yb); }
        const std::vector<cf32> ga = ya.to_host(), gb = yb.to_host();
        ctx.set_overlap(false);
        if (std::memcmp(ga.data(), wa.data(), n * sizeof(cf32)) || std::memcmp(gb.data(), wb.data(), n * sizeof(cf32))) throw Panic(0, "overlap lane changed the output");
        DeviceVec yd(ctx, n / 8);
        fir.filter_decim(da, yd);
        const std::vector<cf32> gd = yd.to_host();
        for (size_t i = 0; i < n / 8; i++) if (std::memcmp(&gd[i], &wa[8 * i], sizeof(cf32))) throw Panic(0, "decimated output");
    });
    EXPECT_PANIC("fir decimation 7000 -> 2333", "Only even decimations are supported", {
        std::vector<cf32> taps(8, cf32(0.125f, 0)); Fir fir(ctx, taps, 2048);
        DeviceVec x(ctx, 7000), y(ctx, 2333); fir.filter_decim(x, y);
    });

    // src/pipeline.rs:24-41,123-137 -- a pipeline whose compute stage is one of the device ops: host slices in, host
    // slices out, bit-identical to the device flavour on the whole slice
    RUN("pipeline: fft frames, correlator chain, correlate + demod, fft + interpolate", {
        const size_t N = 1024, frames = 96, n = N * frames;
        std::vector<cf32> x(n), sigh(N);
        for (size_t i = 0; i < n; i++) x[i] = cf32(std::sin(0.37f * i), std::cos(0.11f * i));
        for (size_t i = 0; i < N; i++) sigh[i] = cf32(1.0f / (1 + i % 7), 0.25f * (i % 3));
        HipFft fft(ctx, N, frames);
        DeviceVec sig(ctx, sigh);
        {   // FFT frames
            DeviceVec d(ctx, x); fft.ifwd(d, Scale::SN());
            const std::vector<cf32> y = pipeline::stage_fft(fft, Scale::SN()).run(ctx, x, N * 10);
            if (y.size() != n || std::memcmp(y.data(), d.to_host().data(), n * sizeof(cf32)) != 0) throw Panic(0, "fft stream != device flavour");
        }
        {   // vec_rfft -> vec_mul -> vec_rifft per frame (benches.rs:410-416)
            DeviceVec d(ctx, x); fft.mul_chain(d, sig, Scale::None(), Scale::N());
            const std::vector<cf32> y = pipeline::stage_mul_chain(fft, sig, Scale::None(), Scale::N()).run(ctx, x, N * 7);
            if (std::memcmp(y.data(), d.to_host().data(), n * sizeof(cf32)) != 0) throw Panic(0, "chain stream != device flavour");
        }
        {   // ... then demod_naive: 8 B in, 2 bytes out per sample
            const pipeline::Stage st = pipeline::stage_correlate_demod(fft, sig, 2);
            if (st.out_count(ctx, n) != 2 * n) throw Panic(0, "out_count");
            const std::vector<uint8_t> b = st.run_bits(ctx, x, N * 16);
            std::vector<uint8_t> want(2 * n);
            void *dbits = nullptr; check(aeth_dev_alloc(ctx.get(), 2 * n, &dbits));
            DeviceVec d(ctx, x);
            check(aeth_fft_mul_ifft_demod(fft.get(), d.ptr(), n, frames, sig.ptr(), N, 0, 0.f, 0, 0.f, 2, nullptr, (uint8_t *)dbits, 2 * n, 1));
            check(aeth_download(ctx.get(), want.data(), dbits, 2 * n)); check(aeth_dev_free(ctx.get(), dbits));
            if (b != want) throw Panic(0, "demod stream != device flavour");
        }
        {   // the transform, then sampling::interpolate per frame: 1 in, n_between + 1 out
            const size_t nb = 4, out_n = (N + (N - 1) * nb) * frames;
            const pipeline::Stage st = pipeline::stage_fft_interpolate(fft, nb, Scale::SN());
            if (st.out_count(ctx, n) != out_n) throw Panic(0, "out_count");
            const std::vector<cf32> y = st.run(ctx, x, N * 9);
            DeviceVec d(ctx, x), o(ctx, out_n); size_t wrote = 0;
            check(aeth_fft_exec_interpolate(fft.get(), d.ptr(), n, frames, HipFft::kFwdSign, AETH_SCALE_SN, 0.f, o.ptr(), out_n, nb, 1, &wrote));
            if (wrote != out_n || std::memcmp(y.data(), o.to_host().data(), out_n * sizeof(cf32)) != 0) throw Panic(0, "interpolate stream != device flavour");
        }
        EXPECT_PANIC("pipeline: ragged input", "Input and FFT must be the same length",
                     { std::vector<cf32> r(N + 1), o(N + 1); pipeline::stage_fft(fft, Scale::None()).run(ctx, r.data(), r.size(), o.data(), o.size()); });
        check(aeth_ctx_trim(ctx.get()));
    });

    std::printf("%s (%d failure%s)\n", failures ? "FAILED" : "PASSED", failures, failures == 1 ? "" : "s");
    return failures ? 1 : 0;
}
