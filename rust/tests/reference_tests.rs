//! The reference's own unit tests and doctests, replayed through the binding.
//! NOT COMPILED IN THIS PIPELINE (no cargo / rustc in the image or on the GPU box): the same vectors run through the
//! same C ABI from C++ (tests/cpp/test_hostapi.cpp) and Python (tests/test_gpu_*.py).  With a toolchain:
//!     LD_LIBRARY_PATH=../aether_primitives_amd/lib cargo test            (needs one MI355X)
//! Each test cites the reference test it restates.
use aether_hip::{sampling, Context, DeviceVec, Fir, HipFft, PinnedPool};
use aether_primitives::fft::{Fft, Scale};
use aether_primitives::vecops::VecOps;
use aether_primitives::{assert_evm, cf32};

fn dev<'c>(ctx: &'c Context, n: usize, re: f32, im: f32) -> DeviceVec<'c> {
    DeviceVec::from_slice(ctx, &vec![cf32::new(re, im); n])
}

// src/vecops.rs:339-424 -- one test per op, device-resident receiver with the trait's method names
#[test]
fn vecops_known_answers() {
    let ctx = Context::new(0);
    let mut v = dev(&ctx, 100, 0.5, 0.5);
    assert_evm!(v.vec_scale(2.0).to_vec(), vec![cf32::new(1.0, 1.0); 100]);                    // :340-346
    let (mut a, b) = (dev(&ctx, 100, 1.0, 1.0), dev(&ctx, 100, 0.0, 2.0));
    assert_evm!(a.vec_mul(&b).to_vec(), vec![cf32::new(-2.0, 2.0); 100]);                      // :349-357
    let (mut a, b) = (dev(&ctx, 100, 2.0, 2.0), dev(&ctx, 100, 2.0, 0.0));
    assert_evm!(a.vec_div(&b).to_vec(), vec![cf32::new(1.0, 1.0); 100]);                       // :360-367
    let mut a = dev(&ctx, 100, 1.0, 1.0);
    assert_evm!(a.vec_conj().to_vec(), vec![cf32::new(1.0, -1.0); 100]);                       // :370-376
    let (mut a, b) = (dev(&ctx, 100, 1.0, 1.0), dev(&ctx, 100, 1.0, 1.0));
    assert_evm!(a.vec_add(&b).to_vec(), vec![cf32::new(2.0, 2.0); 100]);                       // :379-385
    let (mut a, b) = (dev(&ctx, 100, 2.0, 2.0), dev(&ctx, 100, 1.0, 1.0));
    assert_evm!(a.vec_sub(&b).to_vec(), vec![cf32::new(1.0, 1.0); 100]);                       // :388-393
    let e: Vec<cf32> = (0..4).map(|i| cf32::new(i as f32, 0.0)).collect();
    let mut m = DeviceVec::from_slice(&ctx, &e);
    assert_eq!(m.vec_mirror().to_vec(), vec![e[2], e[3], e[0], e[1]]);                         // :396-405
    let (mut a, b) = (dev(&ctx, 100, 2.0, 2.0), dev(&ctx, 100, 1.0, 1.0));
    assert_evm!(a.vec_clone(&b).to_vec(), vec![cf32::new(1.0, 1.0); 100]);                     // :408-414
    assert_evm!(dev(&ctx, 100, 2.0, 2.0).vec_zero().to_vec(), vec![cf32::default(); 100]);     // :417-424
    let mut x = 0;                                                                              // :427-441, stateful closure
    let lin: Vec<cf32> = (0..100).map(|i| cf32::new(i as f32, i as f32)).collect();
    assert_evm!(dev(&ctx, 100, 1.0, 1.0).vec_mutate(|c| { *c = c.scale(x as f32); x += 1; }).to_vec(), lin);
}

// src/vecops.rs:12-38 -- the chained doctest
#[test]
fn vecops_doctest_chain() {
    let ctx = Context::new(0);
    let (twos, ones) = (dev(&ctx, 100, 2.0, 2.0), dev(&ctx, 100, 1.0, 1.0));
    let mut v = dev(&ctx, 100, 2.0, 2.0);
    v.vec_div(&twos).vec_mul(&twos).vec_zero().vec_add(&ones).vec_sub(&twos).vec_clone(&ones)
        .vec_mutate(|c| c.im = -1.0).vec_conj().vec_mirror();
    assert_evm!(v.to_vec(), vec![cf32::new(1.0, 1.0); 100], -80.0);
}

#[test]
#[should_panic(expected = "Vectors must have same length")]                                     // src/vecops.rs:100-104
fn vec_mul_length_assert() {
    let ctx = Context::new(0);
    let (mut a, b) = (dev(&ctx, 10, 1.0, 1.0), dev(&ctx, 9, 1.0, 1.0));
    a.vec_mul(&b);
}

// src/fft.rs:85-120 -- the Cfft doctest with HipFft behind the same trait, on host slices through the reference's VecOps
#[test]
fn fft_doctest_128_ones() {
    let ctx = Context::new(0);
    let mut fft = HipFft::with_len(&ctx, 128);
    let mut data = vec![cf32::new(1.0, 0.0); 128];
    data.vec_rfft(&mut fft, Scale::None);
    let mut right = vec![cf32::default(); 128];
    right[0] = cf32::new(128.0, 0.0);
    assert_evm!(data, right);                                       // every off-DC bin exactly zero
    fft.ibwd(&mut data, Scale::N);
    assert_evm!(data, vec![cf32::new(1.0, 0.0); 128]);
    data.vec_rfft(&mut fft, Scale::SN).vec_scale(2.0).vec_rifft(&mut fft, Scale::SN);
    assert_evm!(data, vec![cf32::new(2.0, 0.0); 128], -72.0);
    assert_eq!(fft.len(), 128);
}

// src/vecops.rs:443-463 -- round trips at N = 100 (2^2 5^2: mixed radix) with a reused plan
#[test]
fn rfft_rifft_roundtrip_100() {
    let ctx = Context::new(0);
    let mut fft = HipFft::with_len(&ctx, 100);
    let v = vec![cf32::new(1.0, 1.0); 100];
    let mut c = v.clone();
    c.vec_rfft(&mut fft, Scale::SN).vec_rifft(&mut fft, Scale::SN);
    assert_evm!(c, v);
}

#[test]
#[should_panic(expected = "Input and FFT must be the same length")]                              // src/fft.rs:163-167
fn fft_length_assert() {
    let ctx = Context::new(0);
    let mut fft = HipFft::with_len(&ctx, 128);
    let mut x = vec![cf32::new(1.0, 0.0); 127];
    fft.ifwd(&mut x, Scale::None);
}

// src/sampling.rs:72-169
#[test]
fn sampling_known_answers() {
    let ctx = Context::new(0);
    let src: Vec<cf32> = [0.0f32, 3.0, 6.0, 9.0].iter().map(|&x| cf32::new(x, x)).collect();
    let mut dst = Vec::new();
    sampling::interpolate(&ctx, &src, &mut dst, 2);                                              // :72-101
    assert_eq!(dst, (0..10).map(|i| cf32::new(i as f32, i as f32)).collect::<Vec<_>>());
    let src: Vec<i32> = (0..21).collect();
    let mut d = vec![0i32; 7];
    sampling::downsample(&ctx, &src, &mut d);                                                    // :131-144
    assert_eq!(d, (0..7).map(|x| x * 3).collect::<Vec<_>>());
    sampling::downsample_sb(&ctx, &src, &mut d);
    assert_eq!(d, (0..7).map(|x| x * 3).collect::<Vec<_>>());
}

// src/sampling.rs:162-169: `#[should_panic]` holds in a debug build only (debug_assert_eq!); a release build of this
// test crate binds the release entry point, exactly as the reference's own test would stop panicking under --release
#[test]
#[cfg_attr(debug_assertions, should_panic(expected = "Only even decimations are supported"))]
fn downsample_uneven() {
    let ctx = Context::new(0);
    let (s, mut d) = (vec![0i32; 7], vec![0i32; 3]);
    sampling::downsample(&ctx, &s, &mut d);
}

// src/pool.rs:228-296 on pinned elements
#[test]
fn pool_taking_and_making() {
    let ctx = Context::new(0);
    let pool = PinnedPool::make(&ctx, 50, 1, false);
    assert_eq!((pool.len(), pool.cap()), (1, 1));
    {
        let c1 = pool.take();
        assert!(c1.is_some(), "First time checkout failed");
        assert_eq!((pool.len(), pool.cap()), (0, 1));
        assert!(pool.take().is_none(), "Third checkout succeeded when it should have failed");
    }
    assert_eq!((pool.len(), pool.cap()), (1, 1));
    let empty = PinnedPool::make(&ctx, 50, 0, false);
    {
        let _e1 = empty.take_or_make();
        let _e2 = empty.take_or_make();
        assert_eq!((empty.len(), empty.cap()), (0, 2));
    }
    assert_eq!((empty.len(), empty.cap()), (2, 2));
}

// the filter the build defines from benches/benches.rs:410-416: impulse in, taps out; pool elements stream directly
#[test]
fn fir_impulse_and_stream_from_pool_elements() {
    let ctx = Context::new(0);
    let taps: Vec<cf32> = (0..64).map(|k| cf32::new(1.0 / (k as f32 + 1.0), 0.01 * k as f32)).collect();
    let mut fir = Fir::new(&ctx, &taps, 2048);
    let n = 1984 * 50;
    let pool = PinnedPool::make(&ctx, n, 2, false);
    let (mut x, mut y) = (pool.take().unwrap(), pool.take().unwrap());
    x.iter_mut().for_each(|c| *c = cf32::default());
    x[0] = cf32::new(1.0, 0.0);
    let st = fir.filter_stream(&x, &mut y);
    assert_eq!(st.pinned as i32, 3);                                 // both sides copied directly, nothing staged
    assert_evm!(y[..64].to_vec(), taps, -60.0);
    assert!(y[64..].iter().all(|c| c.norm() < 1e-6));
}
