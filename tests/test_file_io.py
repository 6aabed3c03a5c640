"""util::file restated behind the ABI (reference src/util/file.rs:12-107, tests :131-214):
CPU tests (no device needed) + the file -> FIR -> file pipeline on the GPU."""
import numpy as np
import pytest

from helpers import bits_equal, rand_c64
import aether_primitives_amd as ap


def test_binary_roundtrip_and_struct_count(tmp_path):
    from aether_primitives_amd import file as aefile, LengthMismatch
    p = tmp_path / "samples.bin"
    x = rand_c64(1, 1000)
    w = aefile.binary_writer(p)
    w.write(x[:400]); w.write(x[400:])                       # consecutive writes append to the stream
    assert p.stat().st_size == 8000 and aefile.count_structs_in_file(p) == 1000
    assert p.read_bytes() == x.tobytes()                     # header-less native-endian dump (file.rs:101-109)
    r = aefile.binary_reader(p)
    assert bits_equal(r.read_vec(250), x[:250]) and bits_equal(r.read(np.empty(750, np.complex64)), x[250:])
    with pytest.raises(LengthMismatch):                      # read_exact past the end (file.rs:53)
        r.read_vec(1)
    (tmp_path / "odd.bin").write_bytes(b"\0" * 13)
    with pytest.raises(LengthMismatch, match="integer number of the requested struct"):   # file.rs:20-23
        aefile.count_structs_in_file(tmp_path / "odd.bin")
    aefile.binary_writer(p)                                  # re-opening truncates (file.rs:86-88)
    assert p.stat().st_size == 0
    ints = np.arange(77, dtype=np.int32)
    wi = aefile.binary_writer(tmp_path / "ints.bin", np.int32); wi.write(ints)
    assert (aefile.binary_reader(tmp_path / "ints.bin", np.int32).read_vec(77) == ints).all()


@pytest.mark.gpu
def test_file_to_file_fir(ctx, oracle, tmp_path):
    import aether_primitives_amd as ap
    taps = oracle.synth_lowpass_taps(64, 0.25)
    x = oracle.synth_cnormal(815, 3_000_001)
    pin, pout = tmp_path / "in.bin", tmp_path / "out.bin"
    ap.file.binary_writer(pin).write(x)
    f = ap.Fir(ctx, taps, 2048)
    st = f.filter_file(pin, pout, chunk=1 << 20)
    assert st["samples"] == x.size and st["chunks"] == 3
    y = ap.file.binary_reader(pout).read_vec(x.size)
    assert bits_equal(y, f.filter(x))
    assert oracle.evm_db(y, oracle.fir_ols_f32(taps, x, 2048, f.hop, threads=4)) <= -120


@pytest.mark.gpu
def test_file_fir_refuses_to_run_in_place(ctx, oracle, tmp_path):
    """in_path == out_path (also through a hard link) must not truncate the recording"""
    import os
    from aether_primitives_amd import Fir
    x = (np.arange(5000) + 1j).astype(np.complex64)
    p = tmp_path / "iq.bin"; x.tofile(p)
    link = tmp_path / "same.bin"; os.link(p, link)
    f = Fir(ctx, oracle.synth_lowpass_taps(64, 0.25), 2048)
    for out in (p, link):
        with pytest.raises(ap.AetherError, match="same file"):
            f.filter_file(str(p), str(out))
        assert np.array_equal(np.fromfile(p, np.complex64), x)          # untouched
    bad = tmp_path / "odd.bin"; bad.write_bytes(b"\0" * 13)
    with pytest.raises(ap.AetherError):
        f.filter_file(str(bad), str(tmp_path / "o.bin"))
    assert not (tmp_path / "o.bin").exists()                              # nothing created for a bad input


@pytest.mark.gpu
def test_file_to_file_through_the_other_pipeline_stages(ctx, tmp_path):
    """raw cf32 file -> FFT frames / correlate + demod / FFT + interpolate -> raw file (aeth_stream_file): the bytes of
    the device-resident call on the whole recording; a file that is not whole frames is refused before anything is written"""
    import aether_primitives_amd as ap
    from aether_primitives_amd import pipeline, modulation, Scale
    from helpers import rand_c64
    N, frames = 1024, 300
    x = rand_c64(9, N * frames)
    pin = tmp_path / "rx.cf32"; ap.file.binary_writer(pin).write(x)
    fft = ap.HipFft(ctx, N, max_batch=frames)
    sig = ctx.vec(rand_c64(10, N, scale=0.3))
    # FFT frames
    st = pipeline.run_file(pipeline.Stage.fft(fft, Scale.SN), pin, tmp_path / "spec.cf32", chunk=N * 64)
    d = ctx.vec(x); fft.ifwd(d, Scale.SN)
    assert st["samples"] == x.size and st["chunks"] == 5
    assert bits_equal(np.fromfile(tmp_path / "spec.cf32", np.complex64), d.to_host())
    # correlate + demod: a file of bit bytes
    pipeline.run_file(pipeline.Stage.correlate_demod(fft, sig, 2), pin, tmp_path / "bits.u8")
    want = modulation.qpsk(ctx).correlate_demod(fft, ctx.vec(x), sig).to_host()
    assert np.array_equal(np.fromfile(tmp_path / "bits.u8", np.uint8), want)
    # FFT + interpolate: 1 in, 4 out
    pipeline.run_file(pipeline.Stage.fft_interpolate(fft, 3, Scale.SN), pin, tmp_path / "up.cf32")
    o = ctx.empty((N + (N - 1) * 3) * frames); fft.rfft_interpolate(ctx.vec(x), o, 3, Scale.SN)
    assert bits_equal(np.fromfile(tmp_path / "up.cf32", np.complex64), o.to_host())
    ragged = tmp_path / "ragged.cf32"; x[: N * 3 + 5].tofile(ragged)
    with pytest.raises(ap.AetherError):
        pipeline.run_file(pipeline.Stage.fft(fft, Scale.SN), ragged, tmp_path / "never.cf32")
    with pytest.raises(ap.AetherError, match="same file"):
        pipeline.run_file(pipeline.Stage.fft(fft, Scale.SN), pin, pin)
    assert np.array_equal(np.fromfile(pin, np.complex64).view(np.uint32), x.view(np.uint32))
