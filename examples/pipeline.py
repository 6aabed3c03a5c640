#!/usr/bin/env python3
"""The reference's pipeline example (examples/pipeline.rs, src/pipeline.rs:52-137, src/pool.rs:43-221)
is a thread per stage over pooled buffers.  Its device counterpart: a host-resident stream goes
through copy-in | upload | compute | download | copy-out -- three HIP streams over three pooled device slots,
host threads for the two copy stages -- with the 64-tap FIR as the compute stage, from memory and from a raw
sample file (src/util/file.rs format), then with another stage (`pipeline::new().add_stage(..)`, pipeline.rs:24-41:
the correlator chain followed by the QPSK demodulator, 8 bytes in and 2 bytes out per sample); prints the
PCIe-inclusive rates and the reference's per-stage report."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import file as afile, pipeline, modulation
from bench import synth_stream, lowpass_taps


def main(n=1 << 24):
    ctx = ap.Context(0)
    fir = ap.Fir(ctx, lowpass_taps(), 2048)
    x = synth_stream(815, n)
    y, stats = fir.filter_stream(x)                                  # memory -> memory
    print(f"stream of {n} samples: {stats['samples'] / stats['seconds'] / 1e9:.2f} GS/s PCIe-inclusive, "
          f"{int(stats['chunks'])} chunks")
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.cf32"), os.path.join(d, "out.cf32")
        afile.binary_writer(src).write(x)                            # util::file::binary_writer
        fstats = fir.filter_file(src, dst)
        z = afile.binary_reader(dst).read_vec(afile.count_structs_in_file(dst))
    same = bool((y.view(np.uint32) == z.view(np.uint32)).all())
    print(f"file -> file: {fstats['samples'] / fstats['seconds'] / 1e9:.2f} GS/s, identical to the in-memory run: {same}")
    # another compute stage: received frames of 2048 -> correlate with a reference signal -> hard QPSK decisions
    fft = ap.HipFft(ctx, 2048, max_batch=4096)
    sig = ctx.vec(np.conj(synth_stream(7, 2048)))
    frames = x[: (n // 2048) * 2048]
    bits, st = pipeline.run(pipeline.Stage.correlate_demod(fft, sig, 2), frames, report=True)
    want = modulation.qpsk(ctx).correlate_demod(fft, ctx.vec(frames), sig).to_host()
    same2 = bool(np.array_equal(bits, want))
    print(f"correlate + demod stage: {st['samples'] / st['seconds'] / 1e9:.2f} GS/s in, {bits.size} bit bytes out, "
          f"identical to the device-resident call: {same2}")
    for line in st["lines"]: print("   " + line)
    return same and same2


if __name__ == "__main__":
    main()
