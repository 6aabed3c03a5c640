#!/usr/bin/env python3
"""The reference's modem example (examples/modem.rs:11-36) on the device: random bits -> QPSK ->
AWGN (power 0.01, seed 815) -> hard demodulation, counted bit errors instead of gnuplot windows.

The reference's `assert_eq!(b, bits)` cannot hold there: its QPSK demod writes `idx & 2` (0 or 2) for
the second bit (src/modulation.rs:54).  compat=True reproduces that output, compat=False gives the
bit itself; both are shown."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import modulation, noise


def main(nbits=1 << 20, power=0.01, seed=815):
    ctx = ap.Context(0)
    m = modulation.qpsk(ctx)                                        # modulation::qpsk()
    bits = np.random.default_rng(seed).integers(0, 2, nbits, dtype=np.uint8)
    tx = m.modulate(modulation.DeviceBits(ctx, nbits, bits))        # m.modulate(&b)
    noise.new(ctx, power, seed).apply(tx)                           # noise::new(0.01, 815).apply(&mut output)
    strict = m.demod_naive(tx, compat=True).to_host()               # m.demod_naive(..): the reference's bytes
    plain = m.demod_naive(tx, compat=False).to_host()
    errors = int((plain != bits).sum())
    quirk = bool(((strict[1::2] >> 1) == bits[1::2]).all() and (strict[0::2] == bits[0::2]).all())
    print(f"{nbits} bits over QPSK + AWGN(power {power}): {errors} bit errors; "
          f"reference-compatible output carries bit 1 as {{0, 2}}: {quirk}")
    # the same modem over HOST memory as a pipeline (src/pipeline.rs): bit bytes up, modulate + AWGN -> demodulate on the
    # device as two chained stages, bit bytes down -- the symbols never cross PCIe.  (The stage with a table of its own
    # would be the correlator in front of the demodulator; here the correlator's reference is a single 1, i.e. identity.)
    from aether_primitives_amd import pipeline
    N = 2048
    if nbits % (2 * N) == 0:
        fft = ap.HipFft(ctx, N, max_batch=64)
        one = np.zeros(N, np.complex64); one[:] = 1.0 / N              # fwd . (1/N) . bwd = identity
        sig = ctx.vec(one)
        rx, st = pipeline.run_chain([pipeline.Stage.modulate_awgn(ctx, 2, power, seed), pipeline.Stage.correlate_demod(fft, sig, 2, compat=False)], bits)
        print(f"as a host pipeline: {int((rx != bits).sum())} bit errors, {st['samples'] / st['seconds'] / 1e9:.2f} Gbit-bytes/s in")
    return errors


if __name__ == "__main__":
    main()
