#!/usr/bin/env python3
"""The spectrum chain behind the reference's waterfall plot (examples/plotting.rs:41-55,
src/util/plot.rs:59-61): noise -> chunks of fft_len -> vec_rfft(Scale::SN) -> vec_mirror, here as
two batched launches over all 500 frames; prints the mean power per bin instead of plotting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import Scale, noise


def main(fft_len=2048, frames=500):
    ctx = ap.Context(0)
    x = ctx.empty(fft_len * frames).vec_zero()
    noise.new(ctx, 1.0, 815).apply(x)                               # noise::new(1.0, 815).iter().take(..)
    fft = ap.HipFft(ctx, fft_len, max_batch=frames)
    x.vec_rfft(fft, Scale.SN).vec_mirror_frames(fft_len)            # c.vec_rfft(&mut fft, Scale::SN).vec_mirror()
    spec = x.to_host().reshape(frames, fft_len)
    db = 10 * np.log10((np.abs(spec) ** 2).mean(axis=0))
    print(f"{frames} x {fft_len}: mean power per bin {db.mean():.2f} dB (flat within {db.max() - db.min():.2f} dB)")
    return db


if __name__ == "__main__":
    main()
