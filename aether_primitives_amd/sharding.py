"""Partitioning of the hot path across GPUs (one process per GPU, no collective).

Every unit on this path is independent of its neighbours -- FFT frames
(`chunks_mut(fft_len)`, reference src/util/plot.rs:59-61), channels, and
overlap-save blocks -- so N GPUs simply take contiguous ranges.  Nothing here
touches a device; these are the index computations the bench and the
multi-process tests share.
"""


def frame_shard(n_frames, rank, world):
    """Contiguous block partition of the frame index: rank g gets [g*F/G, (g+1)*F/G)."""
    lo = n_frames * rank // world
    hi = n_frames * (rank + 1) // world
    return lo, hi - lo


def fir_shard(n_samples, hop, ntaps, rank, world):
    """Output range of one long FIR stream owned by `rank`, aligned to the
    overlap-save hop so that every shard runs exactly the blocks the 1-GPU run
    would (=> bit-identical outputs for any world size).

    Returns dict(out_lo, out_hi, in_lo, hist_lo): the shard filters
    x[in_lo:out_hi] with history x[hist_lo:in_lo] (ntaps-1 samples; empty for the
    first shard = zero initial state) and produces y[out_lo:out_hi].  The history
    is read from the source when staging the shard -- it is never exchanged
    between GPUs."""
    n_blocks = (n_samples + hop - 1) // hop
    b_lo, nb = frame_shard(n_blocks, rank, world)
    out_lo = min(b_lo * hop, n_samples)
    out_hi = min((b_lo + nb) * hop, n_samples)
    hist_lo = max(out_lo - (ntaps - 1), 0) if out_lo > 0 else 0
    return {"out_lo": out_lo, "out_hi": out_hi, "in_lo": out_lo, "hist_lo": hist_lo}


def channel_of(rank, world, n_channels):
    """BASELINE config 4: independent channels, channel c -> GPU c % world."""
    return [c for c in range(n_channels) if c % world == rank]
