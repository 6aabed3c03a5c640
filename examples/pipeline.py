#!/usr/bin/env python3
"""The reference's pipeline example (examples/pipeline.rs, src/pipeline.rs:52-137, src/pool.rs:43-221)
is a thread per stage over pooled buffers.  Its device counterpart: a host-resident stream goes
through the 64-tap FIR as H2D | kernel | D2H on two HIP streams over a pair of pooled device slots,
from memory and from a raw sample file (src/util/file.rs format); prints the PCIe-inclusive rate."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import aether_primitives_amd as ap
from aether_primitives_amd import file as afile
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
    return same


if __name__ == "__main__":
    main()
