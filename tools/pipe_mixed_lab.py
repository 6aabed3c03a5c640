#!/usr/bin/env python3
"""Why is a pageable input next to a pinned output the slow combination of the host pipeline (round 3: downloads 2.4 x
as long, cause not found)?  One process per setting (the knobs are read once): the FIR over 256 Mi samples,
pageable numpy input, output in a 2 GiB pool element,
   staged      the default: the output goes through the host stage too
   mixed/T     AETH_PIPE_MIXED=1 -- downloads straight into the caller's pinned slice -- with T copy threads
and the per-stage report of each.  If the downloads of `mixed` speed up as the copy-in threads are taken away, the
two compete for host memory bandwidth (the copy-in reads 2 GiB of pageable memory and writes the staging elements while
the D2H engine writes 2 GiB of fresh lines); if not, the cause is elsewhere.
   python3 tools/pipe_mixed_lab.py            (driver: spawns the settings)   -> gpurun_out/pipe_mixed_lab.txt"""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    import aether_primitives_amd as ap
    from aether_primitives_amd import pool
    from bench import lowpass_taps, FFT_LEN
    n = 256 << 20
    ctx = ap.Context(0)
    fir = ap.Fir(ctx, lowpass_taps(), FFT_LEN)
    x = np.random.default_rng(1).standard_normal(2 * n, dtype=np.float32).view(np.complex64)
    pp = pool.Pool(ctx, n * 8, initial_len=1)
    e = pp.take()
    y = e.array(np.complex64)
    best = None
    for rep in range(3):
        _, st = fir.filter_stream(x, out=y, report=True)
        if best is None or st["seconds"] < best["seconds"]: best = st
    print(f"{sys.argv[2]:14s} {best['seconds'] * 1e3:8.1f} ms  {n / best['seconds'] / 1e9:5.2f} GS/s  pinned={int(best['pinned'])}  "
          f"busy: copy-in {best['active_copy_in'] * 1e3:6.1f}  upload {best['active_upload'] * 1e3:6.1f}  kernel {best['active_kernel'] * 1e3:5.1f}  "
          f"download {best['active_download'] * 1e3:6.1f}  copy-out {best['active_copy_out'] * 1e3:6.1f} ms", flush=True)
    y = _ = None; e.close(); pp.close()
    sys.exit(0)

settings = [("staged", {})] + [(f"mixed/{t}", {"AETH_PIPE_MIXED": "1", "AETH_PIPE_THREADS": str(t)}) for t in (12, 6, 3, 2, 1)] + \
           [(f"staged/{t}", {"AETH_PIPE_THREADS": str(t)}) for t in (6, 2)]
for name, env in settings:
    e = dict(os.environ, AETH_TUNING="1", **env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], env=e, cwd=ROOT)
