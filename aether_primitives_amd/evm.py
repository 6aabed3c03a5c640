"""assert_evm! (reference: src/lib.rs:26-49) for host arrays, plus the
conventional aggregate EVM.  Pure comparison helpers (numpy); not on the data path."""
import numpy as np


def assert_evm(actual, ref, evm_limit_db=-80.0):
    """Literal macro: per element |act-ref| <= |ref| * (10^(dB/10) as f32), in f32."""
    a = np.asarray(actual, dtype=np.complex64).reshape(-1)
    r = np.asarray(ref, dtype=np.complex64).reshape(-1)
    assert a.size == r.size, "Input slices/vectors must be same length"
    assert float(evm_limit_db) < 0.0, "The EVM threshold must be negative"
    fac = np.float32(10.0 ** (float(evm_limit_db) / 10.0))
    d = a - r
    evm = np.hypot(d.real, d.imag).astype(np.float32)
    lim = (np.hypot(r.real, r.imag).astype(np.float32) * fac).astype(np.float32)
    bad = np.nonzero((evm > lim) | np.isnan(a.real) | np.isnan(a.imag))[0]
    if bad.size:
        i = int(bad[0])
        raise AssertionError(f"EVM limit exceeded: {evm[i]} > {lim[i]}({evm_limit_db}dB) for element {i}. "
                             f"Actual {a[i]}, Expected {r[i]}")


def evm_db(actual, ref):
    """20*log10(||act-ref||_2 / ||ref||_2); -inf when identical."""
    a = np.asarray(actual).reshape(-1).astype(np.complex128)
    r = np.asarray(ref).reshape(-1).astype(np.complex128)
    pe = float(np.sum(np.abs(a - r) ** 2)); pr = float(np.sum(np.abs(r) ** 2))
    if pe == 0.0:
        return float("-inf")
    if pr == 0.0:
        return float("inf")
    return 10.0 * np.log10(pe / pr)
