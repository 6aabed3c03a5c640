"""ctypes loader for the CPU oracle (oracle/libaeth_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py.  The product package
(aether_primitives_amd) must never import this module.

numpy complex64 arrays are bit-identical to the oracle's orc_cf32[] (and to
the reference's cf32 slices, src/lib.rs:8-12), complex128 to orc_cf64[].
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaeth_oracle.so")

SCALE_NONE, SCALE_SN, SCALE_N, SCALE_X = 0, 1, 2, 3
SIGN_REF_FWD, SIGN_REF_BWD = +1, -1


def build(force=False):
    # AETH_ORACLE_SO: load this build of the same sources instead (the sanitizer build of `make -C oracle asan`,
    # tests/test_hostcore_sanitizers.py); nothing is rebuilt then
    global _SO
    alt = os.environ.get("AETH_ORACLE_SO")
    if alt:
        _SO = alt
        return _SO
    srcs = [os.path.join(_HERE, f) for f in ("aeth_oracle.c", "aeth_oracle.h", "fft_template.inc", "awgn_restatement.inc")]
    have_src = all(os.path.exists(s) for s in srcs)
    stale = (not os.path.exists(_SO)) or (
        have_src and os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_native = None


def native_fir_lib():
    """Timing build of the same restatement for bench.py's cpu_baseline leg (SURVEY 8d / BASELINE.md 2):
    `gcc -O3 -march=native -ffp-contract=off`, compiled ON the machine that runs it (oracle/_native/, ignored by
    git and by gpurun).  Returns (CDLL, flags) or (None, reason) -- the caller then times the portable build."""
    global _native
    if _native is not None:
        return _native
    out_dir = os.path.join(_HERE, "_native")
    so = os.path.join(out_dir, "libaeth_oracle_native.so")
    flags = ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=gnu11"]
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", *flags, "-shared", "-o", so, os.path.join(_HERE, "aeth_oracle.c"), "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_fir_ols_f32_mt.restype = C.c_int
        L.orc_fir_ols_f32_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        _native = (L, " ".join(flags))
    except Exception as e:      # no compiler on the box, read-only tree ...
        _native = (None, f"native build failed: {e}")
    return _native


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        vp, sz, f32, i32, f64 = C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_double
        sigs = {
            "orc_vec_scale": (None, [vp, sz, f32]),
            "orc_vec_mul": (i32, [vp, sz, vp, sz]),
            "orc_vec_div": (i32, [vp, sz, vp, sz]),
            "orc_vec_conj": (None, [vp, sz]),
            "orc_vec_add": (i32, [vp, sz, vp, sz]),
            "orc_vec_sub": (i32, [vp, sz, vp, sz]),
            "orc_vec_mirror": (None, [vp, sz]),
            "orc_vec_clone": (i32, [vp, sz, vp, sz]),
            "orc_vec_zero": (None, [vp, sz]),
            "orc_scale_factor": (f32, [i32, sz, f32]),
            "orc_scale_apply": (None, [i32, f32, vp, sz]),
            "orc_fft_plan_create": (vp, [sz]),
            "orc_fft_plan_destroy": (None, [vp]),
            "orc_fft_plan_len": (sz, [vp]),
            "orc_fft_process": (None, [vp, vp, vp, i32]),
            "orc_cfft_outofplace": (i32, [vp, vp, sz, vp, i32, i32, f32]),
            "orc_cfft_inplace": (i32, [vp, vp, sz, i32, i32, f32]),
            "orc_cfft_tmp": (vp, [vp, vp, sz, i32, i32, f32]),
            "orc_fft_f64": (None, [vp, vp, sz, i32]),
            "orc_dft_naive_f64": (None, [vp, vp, sz, i32]),
            "orc_fir_direct_f64": (None, [vp, sz, vp, vp, sz, vp]),
            "orc_fir_ols_f32": (i32, [vp, sz, sz, sz, vp, vp, sz, vp]),
            "orc_fir_ols_f32_mt": (i32, [vp, sz, sz, sz, vp, sz, vp, i32]),
            "orc_correlate_frames": (i32, [vp, sz, vp, sz]),
            "orc_interpolate": (sz, [vp, sz, vp, sz, i32]),
            "orc_downsample": (i32, [vp, sz, vp, sz, sz]),
            "orc_downsample_release": (i32, [vp, sz, vp, sz, sz, i32]),
            "orc_assert_evm": (C.c_long, [vp, sz, vp, sz, f64]),
            "orc_evm_worst_macro_db": (f64, [vp, vp, sz]),
            "orc_evm_aggregate_db": (f64, [vp, vp, sz]),
            "orc_evm_aggregate_db_f64ref": (f64, [vp, vp, sz]),
            "orc_qpsk_modulate": (None, [vp, sz, vp]),
            "orc_qpsk_demod_naive": (None, [vp, sz, vp]),
            "orc_modulate": (i32, [vp, sz, i32, vp, vp]),
            "orc_demod_naive": (i32, [vp, sz, i32, vp, i32, vp]),
            "orc_awgn_apply": (None, [vp, sz, f32, C.c_uint64, C.c_uint64]),
            "orc_awgn_fill": (None, [vp, sz, f32, C.c_uint64, C.c_uint64]),
            "orc_philox4x32_10": (None, [vp, vp, vp]),
            "orc_philox4x32": (None, [vp, vp, i32, vp]),
            "orc_rng_normal_pairs": (None, [vp, C.c_size_t, vp]),
            "orc_time_shape": (f64, [i32, sz, sz, i32]),
            "orc_synth_cnormal": (None, [C.c_uint64, vp, sz]),
            "orc_synth_lowpass_taps": (None, [sz, f64, vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


class LengthMismatch(AssertionError):
    """Stands in for the reference's assert_eq! panics."""


# ---- element-wise (return a new array; the reference mutates in place) ------
def vec_scale(x, s):
    x = _c64(x).copy(); lib().orc_vec_scale(_p(x), x.size, float(np.float32(s))); return x


def _binary(name, a, b, msg="Vectors must have same length"):
    a = _c64(a).copy(); b = _c64(b)
    if getattr(lib(), name)(_p(a), a.size, _p(b), b.size) != 0:
        raise LengthMismatch(msg)
    return a


def vec_mul(a, b): return _binary("orc_vec_mul", a, b)
def vec_div(a, b): return _binary("orc_vec_div", a, b)
def vec_add(a, b): return _binary("orc_vec_add", a, b)
def vec_sub(a, b): return _binary("orc_vec_sub", a, b)
def vec_clone(a, b): return _binary("orc_vec_clone", a, b)


def vec_conj(x):
    x = _c64(x).copy(); lib().orc_vec_conj(_p(x), x.size); return x


def vec_mirror(x):
    x = _c64(x).copy(); lib().orc_vec_mirror(_p(x), x.size); return x


def vec_zero(x):
    x = _c64(x).copy(); lib().orc_vec_zero(_p(x), x.size); return x


def scale_factor(kind, n, x=0.0):
    return np.float32(lib().orc_scale_factor(kind, n, float(np.float32(x))))


def scale_apply(kind, data, x=0.0):
    d = _c64(data).copy(); lib().orc_scale_apply(kind, float(np.float32(x)), _p(d), d.size); return d


# ---- FFT --------------------------------------------------------------------
class Cfft:
    """Mirror of the reference's Cfft (src/fft.rs:134-235) over the oracle FFT."""

    def __init__(self, n):
        self._h = lib().orc_fft_plan_create(n)
        self.n = n

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_fft_plan_destroy(self._h); self._h = None

    def len(self):
        return lib().orc_fft_plan_len(self._h)

    def _oop(self, x, sign, kind, xs):
        x = _c64(x); out = np.empty(self.n, np.complex64)
        if lib().orc_cfft_outofplace(self._h, _p(x), x.size, _p(out), sign, kind, float(np.float32(xs))) != 0:
            raise LengthMismatch("Input and FFT must be the same length")
        return out

    def fwd(self, x, kind=SCALE_NONE, xs=0.0): return self._oop(x, SIGN_REF_FWD, kind, xs)
    def bwd(self, x, kind=SCALE_NONE, xs=0.0): return self._oop(x, SIGN_REF_BWD, kind, xs)

    def exec_sign(self, x, sign, kind=SCALE_NONE, xs=0.0):
        return self._oop(x, sign, kind, xs)

    def frames(self, x, sign, kind=SCALE_NONE, xs=0.0):
        x = _c64(x).reshape(-1, self.n)
        return np.stack([self._oop(f, sign, kind, xs) for f in x]).reshape(-1)

    def tmp(self, x, sign, kind=SCALE_NONE, xs=0.0):
        x = _c64(x)
        ptr = lib().orc_cfft_tmp(self._h, _p(x), x.size, sign, kind, float(np.float32(xs)))
        if not ptr:
            raise LengthMismatch("Input and FFT must be the same length")
        buf = (C.c_float * (2 * self.n)).from_address(ptr)      # n complex values = tmp[len..]
        return np.frombuffer(buf, dtype=np.complex64).copy()


def fft_f64(x, sign):
    x = np.ascontiguousarray(x, dtype=np.complex128); out = np.empty_like(x)
    lib().orc_fft_f64(_p(x), _p(out), x.size, sign); return out


def fft_f64_frames(x, n, sign):
    x = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1, n)
    return np.stack([fft_f64(f, sign) for f in x]).reshape(-1)


def dft_naive_f64(x, sign):
    x = np.ascontiguousarray(x, dtype=np.complex128); out = np.empty_like(x)
    lib().orc_dft_naive_f64(_p(x), _p(out), x.size, sign); return out


# ---- FIR --------------------------------------------------------------------
def fir_direct_f64(h, x, hist=None):
    h = _c64(h); x = _c64(x); y = np.empty(x.size, np.complex128)
    hp = None
    if hist is not None:
        hist = _c64(hist); assert hist.size == h.size - 1; hp = _p(hist)
    lib().orc_fir_direct_f64(_p(h), h.size, hp, _p(x), x.size, _p(y)); return y


def fir_ols_f32(h, x, fft_len=2048, hop=None, hist=None, threads=1):
    h = _c64(h); x = _c64(x); y = np.empty(x.size, np.complex64)
    hop = hop or (fft_len - h.size + 1)
    if threads > 1 and hist is None:
        rc = lib().orc_fir_ols_f32_mt(_p(h), h.size, fft_len, hop, _p(x), x.size, _p(y), threads)
    else:
        hp = None
        if hist is not None:
            hist = _c64(hist); assert hist.size == h.size - 1; hp = _p(hist)
        rc = lib().orc_fir_ols_f32(_p(h), h.size, fft_len, hop, hp, _p(x), x.size, _p(y))
    if rc != 0:
        raise ValueError("bad FIR geometry")
    return y


def correlate_frames(sig_freq, frames):
    sig = _c64(sig_freq); fr = _c64(frames).copy()
    lib().orc_correlate_frames(_p(sig), sig.size, _p(fr), fr.size // sig.size); return fr


# ---- sampling -----------------------------------------------------------------
def interpolate(src, n_between, compat_im=True):
    src = _c64(src)
    if src.size == 0:
        raise IndexError("interpolate on empty src (reference panics: sampling.rs:23)")
    dst = np.empty(src.size + (src.size - 1) * n_between, np.complex64)
    n = lib().orc_interpolate(_p(src), src.size, _p(dst), n_between, 1 if compat_im else 0)
    assert n == dst.size
    return dst


def downsample(src, n_dst, release=False, step_by=False):
    """sampling::downsample (debug build: sampling.rs:28-42); release=True: the same function with the debug_assert
    compiled out (what `cargo bench` runs, benches/benches.rs:113,130); step_by=True: downsample_sb (:49-62)."""
    src = np.ascontiguousarray(src); dst = np.empty(n_dst, src.dtype)
    if release:
        if lib().orc_downsample_release(_p(src), src.size, _p(dst), n_dst, src.dtype.itemsize, 1 if step_by else 0) != 0:
            raise LengthMismatch("the reference panics (division by zero / index out of bounds / step_by(0))")
        return dst
    if lib().orc_downsample(_p(src), src.size, _p(dst), n_dst, src.dtype.itemsize) != 0:
        raise LengthMismatch("Only even decimations are supported")
    return dst


# ---- tolerance ----------------------------------------------------------------
def assert_evm(act, ref, db=-80.0):
    """Literal assert_evm! (src/lib.rs:26-49) + NaN reject."""
    act = _c64(act); ref = _c64(ref)
    r = lib().orc_assert_evm(_p(act), act.size, _p(ref), ref.size, float(db))
    if r == -2:
        raise AssertionError("Input slices/vectors must be same length / The EVM threshold must be negative")
    if r >= 0:
        raise AssertionError(
            f"EVM limit exceeded ({db} dB) for element {r}. Actual {act[r]}, Expected {ref[r]}")


def evm_worst_macro_db(act, ref):
    act = _c64(act); ref = _c64(ref); return lib().orc_evm_worst_macro_db(_p(act), _p(ref), act.size)


def evm_db(act, ref):
    """Conventional aggregate EVM, 20*log10(rms err / rms ref)."""
    act = _c64(act)
    ref = np.ascontiguousarray(ref)
    if ref.dtype == np.complex128:
        return lib().orc_evm_aggregate_db_f64ref(_p(act), _p(ref), act.size)
    ref = _c64(ref)
    return lib().orc_evm_aggregate_db(_p(act), _p(ref), act.size)


# ---- modulation ---------------------------------------------------------------
def qpsk_modulate(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8); out = np.empty(bits.size // 2, np.complex64)
    lib().orc_qpsk_modulate(_p(bits), bits.size, _p(out)); return out


def qpsk_demod_naive(sym):
    sym = _c64(sym); out = np.empty(sym.size * 2, np.uint8)
    lib().orc_qpsk_demod_naive(_p(sym), sym.size, _p(out)); return out


def modulate(bits, bps, table=None):
    bits = np.ascontiguousarray(bits, dtype=np.uint8); out = np.empty(bits.size // bps, np.complex64)
    tp = _p(_c64(table)) if table is not None else None
    if lib().orc_modulate(_p(bits), bits.size, bps, tp, _p(out)) != 0:
        raise LengthMismatch("bit count is not a multiple of BITS_PER_SYMBOL")
    return out


def demod_naive(sym, bps, table=None, compat=True):
    sym = _c64(sym); out = np.empty(sym.size * bps, np.uint8)
    tp = _p(_c64(table)) if table is not None else None
    lib().orc_demod_naive(_p(sym), sym.size, bps, tp, 1 if compat else 0, _p(out)); return out


def awgn_fill(n, power, seed=815, offset=0):
    t = np.empty(n, np.complex64); lib().orc_awgn_fill(_p(t), n, float(np.float32(power)), int(seed), int(offset)); return t


def philox4x32_10(counter, key):
    c = np.ascontiguousarray(counter, np.uint32); k = np.ascontiguousarray(key, np.uint32); o = np.empty(4, np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p)); return o


def time_shape(op, n, b=0, reps=1000):
    """seconds per call of restated op `op` on a criterion shape (see orc_time_shape)"""
    return lib().orc_time_shape(int(op), int(n), int(b), int(reps))


def philox4x32(counter, key, rounds):
    c = np.ascontiguousarray(counter, np.uint32); k = np.ascontiguousarray(key, np.uint32); o = np.empty(4, np.uint32)
    lib().orc_philox4x32(c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p), int(rounds), o.ctypes.data_as(C.c_void_p)); return o


def rng_normal_pairs(a, b):
    """the generator's Box-Muller stage on explicit word pairs (a[i], b[i])"""
    ab = np.ascontiguousarray(np.stack([np.asarray(a, np.uint32), np.asarray(b, np.uint32)], axis=1))
    out = np.empty(ab.shape[0], np.complex64)
    lib().orc_rng_normal_pairs(ab.ctypes.data_as(C.c_void_p), ab.shape[0], _p(out)); return out


def awgn_apply(signal, power, seed=815, offset=0):
    s = _c64(signal).copy(); lib().orc_awgn_apply(_p(s), s.size, float(np.float32(power)), int(seed), int(offset)); return s


# ---- synthetic input ------------------------------------------------------------
def synth_cnormal(seed, n):
    out = np.empty(n, np.complex64); lib().orc_synth_cnormal(int(seed), _p(out), n); return out


def synth_lowpass_taps(ntaps=64, cutoff=0.25):
    out = np.empty(ntaps, np.complex64); lib().orc_synth_lowpass_taps(ntaps, float(cutoff), _p(out)); return out
