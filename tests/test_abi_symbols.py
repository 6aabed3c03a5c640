"""CPU: the C-ABI library loads and exports every symbol include/aether_hip.h
declares; the ctypes table covers the header one to one.  No compute calls."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aether_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"AETH_API[^;(]*?\b(aeth_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_boundary():
    syms = header_symbols()
    assert len(syms) >= 55
    for must in ("aeth_vec_mul", "aeth_fft_create", "aeth_fft_exec", "aeth_fir_exec", "aeth_interpolate",
                 "aeth_downsample", "aeth_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from aether_primitives_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libaether_hip.so not built (run __graft_entry__.build())"
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r"\sT\s+(aeth_[a-z0-9_]+)", out))
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, f"declared in include/aether_hip.h but not exported: {missing}"
    # nothing but the C ABI leaks out (-fvisibility=hidden)
    leaked = [s for s in re.findall(r"\s[TDB]\s+(\S+)", out) if not s.startswith("aeth_") and not s.startswith("__hip")
              and not s.startswith("_fini") and not s.startswith("_init")]
    assert not [s for s in leaked if "oracle" in s.lower() or s.startswith("orc_")]


def test_ctypes_table_matches_header():
    from aether_primitives_amd import _lib
    lib = _lib.load()          # resolves every prototype or raises
    assert sorted(_lib.PROTOTYPES) == header_symbols()
    assert lib.aeth_version() >= 0x000100


def test_rust_binding_declares_the_whole_header():
    """rust/src/ffi.rs (the binding a maintainer adds; uncompiled here) stays one to one with the header."""
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (aeth_[a-z0-9_]+)\s*\(", ffi))
    assert sorted(declared) == header_symbols()


def test_product_never_links_the_oracle():
    """The product path must not import / link anything under oracle/."""
    from aether_primitives_amd import _lib
    ldd = subprocess.check_output(["ldd", _lib.LIB_PATH], text=True)
    assert "oracle" not in ldd
    pkg = os.path.join(ROOT, "aether_primitives_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in txt and "aeth_oracle" not in txt and "orc_" not in txt, f


def test_errors_do_not_need_a_gpu():
    """Argument validation happens before any device work."""
    import ctypes as C
    from aether_primitives_amd import _lib
    lib = _lib.load()
    assert lib.aeth_ctx_sync(None) == _lib.E_ARG
    assert b"null" in lib.aeth_last_error()
    assert abs(lib.aeth_scale_factor(1, 100, 0.0) - 0.1) < 1e-7       # SN: (100f32).sqrt().recip()
    assert lib.aeth_scale_factor(2, 4, 0.0) == 0.25                   # N
    assert lib.aeth_scale_factor(3, 4, 2.0) == 2.0                    # X
    assert lib.aeth_scale_factor(0, 4, 9.0) == 1.0                    # None
    with pytest.raises(_lib.AetherError):
        _lib.check(lib.aeth_fft_exec(None, None, 0, None, 0, 1, 0, 0.0))
    # the pinned pool and the explicit registration validate their arguments before touching the runtime
    p = C.c_void_p()
    assert lib.aeth_pool_create(None, 4096, 1, 0, C.byref(p)) == _lib.E_ARG and not p.value
    assert lib.aeth_pool_take(None, C.byref(p)) == _lib.E_ARG
    assert lib.aeth_pool_len(None) == 0 and lib.aeth_pool_cap(None) == 0 and lib.aeth_pool_destroy(None) == _lib.OK
    assert lib.aeth_host_register(None, None, 0) == _lib.E_ARG
    buf = (C.c_char * 64)()
    assert lib.aeth_host_is_pinned(buf, 64) == 0
    assert lib.aeth_downsample_release(None, None, 8, None, 4, 8, 0) == _lib.E_ARG
