"""ctypes binding of libaether_hip.so (include/aether_hip.h).

The library is the product; this module only declares its prototypes.  There is
no CPU fallback: if the shared object is missing or does not load, importing
fails loudly with instructions (build with `python -c "import __graft_entry__ as
g; g.build()"` or `make -C aether_primitives_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaether_hip.so")
# tools/ only: AETH_LAB_LIB=1 loads the lab build of the SAME sources (`make -C aether_primitives_amd/csrc LAB=1`),
# which keeps the measured-and-not-kept routes reachable through their AETH_* knobs (csrc/aeth_internal.h, lab_int)
if os.environ.get("AETH_LAB_LIB") == "1":
    LIB_PATH = os.path.join(_HERE, "lib", "libaether_hip_lab.so")

OK = 0
E_LEN, E_ARG, E_ALIGN, E_HIP, E_NOMEM, E_UNSUPPORTED = -1, -2, -3, -4, -5, -6

vp, sz, f32, i32 = C.c_void_p, C.c_size_t, C.c_float, C.c_int
pvp = C.POINTER(C.c_void_p)
psz = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); mirrors include/aether_hip.h declaration by declaration
PROTOTYPES = {
    "aeth_last_error": (C.c_char_p, []),
    "aeth_version": (i32, []),
    "aeth_device_count": (i32, [C.POINTER(i32)]),
    "aeth_ctx_create": (i32, [i32, pvp]),
    "aeth_ctx_create_on_stream": (i32, [i32, vp, pvp]),
    "aeth_ctx_destroy": (i32, [vp]),
    "aeth_ctx_sync": (i32, [vp]),
    "aeth_ctx_set_overlap": (i32, [vp, i32]),
    "aeth_ctx_overlap": (i32, [vp]),
    "aeth_ctx_stream": (vp, [vp]),
    "aeth_ctx_device": (i32, [vp]),
    "aeth_dev_alloc": (i32, [vp, sz, pvp]),
    "aeth_dev_free": (i32, [vp, vp]),
    "aeth_upload": (i32, [vp, vp, vp, sz]),
    "aeth_download": (i32, [vp, vp, vp, sz]),
    "aeth_copy_dev": (i32, [vp, vp, vp, sz]),
    "aeth_event_create": (i32, [vp, pvp]),
    "aeth_event_destroy": (i32, [vp]),
    "aeth_event_record": (i32, [vp]),
    "aeth_event_sync": (i32, [vp]),
    "aeth_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "aeth_vec_scale": (i32, [vp, vp, sz, f32]),
    "aeth_vec_mul": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_div": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_conj": (i32, [vp, vp, sz]),
    "aeth_vec_add": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_sub": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_mirror": (i32, [vp, vp, sz]),
    "aeth_vec_clone": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_zero": (i32, [vp, vp, sz]),
    "aeth_vec_mirror_frames": (i32, [vp, vp, sz, sz]),
    "aeth_vec_mul_frames": (i32, [vp, vp, sz, sz, vp, sz]),
    "aeth_host_vec_scale": (i32, [vp, vp, sz, f32]),
    "aeth_host_vec_mul": (i32, [vp, vp, sz, vp, sz]),
    "aeth_host_vec_div": (i32, [vp, vp, sz, vp, sz]),
    "aeth_host_vec_conj": (i32, [vp, vp, sz]),
    "aeth_host_vec_add": (i32, [vp, vp, sz, vp, sz]),
    "aeth_host_vec_sub": (i32, [vp, vp, sz, vp, sz]),
    "aeth_host_vec_mirror": (i32, [vp, vp, sz]),
    "aeth_host_vec_clone": (i32, [vp, vp, sz, vp, sz]),
    "aeth_host_vec_zero": (i32, [vp, vp, sz]),
    "aeth_scale_factor": (f32, [i32, sz, f32]),
    "aeth_scale_apply": (i32, [vp, i32, f32, vp, sz]),
    "aeth_fft_create": (i32, [vp, sz, sz, pvp]),
    "aeth_fft_destroy": (i32, [vp]),
    "aeth_fft_len": (sz, [vp]),
    "aeth_fft_algorithm": (C.c_char_p, [vp]),
    "aeth_fft_exec": (i32, [vp, vp, sz, vp, sz, i32, i32, f32]),
    "aeth_fft_exec_mirrored": (i32, [vp, vp, sz, vp, sz, i32, i32, f32]),
    "aeth_fft_exec_interpolate": (i32, [vp, vp, sz, sz, i32, i32, f32, vp, sz, sz, i32, psz]),
    "aeth_fft_exec_host": (i32, [vp, vp, sz, vp, sz, i32, i32, f32]),
    "aeth_fft_exec_tmp_host": (i32, [vp, vp, sz, i32, i32, f32, pvp]),
    "aeth_fft_exec_tmp": (i32, [vp, vp, sz, sz, i32, i32, f32, pvp]),
    "aeth_fft_mul_ifft": (i32, [vp, vp, sz, sz, vp, sz, i32, f32, i32, f32]),
    "aeth_fft_mul_ifft_demod": (i32, [vp, vp, sz, sz, vp, sz, i32, f32, i32, f32, i32, vp, vp, sz, i32]),
    "aeth_fir_create": (i32, [vp, vp, sz, sz, pvp]),
    "aeth_fir_destroy": (i32, [vp]),
    "aeth_fir_ntaps": (sz, [vp]),
    "aeth_fir_fft_len": (sz, [vp]),
    "aeth_fir_hop": (sz, [vp]),
    "aeth_fir_exec": (i32, [vp, vp, vp, sz, vp]),
    "aeth_fir_exec_decim": (i32, [vp, vp, vp, sz, vp, sz]),
    "aeth_fir_exec_host": (i32, [vp, vp, vp, sz, vp]),
    "aeth_pool_create": (i32, [vp, sz, sz, i32, pvp]),
    "aeth_pool_destroy": (i32, [vp]),
    "aeth_pool_take": (i32, [vp, pvp]),
    "aeth_pool_take_or_make": (i32, [vp, pvp]),
    "aeth_pool_give_back": (i32, [vp, vp]),
    "aeth_pool_len": (sz, [vp]),
    "aeth_pool_cap": (sz, [vp]),
    "aeth_pool_elem_bytes": (sz, [vp]),
    "aeth_host_register": (i32, [vp, vp, sz]),
    "aeth_host_unregister": (i32, [vp, vp]),
    "aeth_host_is_pinned": (i32, [vp, sz]),
    "aeth_vec_chain": (i32, [vp, vp, sz, vp, sz]),
    "aeth_vec_fft": (i32, [vp, vp, sz, i32, i32, f32]),
    "aeth_host_vec_fft": (i32, [vp, vp, sz, i32, i32, f32]),
    "aeth_stream_out_count": (sz, [vp, vp, sz]),
    "aeth_stream_host": (i32, [vp, vp, vp, sz, vp, sz, sz, vp]),
    "aeth_stream_chain_out_count": (sz, [vp, vp, sz, sz]),
    "aeth_stream_host_chain": (i32, [vp, vp, sz, vp, sz, vp, sz, sz, vp, vp]),
    "aeth_stream_host_util": (i32, [vp, vp, vp, sz, vp, sz, sz, vp]),
    "aeth_ctx_trim": (i32, [vp]),
    "aeth_test_fail_staging_after": (None, [i32]),
    "aeth_fir_stream_host": (i32, [vp, vp, sz, vp, sz, vp]),
    "aeth_fir_stream_host_util": (i32, [vp, vp, sz, vp, sz, vp]),
    "aeth_fir_stream_file": (i32, [vp, C.c_char_p, C.c_char_p, sz, vp]),
    "aeth_stream_file": (i32, [vp, vp, C.c_char_p, C.c_char_p, sz, vp]),
    "aeth_file_count_structs": (i32, [C.c_char_p, sz, psz]),
    "aeth_file_read": (i32, [C.c_char_p, sz, vp, sz, sz]),
    "aeth_file_write": (i32, [C.c_char_p, vp, sz, sz, i32]),
    "aeth_interpolate": (i32, [vp, vp, sz, vp, sz, sz, i32, psz]),
    "aeth_interpolate_frames": (i32, [vp, vp, sz, sz, vp, sz, sz, i32, psz]),
    "aeth_host_interpolate": (i32, [vp, vp, sz, vp, sz, sz, i32, psz]),
    "aeth_downsample": (i32, [vp, vp, sz, vp, sz, sz]),
    "aeth_host_downsample": (i32, [vp, vp, sz, vp, sz, sz]),
    "aeth_downsample_release": (i32, [vp, vp, sz, vp, sz, sz, i32]),
    "aeth_host_downsample_release": (i32, [vp, vp, sz, vp, sz, sz, i32]),
    "aeth_modulate": (i32, [vp, vp, sz, i32, vp, vp, sz]),
    "aeth_modulate_awgn": (i32, [vp, vp, sz, i32, vp, vp, sz, f32, C.c_uint64, C.c_uint64]),
    "aeth_demod_naive": (i32, [vp, vp, sz, i32, vp, vp, sz, i32]),
    "aeth_awgn_apply": (i32, [vp, vp, sz, f32, C.c_uint64, C.c_uint64]),
    "aeth_awgn_fill": (i32, [vp, vp, sz, f32, C.c_uint64, C.c_uint64]),
    "aeth_rng_philox4x32_10": (i32, [vp, vp, sz, vp]),
    "aeth_rng_philox4x32": (i32, [vp, vp, sz, i32, vp]),
    "aeth_rng_normal_pairs": (i32, [vp, vp, sz, vp]),
}

_lib = None


class AetherError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code
        self.message = message


class LengthMismatch(AetherError, AssertionError):
    """AETH_E_LEN: the reference panics (assert_eq!) with this message."""


def load():
    """Load libaether_hip.so; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP backend is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C aether_primitives_amd/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().aeth_last_error().decode("utf-8", "replace")
    if rc == E_LEN:
        raise LengthMismatch(rc, msg)
    raise AetherError(rc, msg)
