"""aether_primitives_amd -- MI355X (gfx950) backend for the cf32 hot path of
razorheadfx/aether_primitives (VecOps / Fft / FIR / sampling).

The product is the C-ABI library `lib/libaether_hip.so` (include/aether_hip.h),
hand-written HIP.  This package is the thin host-side mirror of the reference's
Rust interface used by the tests and the bench: same names, same argument
meaning, same error behaviour (length mismatches raise, as the reference
panics).  Nothing here computes on the CPU; without the built library the
import fails.
"""
from ._lib import AetherError, LengthMismatch, load as _load

_load()   # fail loudly at import time if the HIP library is not built

from .context import Context, DeviceVec, HostVec          # noqa: E402
from .fft import Scale, HipFft, SIGN_REF_FWD, SIGN_REF_BWD  # noqa: E402
from .fir import Fir                                      # noqa: E402
from . import sampling                                    # noqa: E402
from . import modulation                                  # noqa: E402
from . import noise                                       # noqa: E402
from . import file                                        # noqa: E402
from . import pool                                        # noqa: E402
from . import pipeline                                    # noqa: E402
from .evm import assert_evm, evm_db                       # noqa: E402

__all__ = ["AetherError", "LengthMismatch", "Context", "DeviceVec", "HostVec", "Scale", "HipFft",
           "SIGN_REF_FWD", "SIGN_REF_BWD", "Fir", "sampling", "modulation", "noise", "assert_evm", "evm_db"]
