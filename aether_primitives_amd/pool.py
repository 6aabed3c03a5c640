"""Pinned host buffers behind the reference's object pool (src/pool.rs:43-221).

`Pool` mirrors `pool::make` / `Pool<T>`: `take()` (None when empty), `take_or_make()`, `len()`, `cap()`; an element
is an `Elem` guard that goes back to the pool when dropped or closed (`Elem::drop`, pool.rs:196-208) and lends its
memory as a numpy array (`Deref`, :210-221).  Elements are page-locked ONCE by the library (hipHostMalloc): samples
produced into them cross PCIe with true asynchronous copies in `Fir.filter_stream`, with no staging copy and
without the library ever registering caller memory."""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import check

ZERO_ON_RETURN = 1


class Elem:
    """guard of one checked-out element (pool.rs:188-221): `.array(dtype)` is its memory, `close()` / `del` returns it.

    A numpy array lent by `.array()` (and every view sliced from it) keeps the element checked out: `close()` -- also
    the end of a `with` block -- hands the element back only once the last such array is gone, so no array ever
    aliases whoever takes the element next (Rust's borrow of `Deref::deref`, pool.rs:210-221, ends with the guard; a
    numpy view cannot be ended from outside, so the guard waits for it instead)."""

    def __init__(self, pool, ptr):
        self._pool, self._ptr = pool, ptr
        self._views = 0            # live buffers lent by array()
        self._closed = False

    @property
    def ptr(self):
        return self._ptr

    def array(self, dtype=np.complex64, count=None):
        """the element's memory as a numpy array; the element goes back to the pool only after the array has died"""
        assert self._ptr and not self._closed, "element already returned to its pool"
        dt = np.dtype(dtype)
        n = self._pool.elem_bytes // dt.itemsize if count is None else count
        assert n * dt.itemsize <= self._pool.elem_bytes
        buf = (C.c_char * (n * dt.itemsize)).from_address(self._ptr)
        self._views += 1
        weakref.finalize(buf, Elem._view_died, self)          # the finalizer holds the guard (and so the pool) alive
        return np.frombuffer(buf, dtype=dt, count=n)

    @staticmethod
    def _view_died(elem):
        elem._views -= 1
        if elem._closed and elem._views == 0:
            elem._give_back()

    def _give_back(self):
        ptr, self._ptr = self._ptr, None
        if ptr and self._pool._h:
            check(self._pool._lib.aeth_pool_give_back(self._pool._h, C.c_void_p(ptr)))

    def close(self):
        self._closed = True
        if self._views == 0:
            self._give_back()

    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pool:
    """pool::make(initial_len, maker, resetter) with maker = one pinned buffer of `elem_bytes`, resetter = nothing
    or (zero_on_return) memset 0"""

    def __init__(self, ctx, elem_bytes, initial_len=0, zero_on_return=False):
        self._lib = _lib.load()
        self.ctx = ctx
        h = C.c_void_p()
        check(self._lib.aeth_pool_create(ctx.h, elem_bytes, initial_len, ZERO_ON_RETURN if zero_on_return else 0, C.byref(h)))
        self._h = h

    @property
    def elem_bytes(self):
        return self._lib.aeth_pool_elem_bytes(self._h)

    def take(self):
        """Pool::take (:78-97): an element, or None when the pool is empty"""
        p = C.c_void_p()
        check(self._lib.aeth_pool_take(self._h, C.byref(p)))
        return Elem(self, p.value) if p.value else None

    def take_or_make(self):
        """Pool::take_or_make (:115-132): grows the pool by one element when it is empty"""
        p = C.c_void_p()
        check(self._lib.aeth_pool_take_or_make(self._h, C.byref(p)))
        return Elem(self, p.value)

    def len(self): return self._lib.aeth_pool_len(self._h)
    def is_empty(self): return self.len() == 0
    def cap(self): return self._lib.aeth_pool_cap(self._h)
    __len__ = len

    def close(self):
        """refused (ArgError) while elements are checked out -- also while numpy arrays lent by `Elem.array()` are alive.
        The C pool keeps its device by value and may outlive its context, so it is destroyed whatever the context's
        state (its pinned elements and their entries in the pinned-range registry would leak otherwise)."""
        if self._h:
            check(self._lib.aeth_pool_destroy(self._h))
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def is_pinned(arr):
    """does this array live wholly inside one pool element or one registered range?"""
    return bool(_lib.load().aeth_host_is_pinned(arr.ctypes.data_as(C.c_void_p), arr.nbytes))


def register(ctx, arr):
    """explicit opt-in: page-lock memory the caller owns (whole pages only; see include/aether_hip.h)"""
    check(_lib.load().aeth_host_register(ctx.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))


def unregister(ctx, arr):
    check(_lib.load().aeth_host_unregister(ctx.h, arr.ctypes.data_as(C.c_void_p)))
