"""GPU: the pinned buffer pool (reference src/pool.rs:43-221, its tests :223-297 replayed) and the host pipeline
over it: pool elements are copied from / to directly, anything else is staged by the library -- caller memory is
never registered behind the caller's back."""
import mmap

import numpy as np
import pytest

import aether_primitives_amd as ap
from aether_primitives_amd import pool
from helpers import bits_equal, rand_c64

pytestmark = pytest.mark.gpu


def test_reference_taking(ctx):
    """pool.rs:228-262"""
    p = pool.Pool(ctx, 50, initial_len=1)
    assert p.len() == 1 and p.cap() == 1
    c1 = p.take()
    assert c1 is not None, "First time checkout failed"
    assert p.len() == 0 and p.cap() == 1
    c1.close()
    assert p.len() == 1 and p.cap() == 1
    c1 = p.take()
    assert c1 is not None, "Second checkout failed, when it should have succeeded"
    c2 = p.take()
    assert c2 is None, "Third checkout succeeded when it should have failed"
    del c1
    assert p.len() == 1 and p.cap() == 1
    p.close()


def test_reference_resetting(ctx):
    """pool.rs:264-276: an element that was written comes back reset (resetter = zero on return)"""
    p = pool.Pool(ctx, 50, initial_len=1, zero_on_return=True)
    with p.take() as e:
        a = e.array(np.uint8)
        a[:] = np.arange(50, dtype=np.uint8)
        assert a.size == 50 and a.sum() > 0
        del a                                      # the borrow ends before the guard does (pool.rs:210-221)
    with p.take() as e:
        assert not e.array(np.uint8).any()
    p.close()


def test_a_lent_array_keeps_its_element_checked_out(ctx):
    """a numpy view of an element must never alias whoever takes the element next, nor outlive the pool's
    memory: the guard hands the element back only when the last array lent from it has died, and the pool refuses to
    close until then."""
    p = pool.Pool(ctx, 64, initial_len=1)
    e = p.take()
    a = e.array(np.uint8)
    tail = a[32:]                                  # a view of the view
    e.close()                                      # (or the end of a `with` block)
    assert p.len() == 0 and p.take() is None       # still out: `a` and `tail` are alive
    del a
    assert p.len() == 0
    with pytest.raises(ap.AetherError, match="still checked out"):
        p.close()
    tail[:] = 7                                    # still valid memory
    del tail
    assert p.len() == 1                            # the last array died: now it is back
    with pytest.raises(AssertionError):
        e.array(np.uint8)                          # a closed guard lends nothing
    p.close()


def test_reference_taking_or_making(ctx):
    """pool.rs:278-296"""
    p = pool.Pool(ctx, 50, initial_len=0)
    e1 = p.take_or_make()
    assert p.len() == 0 and p.cap() == 1
    e2 = p.take_or_make()
    assert p.len() == 0 and p.cap() == 2
    with pytest.raises(ap.AetherError, match="still checked out"):
        p.close()                                  # the reference keeps the pool alive while guards exist; here: refused
    del e1, e2
    assert p.len() == 2 and p.cap() == 2
    p.close()


def test_pool_misuse_is_reported(ctx):
    p, q = pool.Pool(ctx, 4096, 1), pool.Pool(ctx, 4096, 1)
    e = p.take()
    lib = p._lib
    assert lib.aeth_pool_give_back(q._h, e.ptr) != 0           # not an element of q
    assert b"not an element" in lib.aeth_last_error()
    ptr = e.ptr
    e.close()
    assert lib.aeth_pool_give_back(p._h, ptr) != 0             # given back twice
    with pytest.raises(ap.AetherError):
        pool.Pool(ctx, 0)
    p.close(); q.close()


@pytest.mark.parametrize("n", [1984 * 30 + 5, (1 << 20) + 77, 5 << 20])
def test_stream_from_pool_elements_is_direct_and_bit_identical(ctx, n):
    """samples produced INTO pool elements stream with no staging (stats: pinned == 3) and give the bits of the
    one-call flavour; plain numpy memory is staged through the library's own pool (pinned == 0), same bits"""
    taps = rand_c64(1, 64, scale=0.2)
    f = ap.Fir(ctx, taps, 2048)
    x = rand_c64(n, n)
    want = f.filter(x)
    p = pool.Pool(ctx, n * 8, initial_len=2)
    with p.take() as ein, p.take() as eout:
        xin, y = ein.array(np.complex64), eout.array(np.complex64)
        assert pool.is_pinned(xin) and pool.is_pinned(y) and not pool.is_pinned(x)
        xin[:] = x
        y[:] = 0
        _, st = f.filter_stream(xin, out=y, chunk=1984 * 64)
        assert st["pinned"] == 3 and bits_equal(y, want)
        # mixed: pinned input, pageable output and the other way round
        z = np.zeros(n, np.complex64)
        _, st = f.filter_stream(xin, out=z)
        assert st["pinned"] == 1 and bits_equal(z, want)
        y[:] = 0
        _, st = f.filter_stream(x, out=y)                      # a staged input takes the output through the host stage too
        assert st["pinned"] == 0 and bits_equal(y, want)
        del xin, y, _                                          # `_` is the array filter_stream returned
    z, st = f.filter_stream(x, report=True)
    assert st["pinned"] == 0 and bits_equal(z, want)
    assert st["active_copy_in"] > 0 and st["active_copy_out"] > 0 and len(st["lines"]) == 5
    p.close()


def test_pipeline_slots_keep_the_size_on_record():
    """The pipeline's device slots and staging elements live in the context between runs.  A run with ONE large chunk,
    then one with three small chunks (two more slots get allocated), then one with three chunks in between: every slot
    has to have the largest size asked for so far, not the size of the run that first needed it."""
    c = ap.Context(0)
    f = ap.Fir(c, rand_c64(1, 64, scale=0.2), 2048)
    hop = f.hop
    for n, chunk in ((hop * 40, hop * 40), (hop * 15, hop * 5), (hop * 60, hop * 20), (hop * 60 + 7, hop * 20), (hop * 200, hop * 50)):
        x = rand_c64(n, n)
        y, st = f.filter_stream(x, chunk=chunk)
        assert st["chunks"] == -(-n // chunk) and bits_equal(y, f.filter(x)), (n, chunk)
    c.close()


def test_stream_of_nothing_reports_nothing(ctx):
    f = ap.Fir(ctx, rand_c64(1, 64), 2048)
    y, st = f.filter_stream(np.zeros(0, np.complex64), report=True)      # no ZeroDivisionError: no rates for an empty run
    assert y.size == 0 and st["lines"] == [] and st["seconds"] == 0


def test_explicit_registration_rules(ctx):
    """aeth_host_register: whole pages only, no overlap with anything known; then the pipeline copies directly"""
    n = 1 << 18
    m = mmap.mmap(-1, 2 * n * 8)                               # page-aligned anonymous memory
    buf = np.frombuffer(m, np.complex64)
    xin, y = buf[:n], buf[n:]
    with pytest.raises(ap.AetherError, match="page"):
        pool.register(ctx, buf[1:n])                           # not page-aligned
    pool.register(ctx, xin)
    with pytest.raises(ap.AetherError, match="overlaps"):
        pool.register(ctx, buf)                                # touches the registered half
    pool.register(ctx, y)
    assert pool.is_pinned(xin) and pool.is_pinned(y[5:100])
    f = ap.Fir(ctx, rand_c64(2, 64, scale=0.2), 2048)
    x = rand_c64(9, n)
    xin[:] = x
    _, st = f.filter_stream(xin, out=y)
    assert st["pinned"] == 3 and bits_equal(y, f.filter(x))
    pool.unregister(ctx, xin); pool.unregister(ctx, y)
    assert not pool.is_pinned(xin)
    with pytest.raises(ap.AetherError, match="not registered"):
        pool.unregister(ctx, xin)
    # (the mapping goes away with its last numpy view)
