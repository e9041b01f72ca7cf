"""CPU side of the round-3 additions to the C ABI: the options API (what used to be environment variables read per
call), the sharding entry points that need no device, argument checking.  No compute: the library loads without a GPU."""
import ctypes as C

import pytest

from sdfkit_amd import _native as N
from sdfkit_amd import dist as D


def test_options_round_trip_and_ranges():
    L = N.lib()
    for key, good, bad in ((N.OPT_LANES, (0, 2, 3, 4), (-1, 5)), (N.OPT_TOKENS, (-1, 0, 1, 3), (-2, 4)), (N.OPT_GRAPHS, (0, 1, 2), (3,)),
                           (N.OPT_COPY_MODE, (0, 1, 2), (3, -1)), (N.OPT_CORNER_EVAL, (0, 1), (2,)), (N.OPT_VCOLOR_EVAL, (0, 1), (2,)),
                           (N.OPT_DIST_EXCHANGE, (0, 1, 2, 3), (4, -1)), (N.OPT_DIST_LANES, (0, 2, 3), (4, -1)), (N.OPT_CODE_CACHE, (0, 1), (2,)),
                           (N.OPT_PREFAULT_HUGE, (0, 1), (2,)), (N.OPT_DIST_INDEX16, (0, 1), (2,)), (N.OPT_STREAM_PLACEMENT, (0, 1), (2, -1)), (N.OPT_IDLE_LANE, (0, 1), (2,)), (N.OPT_IDLE_PROGRAMS, (0, 32), (-1, 1025)), (N.OPT_ELIDE_VOLUME, (1, 2, 0), (3, -1)),
                           (N.OPT_COLOR_PASSES, (1, 2, 0), (3, -1))):
        before = N.get_option(key)
        try:
            for v in good:
                N.set_option(key, v)
                assert N.get_option(key) == v
            for v in bad:
                assert L.sdfk_set_option(key, v) == 1                    # SDFK_ERR_INVALID
                assert b"out of range" in L.sdfk_last_error()
                assert N.get_option(key) == good[-1]                     # unchanged
        finally:
            N.set_option(key, before)
    assert L.sdfk_set_option(N.OPT_HW_QUEUES, 8) == 1 and b"read-only" in L.sdfk_last_error()
    assert N.get_option(N.OPT_HW_QUEUES) in (0, 8) or N.get_option(N.OPT_HW_QUEUES) > 0
    assert L.sdfk_set_option(99, 0) == 1 and L.sdfk_get_option(99, C.byref(C.c_int64())) == 1
    assert L.sdfk_get_option(N.OPT_LANES, None) == 1


def test_option_context_manager_restores():
    before = N.get_option(N.OPT_GRAPHS)
    with N.option(N.OPT_GRAPHS, 0):
        assert N.get_option(N.OPT_GRAPHS) == 0
    assert N.get_option(N.OPT_GRAPHS) == before


def test_sharding_entry_points_without_a_device():
    L = N.lib()
    assert D.info() == (1, 0, 0)                                           # no context: world 1, rank 0, backend none
    assert D.slab(512, 8, 0) == (0, 64, 0, 68) and D.slab(512, 8, 7) == (448, 511, 444, 68)
    assert D.slab(1024, 8, 3)[:2] == (384, 512)
    assert L.sdfk_dist_slab(0, 1, 0, None, None, None, None) == 1
    assert L.sdfk_dist_slab(8, 2, 2, None, None, None, None) == 1
    h = C.c_void_p()
    f3 = N.f3([0, 0, 0])
    # no device, no context: every compute entry fails loudly, none falls back to anything
    assert L.sdfk_dist_session_create(C.c_void_p(1), f3, f3, 8, 8, 8, 1, C.c_float(0), 2, C.byref(h)) in (1, 2) and not h.value
    assert L.sdfk_dist_submit(None) == 1 and L.sdfk_dist_collect(None, None, None) == 1
    assert L.sdfk_dist_mesh(None, C.byref(h)) == 1 and L.sdfk_dist_counts(None, None) == 1
    assert L.sdfk_dist_init(2, 0, None) == 1
    assert L.sdfk_dist_init_host(2, 0, None, None) == 1
    assert L.sdfk_mesh_size_hint(None, None, None, None) == 1
    assert L.sdfk_host_prefault(None, 0) == 0 and L.sdfk_host_prefault(None, -1) == 1
    L.sdfk_dist_shutdown()                                                 # nothing to shut down: no-op


def test_host_prefault_makes_pages_resident():
    """sdfk_host_prefault needs no device: the pages of a fresh mapping are resident afterwards (mincore)."""
    import mmap
    L = N.lib()
    n = 3 << 20
    mm = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    buf = (C.c_char * n).from_buffer(mm)
    addr = C.addressof(buf)
    libc = C.CDLL(None, use_errno=True)
    vec = (C.c_ubyte * (n // 4096))()
    assert libc.mincore(C.c_void_p(addr), C.c_size_t(n), vec) == 0 and not any(v & 1 for v in vec)
    assert L.sdfk_host_prefault(C.c_void_p(addr + 5), n - 9) == 0            # unaligned range
    assert libc.mincore(C.c_void_p(addr), C.c_size_t(n), vec) == 0 and all(v & 1 for v in vec)
    assert mm[:16] == b"\0" * 16
    del buf
    mm.close()


def test_cache_dir_setter_accepts_null():
    L = N.lib()
    assert L.sdfk_set_cache_dir(b"/nonexistent/dir/for/sdfkit") == 0 and L.sdfk_set_cache_dir(None) == 0
