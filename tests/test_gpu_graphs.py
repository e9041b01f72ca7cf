"""Captured launch graphs behind sdfk_sample_march (csrc/lib_march.hip, "captured launch graphs"): the repeat job of a
(program, bounds, grid, clip, iso) is one hipGraphLaunch into buffers the library keeps.  Results must be the ordinary
path's, bit for bit, whatever the callers do with the handles: hold many, drop them unread, change scene on the same
grid, outgrow the captured capacities.  All through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import Mesh
from sdfkit_amd import _native as N
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu

MN, MX = [-2.8125] * 3, [2.8125] * 3


def oracle_mesh(scene, mn, mx, dims, clip, iso=0.0):
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    return O.march(ov, oc, mn, mx, iso)


def raw(sdf, mn, mx, dims, clip, iso=0.0):
    m = C.c_void_p()
    N.check(N.lib().sdfk_sample_march(sdf.program(), N.f3(mn), N.f3(mx), *dims, 1 if clip else 0, C.c_float(iso), 1, C.byref(m)))
    return m


def graphs_expected():
    """(the suite is also run with SDFK_GRAPHS=0 / SDFK_LANES=0 in the environment -- the start-up defaults of the options:
    results must hold, graphs need not appear)"""
    return N.get_option(N.OPT_GRAPHS) != 0 and N.get_option(N.OPT_LANES) not in (0, 1)


def stats():
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    N.check(N.lib().sdfk_graph_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def test_repeat_calls_replay_a_graph_and_match_the_oracle(gpu):
    scene, sdf = S.CATALOGUE["readme_repeat_xy"]()
    dims = (44, 40, 48)
    om = oracle_mesh(scene, MN, MX, dims, True)
    _, launches0, _ = stats()
    for _ in range(16):     # 1: exact path (sets the hints); one sighting per lane; one build per lane; then replays
        assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, True)), om)
    jobs, launches, nbytes = stats()
    if graphs_expected():
        assert jobs >= 1 and launches - launches0 >= 6 and nbytes > 0


def test_graphs_off_on_same_bits(gpu):
    scene, sdf = S.CATALOGUE["union8"]()
    dims = (40, 44, 36)
    res = {}
    lanes_on = N.get_option(N.OPT_LANES) not in (0, 1)
    before = N.get_option(N.OPT_GRAPHS)
    try:
        for k, mode in enumerate((0, 1, 0, 2)):
            N.set_option(N.OPT_GRAPHS, mode)
            _, l0, _ = stats()
            # (a job is captured at the second sighting of its key on a lane and replayed from the third: three rounds of the lanes)
            res[k] = [Mesh._from_handle(raw(sdf, MN, MX, dims, False)) for _ in range(3 * max(N.get_option(N.OPT_LANES), 1) + 2)]
            _, l1, _ = stats()
            assert (l1 > l0) == (mode != 0 and lanes_on)
    finally:
        N.set_option(N.OPT_GRAPHS, before)
    om = oracle_mesh(scene, MN, MX, dims, False)
    for ms in res.values():
        for m in ms:
            assert_mesh_equal(m, om)


def test_many_live_handles_of_one_key(gpu):
    """More live handles than captured jobs per key and lane: the rest take the ordinary path; read in reverse."""
    scene, sdf = S.CATALOGUE["sdf_with_color"]()
    dims = (40, 36, 44)
    om = oracle_mesh(scene, MN, MX, dims, True)
    assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, True)), om)
    hs = [raw(sdf, MN, MX, dims, True) for _ in range(30)]
    for h in reversed(hs):
        assert_mesh_equal(Mesh._from_handle(h), om)
    hs = [raw(sdf, MN, MX, dims, True) for _ in range(30)]     # again: every captured job is free again
    for h in hs:
        assert_mesh_equal(Mesh._from_handle(h), om)


def test_unread_handles_dropped_between_replays(gpu):
    L = N.lib()
    scene, sdf = S.CATALOGUE["union8"]()
    dims = (36, 40, 32)
    om = oracle_mesh(scene, MN, MX, dims, True)
    assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, True)), om)
    for i in range(200):
        h = raw(sdf, MN, MX, dims, True)
        if i % 7 == 3:
            assert_mesh_equal(Mesh._from_handle(h), om)
        else:
            L.sdfk_mesh_free(h)
    assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, True)), om)


def test_result_outgrows_the_captured_capacities(gpu):
    """Hints come from the grid shape: a small sphere first, then a scene with a far larger mesh on the same grid.  The
    captured job of the large scene is first built too small (redone exactly), then rebuilt with the new hints."""
    dims = (52, 48, 44)
    small_scene, small = S.sphere_w(0.3)
    big_scene, big = S.CATALOGUE["readme_repeat_xy"]()
    om_small = oracle_mesh(small_scene, MN, MX, dims, True)
    om_big = oracle_mesh(big_scene, MN, MX, dims, True)
    for _ in range(5):
        assert_mesh_equal(Mesh._from_handle(raw(small, MN, MX, dims, True)), om_small)
    assert om_big.vertices.shape[0] > 2 * om_small.vertices.shape[0] + 8192
    for _ in range(10):
        assert_mesh_equal(Mesh._from_handle(raw(big, MN, MX, dims, True)), om_big)
    for _ in range(5):      # and back: capacities larger than needed are fine
        assert_mesh_equal(Mesh._from_handle(raw(small, MN, MX, dims, True)), om_small)
        assert_mesh_equal(Mesh._from_handle(raw(big, MN, MX, dims, True)), om_big)


def test_keys_differ_by_iso_clip_and_bounds(gpu):
    scene, sdf = S.CATALOGUE["sphere_w"]()
    dims = (40, 40, 40)
    mn2, mx2 = [-2.0, -2.5, -3.0], [2.5, 2.0, 3.0]
    cases = [(MN, MX, True, 0.0), (MN, MX, False, 0.0), (MN, MX, True, 0.25), (mn2, mx2, True, 0.0), (mn2, mx2, False, -0.125)]
    oms = [oracle_mesh(scene, mn, mx, dims, clip, iso) for mn, mx, clip, iso in cases]
    for _ in range(6):
        hs = [raw(sdf, mn, mx, dims, clip, iso) for mn, mx, clip, iso in cases]
        for h, om in zip(hs, oms):
            assert_mesh_equal(Mesh._from_handle(h), om)


def test_many_programs_evict_least_recently_used(gpu):
    """More keys than the library keeps captured jobs for: old ones are destroyed, results stay right."""
    dims = (28, 28, 28)
    for rnd in range(2):
        for k in range(14):
            scene, sdf = S.sphere_w(0.5 + 0.125 * k)
            om = oracle_mesh(scene, MN, MX, dims, False)
            for _ in range(5):
                assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, False)), om)
    jobs, _, _ = stats()
    assert jobs <= 24


def test_inside_a_lane_section(gpu):
    L = N.lib()
    scene, sdf = S.CATALOGUE["union8"]()
    dims = (36, 32, 40)
    om = oracle_mesh(scene, MN, MX, dims, True)
    assert_mesh_equal(Mesh._from_handle(raw(sdf, MN, MX, dims, True)), om)
    for _ in range(4):
        N.check(L.sdfk_lane_begin(2, None))
        try:
            h = raw(sdf, MN, MX, dims, True)
        finally:
            N.check(L.sdfk_lane_end(0))
        assert_mesh_equal(Mesh._from_handle(h), om)
