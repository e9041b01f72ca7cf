"""Deferred completion of meshes (speculative path), the internal lanes and the re-evaluated
cell corners: results must not depend on WHEN a mesh is read, on what happens to its source
volume in between, or on which corner path (gather from the volume / re-evaluation of the
program) produced the records.  All through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import MarchingCubes, Mesh, Sdfs, Voxels
from sdfkit_amd import _native as N
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu

MN, MX = [-2.8125] * 3, [2.8125] * 3


def oracle_mesh(scene, mn, mx, dims, clip):
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    return O.march(ov, oc, mn, mx)


def raw_sample_march(sdf, mn, mx, dims, clip):
    """sdfk_sample_march without touching the result: returns the (possibly pending) handle."""
    L = N.lib()
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(sdf.program(), N.f3(mn), N.f3(mx), *dims, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m)))
    return m


@pytest.mark.parametrize("name", ["readme_repeat_xy", "union8", "sphere_w"])
def test_corner_reevaluation_equals_gather(gpu, name):
    scene, sdf = S.CATALOGUE[name]()
    dims = (44, 40, 48)
    om = oracle_mesh(scene, MN, MX, dims, True)
    a = sdf.ToMesh(MN, MX, *dims)                       # corners re-evaluated by the program
    with N.option(N.OPT_CORNER_EVAL, 0):
        b = sdf.ToMesh(MN, MX, *dims)                   # corners gathered from the stored volume
    assert_mesh_equal(a, om)
    assert_mesh_equal(b, om)
    for f in ("Vertices", "Colors", "Normals", "Triangles"):
        assert np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True)


def test_many_queued_meshes_read_late_and_out_of_order(gpu):
    """More jobs than pending slots, alternating shapes and scenes, read back in reverse."""
    jobs = []
    for rep in range(3):      # the first round sets the size hints, later rounds are speculative
        for name, dims in (("readme_repeat_xy", (40, 36, 44)), ("union8", (36, 40, 32)), ("sdf_with_color", (40, 36, 44))):
            scene, sdf = S.CATALOGUE[name]()
            jobs.append((name, dims, scene, raw_sample_march(sdf, MN, MX, dims, True), sdf))
    for name, dims, scene, h, _ in reversed(jobs):
        m = Mesh._from_handle(h)
        assert_mesh_equal(m, oracle_mesh(scene, MN, MX, dims, True))


def test_phase_tokens_do_not_change_results(gpu):
    """SDFK_OPT_TOKENS (csrc/lib_context.hip, phase_token_wait/pass): the sampling kernels / k_vertices of consecutive jobs on the
    lanes wait for each other's events.  Forced on for small grids here (the default applies them from 2^27 voxels up),
    graphs off so that every job takes the ordinary path: many jobs in flight, two scenes, read late."""
    with N.option(N.OPT_GRAPHS, 0):
        for mask in (1, 3, 2):
            with N.option(N.OPT_TOKENS, mask):
                jobs = []
                for rep in range(4):
                    for name, dims in (("readme_repeat_xy", (40, 36, 44)), ("union8", (36, 40, 32))):
                        scene, sdf = S.CATALOGUE[name]()
                        jobs.append((scene, dims, raw_sample_march(sdf, MN, MX, dims, True), sdf))
                        if len(jobs) > 5:
                            scene0, dims0, h0, _ = jobs.pop(0)
                            assert_mesh_equal(Mesh._from_handle(h0), oracle_mesh(scene0, MN, MX, dims0, True))
                for scene0, dims0, h0, _ in jobs:
                    assert_mesh_equal(Mesh._from_handle(h0), oracle_mesh(scene0, MN, MX, dims0, True))


def test_unread_meshes_can_be_freed(gpu):
    """Nobody ever reads these: freeing a pending mesh must neither wait nor corrupt later work
    (more jobs than the ring of result slots)."""
    L = N.lib()
    scene, sdf = S.CATALOGUE["union8"]()
    dims = (36, 40, 32)
    om = oracle_mesh(scene, MN, MX, dims, True)
    assert_mesh_equal(sdf.ToMesh(MN, MX, *dims), om)    # hint
    for _ in range(150):
        L.sdfk_mesh_free(raw_sample_march(sdf, MN, MX, dims, True))
    assert_mesh_equal(sdf.ToMesh(MN, MX, *dims), om)


def test_source_volume_changes_after_march(gpu):
    """CreateMesh(volume) may return before the GPU is done; the volume is then overwritten /
    freed: the mesh must still be the mesh of the OLD contents."""
    scene, sdf = S.CATALOGUE["readme_repeat_xy"]()
    dims = (40, 36, 44)
    om = oracle_mesh(scene, MN, MX, dims, True)
    L = N.lib()
    for mutate in ("resample", "clip", "upload", "free"):
        vol = sdf.ToVoxels(MN, MX, *dims, clipToBounds=True)
        MarchingCubes.CreateMesh(vol)                   # hint for this shape
        h = C.c_void_p()
        N.check(L.sdfk_march(vol._h, C.c_float(0.0), 1, C.byref(h)))   # pending handle
        if mutate == "resample":
            vol._sample(Sdfs.Sphere(0.3))
        elif mutate == "clip":
            vol._sample(Sdfs.Sphere(3.0))
            vol.ClipToBounds()
        elif mutate == "upload":
            vol[3, 4, 5] = 7.0
            vol._sync_to_device()
        else:
            vol._free()
        assert_mesh_equal(Mesh._from_handle(h), om)


def test_speculative_guess_too_small_is_redone_on_first_read(gpu):
    n = 96
    small = Sdfs.Sphere(0.2)
    scene, big = S.CATALOGUE["union8"]()
    assert len(small.ToMesh([-1.5] * 3, [1.5] * 3, n, n, n, clipToBounds=False).Vertices) < 2000   # hint: tiny mesh
    handles = [raw_sample_march(big, MN, MX, (n, n, n), True) for _ in range(3)]      # all under-sized
    om = oracle_mesh(scene, MN, MX, (n, n, n), True)
    assert len(om.vertices) > 12000   # far beyond hint * 1.25 + 4096
    for h in handles:
        assert_mesh_equal(Mesh._from_handle(h), om)


@pytest.mark.parametrize("world", [2, 4])
def test_queued_slabs_are_packed_by_the_device(gpu, world):
    """GpuSlabWorker.enqueue: sample + mesh + pack of a slab without any host wait; payloads side
    by side as an all-gather leaves them, one rebase launch -> the single-volume mesh.  Then the
    same with size hints that are far too small: the headers must say so (-1), nothing else."""
    import torch
    from sdfkit_amd import dist as D
    from tests import slab_worker as W
    scene, sdf = S.CATALOGUE["readme_repeat_xy"]()
    dims = (44, 40, 48)
    whole = sdf.ToMesh(MN, MX, *dims)
    L = N.lib()
    N.bind_torch_stream()
    try:
        workers = [W.GpuSlabWorker(sdf, MN, MX, *dims, r, world, True, 0.0) for r in range(world)]
        counts = [w.run_local() for w in workers]                       # exact path: sets the hints
        stride = max(D.SLAB_HEADER_BYTES + 36 * a + 4 * b for a, b in counts) + 512
        for _ in range(3):
            g = torch.zeros((world, stride), dtype=torch.uint8, device="cuda")
            for r, w in enumerate(workers):
                w.enqueue(g[r])
            N.check(L.sdfk_slabs_rebase(C.c_void_p(g.data_ptr()), world, stride))
            torch.cuda.synchronize()
            h = g[:, :16].cpu().numpy().view(np.int64)
            assert [tuple(x) for x in h] == counts
            V, Cc, Nn, T, bmin, bmax = D.unpack_self_describing(g.cpu().numpy())
            assert np.array_equal(T, whole.Triangles) and np.array_equal(V, whole.Vertices)
            assert np.array_equal(Cc, whole.Colors) and np.array_equal(Nn, whole.Normals, equal_nan=True)
            assert np.array_equal(bmin, whole.Min) and np.array_equal(bmax, whole.Max)
        for w in workers:
            w.close()
        # hints from a tiny mesh of the same slab shapes, then the big scene again: under-sized guesses
        tiny = Sdfs.Sphere(0.05)
        small = [W.GpuSlabWorker(tiny, MN, MX, *dims, r, world, True, 0.0) for r in range(world)]
        for w in small:
            w.run_local()
            w.close()
        big_dims_workers = [W.GpuSlabWorker(S.CATALOGUE["union8"]()[1], MN, MX, 4 * dims[0], 4 * dims[1], dims[2], r, world, True, 0.0)
                            for r in range(world)]
        tiny_same_shape = [W.GpuSlabWorker(tiny, MN, MX, 4 * dims[0], 4 * dims[1], dims[2], r, world, True, 0.0) for r in range(world)]
        for w in tiny_same_shape:
            w.run_local()
            w.close()
        g = torch.zeros((world, 1 << 20), dtype=torch.uint8, device="cuda")
        for r, w in enumerate(big_dims_workers):
            w.enqueue(g[r])
        torch.cuda.synchronize()
        h = g[:, :16].cpu().numpy().view(np.int64)
        assert (h == -1).all(), h
        exact = [w.run_local() for w in big_dims_workers]              # and the exact path still works afterwards
        assert sum(nv for nv, _ in exact) > 20000
        for w in big_dims_workers:
            w.close()
    finally:
        torch.cuda.synchronize()
        N.check(L.sdfk_set_stream(None))
        torch.cuda.set_stream(torch.cuda.default_stream())


def test_c4_union8_1024_eight_slabs_equal_whole(gpu):
    """BASELINE config C4 at its full size on ONE GPU: 1024^3 CSG union of 8 primitives, meshed
    whole and as the 8 Z slabs the 8-GPU run uses (queued back to back, packed by the device,
    rebased in one launch).  Size-independent properties + slabs == whole, bit for bit."""
    import torch
    from sdfkit_amd import dist as D
    from tests import slab_worker as W
    scene, sdf = S.CATALOGUE["union8"]()
    n, world = 1024, 8
    MN, MX = [-2.0] * 3, [2.0] * 3        # bounds -2..2 (primitives reach +-1.6), clipToBounds
    L = N.lib()
    N.bind_torch_stream()
    try:
        whole = sdf.ToMesh(MN, MX, n, n, n)
        nv, t = len(whole.Vertices), whole.Triangles
        assert nv > 2_500_000 and t.min() == 0 and t.max() == nv - 1   # (1.6 M on the wider -2.8125..2.8125 box of round 1)
        first = np.full(nv, len(t), np.int64)
        np.minimum.at(first, t, np.arange(len(t)))
        assert np.all(np.diff(first) > 0)                       # numbered in order of first reference
        assert np.abs(np.linalg.norm(whole.Normals.astype(np.float64), axis=1) - 1).max() < 1e-5
        assert np.all(whole.Min >= np.float32(-2.0)) and np.all(whole.Max <= np.float32(2.0))
        tri = t.reshape(-1, 3)
        e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
        key = e[:, 0].astype(np.int64) * nv + e[:, 1]
        _, cnt = np.unique(key, return_counts=True)
        assert np.all(cnt == 2)                                  # closed surface
        workers = [W.GpuSlabWorker(sdf, MN, MX, n, n, n, r, world, True, 0.0) for r in range(world)]
        counts = [w.run_local() for w in workers]
        assert sum(a for a, _ in counts) == nv and sum(b for _, b in counts) == len(t)
        stride = (max(D.SLAB_HEADER_BYTES + 36 * a + 4 * b for a, b in counts) + 4096 + 255) // 256 * 256
        g = torch.zeros((world, stride), dtype=torch.uint8, device="cuda")
        for r, w in enumerate(workers):
            w.enqueue(g[r])
        N.check(L.sdfk_slabs_rebase(C.c_void_p(g.data_ptr()), world, stride))
        torch.cuda.synchronize()
        V, Cc, Nn, T, bmin, bmax = D.unpack_self_describing(g.cpu().numpy())
        for w in workers:
            w.close()
        assert np.array_equal(T, whole.Triangles) and np.array_equal(V, whole.Vertices)
        assert np.array_equal(Cc, whole.Colors) and np.array_equal(Nn, whole.Normals, equal_nan=True)
        assert np.array_equal(bmin, whole.Min) and np.array_equal(bmax, whole.Max)
    finally:
        torch.cuda.synchronize()
        N.check(L.sdfk_set_stream(None))
        torch.cuda.set_stream(torch.cuda.default_stream())


@pytest.mark.parametrize("dims", [(32, 32, 32), (40, 12, 36), (65, 7, 8), (1, 9, 12), (70, 3, 4), (130, 5, 64)])
@pytest.mark.parametrize("name", ["readme_repeat_xy", "sphere_w", "union8"])
def test_explicit_clip_keeps_the_cached_views(gpu, name, dims):
    """SampleSdf -> ClipToBounds() -> CreateMesh: the clip patches the cached sign bits and the
    remembered sampling program instead of dropping them, so the mesh must still equal the
    oracle's -- and no dense pass over the volume (k_signbits) or corner gather may run."""
    scene, sdf = S.CATALOGUE[name]()
    ov, oc = O.sample(scene, MN, MX, *dims)
    O.clip_to_bounds(ov, MN, MX)
    om = O.march(ov, oc, MN, MX)
    L = N.lib()
    vol = Voxels.SampleSdf(sdf, MN, MX, *dims)
    vol.ClipToBounds()
    N.check(L.sdfk_profile_reset())
    N.check(L.sdfk_profile_enable(1))
    m = MarchingCubes.CreateMesh(vol)
    N.check(L.sdfk_profile_enable(0))
    ran = set(N.profile_snapshot())
    assert_mesh_equal(m, om)
    assert np.array_equal(vol.Values, ov)
    if min(dims) > 1:
        assert "k_signbits" not in ran, ran
        assert "k_gather_corners" not in ran or not N.get_option(N.OPT_CORNER_EVAL), ran   # (SDFK_NO_CORNER_EVAL=1: the gather path, on purpose)
    m2 = MarchingCubes.CreateMesh(vol, 0.25)    # another iso value: bits are recomputed, corners still re-evaluated
    assert_mesh_equal(m2, O.march(ov, oc, MN, MX, iso=0.25))


def test_held_mesh_survives_many_other_jobs(gpu):
    """A queued mesh that is read only after far more other jobs than there are result slots (64):
    its slot must not have been handed to anyone else in the meantime (found by tools/stress.py)."""
    scene, sdf = S.CATALOGUE["union8"]()
    dims = (24, 52, 28)
    om = oracle_mesh(scene, MN, MX, dims, True)
    assert_mesh_equal(sdf.ToMesh(MN, MX, *dims), om)                 # size hint for this shape
    held = raw_sample_march(sdf, MN, MX, dims, True)
    other_scene, other = S.CATALOGUE["readme_repeat_xy"]()
    oo = oracle_mesh(other_scene, MN, MX, (40, 36, 44), True)
    L = N.lib()
    for k in range(150):                                             # resolved, dropped and two-stage jobs in between
        if k % 3 == 0:
            L.sdfk_mesh_free(raw_sample_march(other, MN, MX, (40, 36, 44), True))
        elif k % 3 == 1:
            assert_mesh_equal(other.ToMesh(MN, MX, 40, 36, 44), oo)
        else:
            vol = Voxels.SampleSdf(other, MN, MX, 40, 36, 44)
            vol.ClipToBounds()
            assert_mesh_equal(MarchingCubes.CreateMesh(vol), oo)
    assert_mesh_equal(Mesh._from_handle(held), om)


def test_stream_placement_keeps_the_lanes_apart(gpu):
    """sdfk_init measured which of its streams run side by side (csrc/lib_context.hip, "stream placement"): lanes 1-3 sit in three
    different classes, none of them lane 0's; the stream kept for a sharded rank's exchange is in lane 0's class."""
    from sdfkit_amd import _native as N
    p = N.stream_placement()
    if not N.get_option(N.OPT_STREAM_PLACEMENT):
        assert not p["measured"]
        return
    assert p["measured"] and 2 <= p["classes"] <= 8, p
    lanes = p["lanes"]
    if p["classes"] >= 4:
        assert len({lanes[0], lanes[1], lanes[2], lanes[3]}) == 4 and -1 not in lanes[1:4], p
        assert p["exchange"] in (-1, lanes[0]), p
