"""Host-side plumbing of the C ABI on the GPU box: the device-to-host copy strategies of
sdfk_mesh_copy / sdfk_volume_download (threaded first touch, pinned staging), the on-disk cache of
JIT-compiled code objects, and calls arriving from threads the library has not seen before."""
import ctypes as C
import glob
import os
import threading

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import Mesh, SdfExprs, Sdfs, Voxels
from sdfkit_amd import _native as N
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu


def _raw(sdf, mn, mx, dims, clip):
    m = C.c_void_p()
    N.check(N.lib().sdfk_sample_march(sdf.program(), N.f3(mn), N.f3(mx), *dims, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m)))
    return m


@pytest.mark.parametrize("name,clip", [("sphere_w", False), ("readme_repeat_xy", True)])
def test_mesh_copy_strategies_agree(gpu, name, clip):
    """SDFK_OPT_COPY_MODE 0 (pre-fault on the pool + runtime copy), 1 (bounded pinned staging ring + pool memcpy) and
    2 (plain copies) deliver the same bytes; a mesh of > 1 MiB so that the helpers are really used."""
    scene, sdf = S.CATALOGUE[name]()
    mn, mx = ([-1.5] * 3, [1.5] * 3) if name == "sphere_w" else ([-2.8125] * 3, [2.8125] * 3)
    dims = (160, 152, 168)
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) * 36 + len(om.triangles) * 4 > (1 << 20)
    got = []
    for mode in (0, 1, 2, 1, 0):
        with N.option(N.OPT_COPY_MODE, mode):
            m = Mesh._from_handle(_raw(sdf, mn, mx, dims, clip))
            assert_mesh_equal(m, om)
            got.append(m)
    for m in got[1:]:
        for f in ("Vertices", "Colors", "Normals", "Triangles"):
            assert np.array_equal(getattr(m, f), getattr(got[0], f), equal_nan=True)


def test_mesh_copy_partial_and_unaligned_destinations(gpu):
    """NULL destinations are skipped; destinations need no alignment beyond their element type."""
    scene, sdf = S.sphere_w(1.0)
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (128, 128, 128)
    ref = Mesh._from_handle(_raw(sdf, mn, mx, dims, False))
    L = N.lib()
    for mode in (0, 1):
        with N.option(N.OPT_COPY_MODE, mode):
            h = _raw(sdf, mn, mx, dims, False)
            nv, ni = C.c_int64(), C.c_int64()
            N.check(L.sdfk_mesh_counts(h, C.byref(nv), C.byref(ni)))
            big = np.full(nv.value * 3 + ni.value + 64, -7.0, np.float32)
            v = big[1:1 + nv.value * 3]                      # 4-byte aligned only
            t = big[nv.value * 3 + 3:nv.value * 3 + 3 + ni.value].view(np.int32)
            N.check(L.sdfk_mesh_copy(h, v.ctypes.data, None, None, t.ctypes.data))
            assert np.array_equal(v.reshape(-1, 3), ref.Vertices) and np.array_equal(t, ref.Triangles)
            assert big[0] == -7.0 and big[1 + nv.value * 3] == -7.0 and big[-1] == -7.0     # nothing outside the arrays was written
            c = np.full((nv.value, 3), 5.0, np.float32)
            N.check(L.sdfk_mesh_copy(h, None, c.ctypes.data, None, None))
            assert not c.any()                                # W-only SDF: colours are zero (Voxels.cs:88-92)
            L.sdfk_mesh_free(h)


def test_volume_download_through_the_pool(gpu):
    scene, sdf = S.CATALOGUE["readme_repeat_xy"]()
    mn, mx, dims = [-2.8125] * 3, [2.8125] * 3, (96, 100, 104)
    ov, oc = O.sample(scene, mn, mx, *dims)
    for mode in (0, 1, 2):
        with N.option(N.OPT_COPY_MODE, mode):
            vol = Voxels.SampleSdf(sdf, mn, mx, *dims)
            assert np.array_equal(vol.Values, ov) and np.array_equal(vol.Colors, oc)
    w = Voxels.SampleSdf(Sdfs.Sphere(1.0), mn, mx, *dims)    # no colour array on the device: zeros come from the pool
    assert not w.Colors.any() and w.Colors.shape == dims + (3,)


def _stats():
    a, b, c = C.c_int64(), C.c_int64(), C.c_double()
    N.check(N.lib().sdfk_jit_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def test_code_object_cache_on_disk(gpu, tmp_path):
    """First creation of a program compiles with hiprtc and leaves a code object in the cache directory; the next
    creation of the same program (a new process would do the same) loads it; a damaged entry is recompiled."""
    N.check(N.lib().sdfk_set_cache_dir(str(tmp_path / "jit").encode()))
    idle = N.get_option(N.OPT_IDLE_PROGRAMS)
    N.set_option(N.OPT_IDLE_PROGRAMS, 0)   # (a structure's modules are unloaded with its last program: every build() below starts cold)
    try:
        mn, mx, dims = [-2.0] * 3, [2.0] * 3, (40, 44, 48)

        def build():
            return SdfExprs.Sphere(0.8125, (0.25, 0.5, 0.75)).RepeatX(1.5).ToSdf()   # a fresh Sdf: no in-process program yet

        scene = O.Scene()
        scene.f_repeat_x(scene.f_sphere(0.8125, (0.25, 0.5, 0.75)), 1.5)
        ov, oc = O.sample(scene, mn, mx, *dims)
        O.clip_to_bounds(ov, mn, mx)
        om = O.march(ov, oc, mn, mx)
        c0, h0, _ = _stats()
        assert_mesh_equal(build().ToMesh(mn, mx, *dims), om)
        c1, h1, ms = _stats()
        assert (c1, h1) == (c0 + 1, h0) and ms > 0
        files = glob.glob(str(tmp_path / "jit" / "*.co"))
        assert len(files) == 1 and os.path.getsize(files[0]) > 10000
        assert_mesh_equal(build().ToMesh(mn, mx, *dims), om)
        c2, h2, _ = _stats()
        assert (c2, h2) == (c1, h1 + 1)                      # loaded, not compiled
        # damage the code object (keep the key): the load fails, the library recompiles and replaces the entry
        raw = bytearray(open(files[0], "rb").read())
        raw[-4096:] = b"\\0" * 4096
        raw[len(raw) // 2:len(raw) // 2 + 4096] = b"\\xff" * 4096
        open(files[0], "wb").write(bytes(raw))
        assert_mesh_equal(build().ToMesh(mn, mx, *dims), om)
        c3, h3, _ = _stats()
        assert c3 == c2 + 1
        with N.option(N.OPT_CODE_CACHE, 0):
            assert_mesh_equal(build().ToMesh(mn, mx, *dims), om)
        c4, h4, _ = _stats()
        assert (c4, h4) == (c3 + 1, h3)
        # a directory somebody else could write to is refused (the cache stays off: compiled again, nothing stored)
        loose = tmp_path / "loose"
        loose.mkdir()
        os.chmod(loose, 0o777)
        N.check(N.lib().sdfk_set_cache_dir(str(loose).encode()))
        assert_mesh_equal(build().ToMesh(mn, mx, *dims), om)
        c5, h5, _ = _stats()
        assert (c5, h5) == (c4 + 1, h4) and not glob.glob(str(loose / "*.co"))
    finally:
        N.check(N.lib().sdfk_set_cache_dir(None))
        N.set_option(N.OPT_IDLE_PROGRAMS, idle)


def test_calls_from_other_threads(gpu):
    """The HIP current device is per thread; every entry point binds the calling thread to the library's device.
    Whole meshes built on threads the library has never seen, concurrently (the ABI serialises internally)."""
    scene, sdf = S.CATALOGUE["union8"]()
    mn, mx, dims = [-2.0] * 3, [2.0] * 3, (40, 36, 44)
    ov, oc = O.sample(scene, mn, mx, *dims)
    O.clip_to_bounds(ov, mn, mx)
    om = O.march(ov, oc, mn, mx)
    sdf.program()
    out, err = [None] * 4, []

    def work(k):
        try:
            for _ in range(3):
                out[k] = sdf.ToMesh(mn, mx, *dims)
        except Exception as e:   # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for m in out:
        assert_mesh_equal(m, om)


def test_mesh_arrays_come_from_the_pinned_arena(gpu):
    """Mesh._from_handle allocates V/C/N/T in the library's pinned host arena (plain DMA, recycled blocks);
    SDFK_PINNED_ARRAYS=0 gives ordinary numpy arrays with the same contents."""
    scene, sdf = S.sphere_w(1.0)
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (96, 96, 96)
    ov, oc = O.sample(scene, mn, mx, *dims)
    om = O.march(ov, oc, mn, mx)
    L = N.lib()
    ptrs = set()
    for _ in range(4):
        m = Mesh._from_handle(_raw(sdf, mn, mx, dims, False))
        assert_mesh_equal(m, om)
        assert m.Vertices.flags.writeable and m.Vertices.base is not None
        ptrs.add(m.Vertices.ctypes.data)
        del m
    assert len(ptrs) <= 2          # blocks of collected meshes are handed out again
    os.environ["SDFK_PINNED_ARRAYS"] = "0"
    try:
        assert_mesh_equal(Mesh._from_handle(_raw(sdf, mn, mx, dims, False)), om)
    finally:
        del os.environ["SDFK_PINNED_ARRAYS"]
    p = C.c_void_p()
    N.check(L.sdfk_host_alloc(3 << 20, C.byref(p)))
    assert p.value
    np.ctypeslib.as_array((C.c_uint8 * 64).from_address(p.value))[:] = 7    # plain host memory
    L.sdfk_host_free(p)
    L.sdfk_host_free(p)            # double free of an arena block is ignored
    seen = []
    for _ in range(8):             # the freed block comes back (possibly after others of its size class)
        q = C.c_void_p()
        N.check(L.sdfk_host_alloc(3 << 20, C.byref(q)))
        seen.append(q.value)
    assert p.value in seen
    for v in seen:
        L.sdfk_host_free(C.c_void_p(v))


@pytest.mark.parametrize("dims", [(9, 7, 13), (12, 10, 250), (6, 5, 64)])
def test_padded_rows_upload_download_round_trip(gpu, dims):
    """Device volumes pad their z rows to a multiple of 4 voxels; the host arrays stay dense [nx][ny][nz]
    (Voxels.cs:8-9): upload -> download is the identity for values and colours, whatever nz is, and meshing an
    uploaded volume equals the oracle's mesh of the same arrays."""
    L = N.lib()
    rng = np.random.default_rng(11)
    nx, ny, nz = dims
    vals = rng.uniform(-1, 1, dims).astype(np.float32)
    cols = rng.uniform(0, 1, dims + (3,)).astype(np.float32)
    mn, mx = [-1.0, -1.5, -2.0], [1.0, 1.5, 2.0]
    h = C.c_void_p()
    N.check(L.sdfk_volume_create(nx, ny, nz, N.f3(mn), N.f3(mx), 1, C.byref(h)))
    pitch = C.c_int32()
    N.check(L.sdfk_volume_row_pitch(h, C.byref(pitch)))
    assert pitch.value == (nz + 3) // 4 * 4
    N.check(L.sdfk_volume_upload(h, vals.ctypes.data, cols.ctypes.data))
    v2, c2 = np.full(dims, 9.0, np.float32), np.full(dims + (3,), 9.0, np.float32)
    N.check(L.sdfk_volume_download(h, v2.ctypes.data, c2.ctypes.data))
    assert np.array_equal(v2, vals) and np.array_equal(c2, cols)
    m = C.c_void_p()
    N.check(L.sdfk_march(h, C.c_float(0.0), 1, C.byref(m)))
    assert_mesh_equal(Mesh._from_handle(m), O.march(vals, cols, mn, mx))
    L.sdfk_volume_free(h)


def test_device_colours_of_a_w_only_mesh_are_zero_on_demand(gpu):
    """A .W-only program's mesh colours are all zero (Voxels.cs:88-92) and k_vertices does not store them (12 bytes per vertex saved per
    job): whoever asks for DEVICE colours -- sdfk_mesh_device_ptrs, sdfk_mesh_copy_device -- gets them zeroed then, whatever the buffer
    held before (the pool hands out recycled blocks)."""
    import torch
    L = N.lib()
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (96, 96, 96)
    colored = S.CATALOGUE["sdf_with_color"]()[1]
    plain = S.sphere_w(1.0)[1]
    for rep in range(3):
        # dirty the pool with a coloured mesh of a similar size first
        h0 = _raw(colored, mn, mx, dims, False)
        a0, b0 = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(h0, C.byref(a0), C.byref(b0)))
        L.sdfk_mesh_free(h0)
        h = _raw(plain, mn, mx, dims, False)
        a, b = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(h, C.byref(a), C.byref(b)))
        nv = a.value
        assert nv > 1000
        dst = torch.full((nv, 3), 5.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        N.check(L.sdfk_mesh_copy_device(h, None, C.c_void_p(dst.data_ptr()), None, None))
        N.check(L.sdfk_synchronize())
        assert not dst.cpu().numpy().any()
        pv, pc = C.c_void_p(), C.c_void_p()
        N.check(L.sdfk_mesh_device_ptrs(h, C.byref(pv), C.byref(pc), None, None))
        N.check(L.sdfk_synchronize())
        assert pc.value
        # (the array behind that pointer is now valid: a second device copy takes it from there)
        dst2 = torch.full((nv, 3), 3.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        N.check(L.sdfk_mesh_copy_device(h, None, C.c_void_p(dst2.data_ptr()), None, None))
        N.check(L.sdfk_synchronize())
        assert not dst2.cpu().numpy().any()
        # the host copy agrees (it never reads the device colours of such a mesh)
        m = Mesh._from_handle(h)
        assert not m.Colors.any() and len(m.Vertices) == nv
