"""MeshArrayPool: the exact-length arrays of a mesh that has been handed back serve the next mesh of the same size
(what the C# shim's Mesh.FromNative / Mesh.Recycle do, shim/SdfKit.Hip/Voxels.Hip.cs; Mesh.cs:10-13 is four managed arrays)."""
import numpy as np
import pytest

from sdfkit_amd.api import Mesh, MeshArrayPool


def test_pool_policy_on_the_host():
    made = []

    def alloc(shape, dtype):
        a = np.empty(shape, dtype)
        made.append(a)
        return a

    p = MeshArrayPool(alloc=alloc, per_key=2, max_bytes=3 * 1200)
    a = p.rent((100, 3), np.float32)
    b = p.rent((100, 3), np.float32)
    assert (p.hits, p.misses) == (0, 2) and a is not b and a.shape == (100, 3)
    p.give_back(a)
    assert p.rent((100, 3), np.float32) is a and p.hits == 1            # exact size: a hit
    assert p.rent((101, 3), np.float32) is not a and p.misses == 3      # another length: a miss (arrays are exact-length)
    assert p.rent((300,), np.int32) is not a                            # same bytes, other type / shape: a miss
    p.give_back(a); p.give_back(b); p.give_back(np.empty((100, 3), np.float32))
    assert len(p._free[(np.dtype(np.float32).str, (100, 3))]) == 2      # at most per_key arrays per size
    p.give_back(np.empty((50, 3), np.float32)); p.give_back(np.empty((51, 3), np.float32)); p.give_back(np.empty((52, 3), np.float32))
    assert p.bytes <= 3 * 1200 and (np.dtype(np.float32).str, (100, 3)) not in p._free    # the oldest size went first
    p.give_back(np.empty((0, 3), np.float32))                           # empty arrays are not kept
    p.clear()
    assert p.bytes == 0 and not p._free


def test_recycle_empties_the_mesh():
    p = MeshArrayPool(alloc=np.empty)
    v = p.rent((7, 3), np.float32)
    m = Mesh(v, p.rent((7, 3), np.float32), p.rent((7, 3), np.float32), p.rent((12,), np.int32))
    m._pool = p
    m.Recycle()
    assert len(m.Vertices) == 0 and len(m.Triangles) == 0 and m._pool is None
    assert p.rent((7, 3), np.float32) is not None and p.hits == 1
    m.Recycle()          # a second call returns nothing twice
    assert p.bytes == 2 * 84 + 48


@pytest.mark.gpu
def test_the_next_mesh_of_the_same_size_gets_the_recycled_arrays():
    from sdfkit_amd import Sdfs
    from sdfkit_amd import _native as N
    N.init(0)
    mn, mx, n = [-1.5] * 3, [1.5] * 3, 56
    sdf = Sdfs.Sphere(1.0)
    ref = sdf.ToMesh(mn, mx, n, n, n)
    want = [x.copy() for x in (ref.Vertices, ref.Colors, ref.Normals, ref.Triangles)]
    pool = Mesh.Pool
    h0, m0 = pool.hits, pool.misses
    addr = [x.ctypes.data for x in (ref.Vertices, ref.Colors, ref.Normals, ref.Triangles)]
    for x in (ref.Vertices, ref.Colors, ref.Normals):
        x[:] = 5.0                             # whatever the last owner left behind
    ref.Triangles[:] = -1
    ref.Recycle()
    again = sdf.ToMesh(mn, mx, n, n, n)
    assert pool.hits == h0 + 4 and pool.misses == m0
    assert sorted(x.ctypes.data for x in (again.Vertices, again.Colors, again.Normals)) == sorted(addr[:3]) and again.Triangles.ctypes.data == addr[3]
    for got, w in zip((again.Vertices, again.Colors, again.Normals, again.Triangles), want):
        assert np.array_equal(got, w)          # (the colours of a .W-only program are cleared by the library: a recycled array is not zero)
    other = Sdfs.Sphere(0.9).ToMesh(mn, mx, n, n, n)      # another size: a miss, fresh arrays
    assert len(other.Vertices) != len(again.Vertices) and pool.misses == m0 + 4
