"""Mesh.Transform(Matrix4x4) (Mesh.cs:47-64): the host form of the Python mirror (numpy, float32) and the device form
(sdfk_mesh_transform) against the oracle's restatement (orc_transform_arrays).  The reference holds no test of it beyond
the fixed T*S*T of CreateMesh (whose results the known-answer tests pin); arbitrary matrices are pinned oracle <-> product."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import Mesh


def _matrices():
    rng = np.random.default_rng(77)
    out = [np.eye(4, dtype=np.float32)]
    t = np.eye(4, dtype=np.float32); t[3, :3] = [1.5, -2.25, 0.75]
    out.append(t)
    s = np.diag([2.0, 0.5, -3.0, 1.0]).astype(np.float32)
    out.append(s)
    ang = 0.7
    r = np.eye(4, dtype=np.float32)
    r[0, 0], r[0, 1], r[1, 0], r[1, 1] = np.cos(ang), np.sin(ang), -np.sin(ang), np.cos(ang)
    out.append((r @ s @ t).astype(np.float32))
    for _ in range(3):
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = rng.uniform(-2, 2, (3, 3))
        m[3, :3] = rng.uniform(-5, 5, 3)
        out.append(m)
    out.append(np.diag([1.0, 1.0, 0.0, 1.0]).astype(np.float32))      # singular: Matrix4x4.Invert fails, normals become NaN
    return out


def _mesh_arrays(n=5000, seed=3):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    q = rng.standard_normal((n, 3)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
    q[7] = 0.0                                                         # a zero normal: 0 / 0
    return v, q


@pytest.mark.parametrize("k", range(8))
def test_host_transform_matches_the_oracle(k):
    M = _matrices()[k]
    v, q = _mesh_arrays()
    ov, oq, omin, omax = O.transform(v, q, M)
    m = Mesh(v.copy(), np.zeros_like(v), q.copy(), np.zeros(0, np.int32))
    m.Transform(M)
    assert np.array_equal(m.Vertices, ov) and np.array_equal(m.Normals, oq, equal_nan=True)
    assert np.array_equal(m.Min, omin) and np.array_equal(m.Max, omax)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(8))
def test_device_transform_matches_the_oracle(gpu, k):
    from sdfkit_amd import Sdfs
    from sdfkit_amd import _native as N
    L = N.lib()
    M = _matrices()[k]
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (48, 44, 40)
    sdf = Sdfs.Sphere(1.0)
    ref = sdf.ToMesh(mn, mx, *dims)
    ov, oq, omin, omax = O.transform(ref.Vertices, ref.Normals, M)
    NM = Mesh.normal_matrix(M)
    fa = lambda a: (C.c_float * 16)(*[float(x) for x in np.asarray(a, np.float32).ravel()])
    # Several calls in a row, default options: from the second sighting of the key on a lane the job is a captured launch graph
    # and the mesh handle BORROWS the job's arrays -- transformed in place like an owned mesh (the job stays busy until the
    # handle is freed), and the NEXT mesh of the same job comes out untransformed.
    for rep in range(14):
        h = C.c_void_p()
        N.check(L.sdfk_sample_march(sdf.program(), N.f3(mn), N.f3(mx), *dims, 1, C.c_float(0.0), 1, C.byref(h)))
        N.check(L.sdfk_mesh_transform(h, fa(M), fa(NM)))
        m = Mesh._from_handle(h)
        assert np.array_equal(m.Vertices, ov) and np.array_equal(m.Normals, oq, equal_nan=True), rep
        assert np.array_equal(m.Triangles, ref.Triangles)
        assert np.array_equal(m.Min, omin) and np.array_equal(m.Max, omax)
    jobs, launches = C.c_int64(), C.c_int64()
    N.check(L.sdfk_graph_stats(C.byref(jobs), C.byref(launches), None))
    if N.get_option(N.OPT_GRAPHS):
        assert launches.value >= 1          # (a captured job is built on the second sighting of its key on a lane and replayed from the third: some of the meshes above borrowed its arrays)
