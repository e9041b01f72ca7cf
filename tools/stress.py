#!/usr/bin/env python3
"""GPU soak test: a random interleaving of the API (fused ToMesh jobs read late / never read,
two-stage volumes with explicit clip and edits, other iso values and steps, ray-marched frames,
large grids under a randomly chosen SDFK_OPT_ELIDE_VOLUME, programs of one structure with new constants,
recycled mesh arrays, the SDF at random points, a local node of two ranks sharing the GPU -- device mesh and host-array form) for SECONDS (default 60), every result checked against oracle results computed up front."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O
from sdfkit_amd import MarchingCubes, Mesh, RayMarcher, Voxels
from sdfkit_amd import _native as N
from tests import scenes as S

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MN, MX = [-2.8125] * 3, [2.8125] * 3
CASES = []
for name in ("readme_repeat_xy", "union8", "sphere_w", "sdf_with_color", "repeat_xz_box"):
    for dims in ((40, 36, 44), (64, 64, 64), (33, 30, 26), (24, 52, 28)):
        scene, sdf = S.CATALOGUE[name]()
        v, c = O.sample(scene, MN, MX, *dims)
        O.clip_to_bounds(v, MN, MX)
        CASES.append(dict(name=name, dims=dims, sdf=sdf, scene=scene, v=v, c=c, m=O.march(v, c, MN, MX), m25=None, ray=None))
# grids above the captured-graph limit: the only ones SDFK_OPT_ELIDE_VOLUME applies to (0 stored, 1 sign-only, 2 block culling)
BIG = []
for name, dims in (("readme_repeat_xy", (264, 260, 256)), ("sphere_w", (272, 250, 260)), ("union8", (260, 264, 256))):
    scene, sdf = S.CATALOGUE[name]()
    v, c = O.sample(scene, MN, MX, *dims)
    O.clip_to_bounds(v, MN, MX)
    BIG.append(dict(name=name, dims=dims, sdf=sdf, m=O.march(v, c, MN, MX)))
N.init()
L = N.lib()


def same(m, o):
    return (np.array_equal(m.Triangles, o.triangles) and np.array_equal(m.Vertices, o.vertices) and np.array_equal(m.Colors, o.colors)
            and np.array_equal(m.Normals, o.normals, equal_nan=True) and np.array_equal(m.Min, o.min) and np.array_equal(m.Max, o.max))


def raw(case):
    h = C.c_void_p()
    N.check(L.sdfk_sample_march(case["sdf"].program(), N.f3(MN), N.f3(MX), *case["dims"], 1, C.c_float(0.0), 1, C.byref(h)))
    return h


from sdfkit_amd import dist as D
node = D.Node([0, 0])                            # two ranks (threads of the library, a device context each) sharing GPU 0
held, ops, t0 = [], 0, time.time()
while time.time() - t0 < secs:
    k = int(rng.integers(0, 13))
    case = CASES[int(rng.integers(0, len(CASES)))]
    if k == 11:                                  # the node: the whole mesh on the first rank's device, or straight into host arrays
        mesh = (node.to_mesh if rng.random() < 0.5 else node.to_mesh_host)(case["sdf"], MN, MX, *case["dims"])
        assert same(mesh, case["m"]), ("node", case["name"], case["dims"])
    elif k == 12:                                # SdfEx.Sample at random points
        pts = rng.uniform(-3, 3, (257, 3)).astype(np.float32)
        got = case["sdf"].Sample(pts)
        want = np.stack([O.eval_point(case["scene"], p) for p in pts[:64]])
        assert np.array_equal(got[:64, 3].view(np.uint32), want[:, 3].view(np.uint32)), ("points", case["name"])
    elif k == 8:                                   # a large grid under a random volume-elision mode, mesh arrays recycled
        big = BIG[int(rng.integers(0, len(BIG)))]
        with N.option(N.OPT_ELIDE_VOLUME, int(rng.integers(0, 3))), N.option(N.OPT_COLOR_PASSES, int(rng.integers(0, 3))):
            mesh = big["sdf"].ToMesh(MN, MX, *big["dims"])
        assert same(mesh, big["m"]), ("big", big["name"], N.get_option(N.OPT_ELIDE_VOLUME))
        mesh.Recycle()
    elif k == 9:                                 # the sphere's structure with a radius nobody has seen: no compile, the oracle's mesh
        from sdfkit_amd import Sdfs
        r = float(np.float32(0.4 + 0.9 * rng.random()))
        dims = ((40, 36, 44), (64, 64, 64))[int(rng.integers(0, 2))]
        sc = O.Scene(); sc.sphere_w(r)
        v, c = O.sample(sc, MN, MX, *dims)
        O.clip_to_bounds(v, MN, MX)
        assert same(Sdfs.Sphere(r).ToMesh(MN, MX, *dims), O.march(v, c, MN, MX)), ("radius", r, dims)
    elif k == 10 and held:                       # read a held mesh and hand its arrays back at once
        case2, h = held.pop(int(rng.integers(0, len(held))))
        mesh = Mesh._from_handle(h)
        assert same(mesh, case2["m"]), ("late read + recycle", case2["name"], case2["dims"])
        mesh.Recycle()
    elif k <= 2:                                 # queue a fused job, read it later (or never)
        held.append((case, raw(case)))
    elif k == 3 and held:                        # read one of the held meshes, out of order
        case2, h = held.pop(int(rng.integers(0, len(held))))
        assert same(Mesh._from_handle(h), case2["m"]), ("late read", case2["name"], case2["dims"])
    elif k == 4 and held:                        # drop one unread
        L.sdfk_mesh_free(held.pop(int(rng.integers(0, len(held))))[1])
    elif k == 5:                                 # two-stage: sample (colour volumes in one pass or two, at random), explicit clip, mesh; then another iso
        with N.option(N.OPT_COLOR_PASSES, int(rng.integers(0, 3))):
            vol = Voxels.SampleSdf(case["sdf"], MN, MX, *case["dims"])
            if rng.random() < 0.25:              # and the volume itself, Values and Colors, against the oracle's (before the explicit clip)
                v0, c0 = O.sample(case["scene"], MN, MX, *case["dims"])
                assert np.array_equal(vol.Values, v0) and np.array_equal(vol.Colors, c0), ("volume", case["name"], case["dims"], N.get_option(N.OPT_COLOR_PASSES))
        vol.ClipToBounds()
        assert same(MarchingCubes.CreateMesh(vol), case["m"]), ("two-stage", case["name"], case["dims"])
        if case["m25"] is None:
            case["m25"] = O.march(case["v"], case["c"], MN, MX, iso=0.25)
        assert same(MarchingCubes.CreateMesh(vol, 0.25), case["m25"]), ("iso", case["name"], case["dims"])
    elif k == 6:                                 # host-array volume (uploaded), edited in place
        v = case["v"].copy()
        v[3, 4, 5] = -v[3, 4, 5]
        assert same(MarchingCubes.CreateMesh(Voxels(v, case["c"], MN, MX)), O.march(v, case["c"], MN, MX)), ("edited", case["name"])
    else:                                        # a small ray-marched frame
        if case["ray"] is None:
            case["ray"] = O.raymarch(case["scene"], 48, 27, iterations=24)
        rm = RayMarcher(48, 27, case["sdf"])
        rm.DepthIterations = 24
        assert np.array_equal(rm.RenderDepth().Values, case["ray"][0], equal_nan=True)
        assert np.array_equal(rm.Render().Values, case["ray"][1], equal_nan=True)
    while len(held) > 12:
        case2, h = held.pop(0)
        assert same(Mesh._from_handle(h), case2["m"]), ("overflow read", case2["name"], case2["dims"])
    ops += 1
for case2, h in held:
    assert same(Mesh._from_handle(h), case2["m"])
node.close()
print(f"stress ok: {ops} operations in {time.time() - t0:.1f} s")
