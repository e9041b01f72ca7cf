"""Committed golden vectors (tests/golden/meshes.json, small_meshes.npz; made by
tools/gen_golden.py from the CPU oracle in the build container).

CPU: the oracle still reproduces them (pins the oracle against drift).
GPU (-m gpu): the HIP path, through the C ABI, reproduces them bit for bit -- without the oracle
in the loop: counts, AABB and FNV-1a-64 of every output array; full arrays for grids <= 32."""
import json
import os

import numpy as np
import pytest

from tests.test_gpu_parity import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "meshes.json")))
SMALL = np.load(os.path.join(ROOT, "tests", "golden", "small_meshes.npz"))


def fnv(a):
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).view(np.uint8).ravel().tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def random_volume(seed):
    rng = np.random.default_rng(seed)
    shape = (17, 17, 17) if seed < 4 else (9 + seed, 21, 14)
    v = rng.uniform(-1, 1, shape).astype(np.float32)
    c = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    return v, c, [-1, -2, -3], [1.5, 2.5, 3.5]


def check(name, vertices, colors, normals, triangles, mn, mx):
    f = FIX[name]
    assert (len(vertices), len(triangles)) == (f["vertices"], f["indices"]), name
    assert [float(x) for x in mn] == f["min"] and [float(x) for x in mx] == f["max"], name
    assert fnv(triangles) == f["fnv_triangles"], f"{name}: triangle topology"
    assert fnv(vertices) == f["fnv_vertices"], f"{name}: vertex positions"
    assert fnv(colors) == f["fnv_colors"], f"{name}: colours"
    assert fnv(normals) == f["fnv_normals"], f"{name}: normals"
    if f"{name}.vertices" in SMALL:
        assert np.array_equal(vertices, SMALL[f"{name}.vertices"]) and np.array_equal(triangles, SMALL[f"{name}.triangles"])
        assert np.array_equal(colors, SMALL[f"{name}.colors"]) and np.array_equal(normals, SMALL[f"{name}.normals"], equal_nan=True)


def test_fixture_counts_are_the_reference_literals():
    for name, _, _, _, _, _, expect in GOLDEN:
        assert FIX[name]["vertices"] == expect   # Tests/MarchingCubesTests.cs, Tests/SdfTests.cs, SURVEY C1


@pytest.mark.parametrize("case", GOLDEN, ids=[g[0] for g in GOLDEN])
def test_oracle_reproduces_fixture(case):
    from oracle import oracle as O
    name, mk, mn, mx, n, clip, _ = case
    if n > 64:
        pytest.skip("the 128^3 case takes the CPU oracle a while; covered on the GPU box and by test_oracle_golden")
    scene, _ = mk()
    v, c = O.sample(scene, mn, mx, n, n, n)
    if clip:
        O.clip_to_bounds(v, mn, mx)
    assert fnv(v) == FIX[name]["fnv_values"]
    m = O.march(v, c, mn, mx)
    check(name, m.vertices, m.colors, m.normals, m.triangles, m.min, m.max)


@pytest.mark.parametrize("seed", range(8))
def test_oracle_reproduces_random_volume_fixture(seed):
    from oracle import oracle as O
    v, c, mn, mx = random_volume(seed)
    m = O.march(v, c, mn, mx)
    check(f"random{seed}", m.vertices, m.colors, m.normals, m.triangles, m.min, m.max)
    assert int(m.impossible13) == FIX[f"random{seed}"]["impossible13"] and len(m.cells) == FIX[f"random{seed}"]["active_cells"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN, ids=[g[0] for g in GOLDEN])
def test_hip_reproduces_fixture(gpu, case):
    name, mk, mn, mx, n, clip, _ = case
    _, sdf = mk()
    m = sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip)
    check(name, m.Vertices, m.Colors, m.Normals, m.Triangles, m.Min, m.Max)
    v = sdf.ToVoxels(mn, mx, n, n, n, clipToBounds=clip)
    assert fnv(v.Values) == FIX[name]["fnv_values"]
    assert m.ActiveCells == FIX[name]["active_cells"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_hip_reproduces_random_volume_fixture(gpu, seed):
    from sdfkit_amd import MarchingCubes, Voxels
    v, c, mn, mx = random_volume(seed)
    m = MarchingCubes.CreateMesh(Voxels(v, c, mn, mx))
    check(f"random{seed}", m.Vertices, m.Colors, m.Normals, m.Triangles, m.Min, m.Max)
    assert m.ImpossibleCase13Cells == FIX[f"random{seed}"]["impossible13"] and m.ActiveCells == FIX[f"random{seed}"]["active_cells"]


RAY = sorted(k[4:] for k in FIX if k.startswith("ray."))


@pytest.mark.parametrize("name", RAY)
def test_oracle_reproduces_raymarch_fixture(name):
    from oracle import oracle as O
    from tests import scenes as S
    f = FIX[f"ray.{name}"]
    scene, _ = S.CATALOGUE[name]()
    view = O.look_at(*f["camera"]) if f["camera"] else None
    d, rgb = O.raymarch(scene, f["width"], f["height"], view=view, iterations=f["iterations"])
    assert fnv(d) == f["fnv_depth"] and fnv(rgb) == f["fnv_rgb"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", RAY)
def test_hip_reproduces_raymarch_fixture(gpu, name):
    from sdfkit_amd import Matrix4x4, RayMarcher
    from tests import scenes as S
    f = FIX[f"ray.{name}"]
    _, sdf = S.CATALOGUE[name]()
    rm = RayMarcher(f["width"], f["height"], sdf)
    rm.DepthIterations = f["iterations"]
    if f["camera"]:
        rm.ViewTransform = Matrix4x4.CreateLookAt(*f["camera"])
    d = rm.RenderDepth()
    assert fnv(d.Values) == f["fnv_depth"]
    assert float(d[f["width"] // 2, f["height"] // 2]) == f["depth_centre"] and float(d[0, 0]) == f["depth_corner"]
    assert fnv(rm.Render().Values) == f["fnv_rgb"]
