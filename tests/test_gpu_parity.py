"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the
same inputs.  Bars (BASELINE.json north_star): sampled values bit-exact; cell case indices
and triangle topology bit-exact; vertex positions within 1e-5 abs (we additionally record
whether they are bit-exact); normals/colours within 1e-5 (unpinned by the reference)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import MarchingCubes, Sdfs, Voxels
from sdfkit_amd import _native as N
from tests import scenes as S

pytestmark = pytest.mark.gpu

POS_TOL = 1e-5      # north_star: "within 1e-5 abs on interpolated vertex positions"
ATTR_TOL = 1e-5


def assert_mesh_equal(m, o, exact=True):
    """m: sdfkit_amd.Mesh (GPU), o: OracleMesh."""
    assert len(m.Vertices) == len(o.vertices), (len(m.Vertices), len(o.vertices))
    assert len(m.Triangles) == len(o.triangles)
    assert np.array_equal(m.Triangles, o.triangles), "triangle topology differs"
    if len(o.vertices) == 0:
        return
    dv = np.abs(m.Vertices.astype(np.float64) - o.vertices.astype(np.float64))
    assert np.nanmax(dv) <= POS_TOL, f"vertex positions differ by {np.nanmax(dv)}"
    dc = np.abs(m.Colors.astype(np.float64) - o.colors.astype(np.float64))
    assert np.nanmax(dc) <= ATTR_TOL, f"colours differ by {np.nanmax(dc)}"
    nan_m, nan_o = np.isnan(m.Normals), np.isnan(o.normals)
    assert np.array_equal(nan_m, nan_o)
    dn = np.abs(np.nan_to_num(m.Normals).astype(np.float64) - np.nan_to_num(o.normals).astype(np.float64))
    assert dn.max() <= ATTR_TOL, f"normals differ by {dn.max()}"
    if exact:  # stronger than the bar: the double-precision cell math is reproduced exactly
        assert np.array_equal(m.Vertices, o.vertices)
        assert np.array_equal(m.Colors, o.colors)
        assert np.array_equal(m.Normals, o.normals, equal_nan=True)
    assert np.array_equal(m.Min, o.min) and np.array_equal(m.Max, o.max)


# ---------------------------------------------------------------------------
# sampling (Voxels.SampleSdf + ClipToBounds)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(S.CATALOGUE))
@pytest.mark.parametrize("dims,clip", [((32, 32, 32), False), ((17, 23, 29), True), ((40, 12, 36), True)])
def test_sample_bit_exact(gpu, name, dims, clip):
    scene, sdf = S.CATALOGUE[name]()
    mn, mx = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    v = sdf.ToVoxels(mn, mx, *dims, clipToBounds=clip)
    assert np.array_equal(v.Values, ov), f"{name}: values differ (max {np.abs(v.Values - ov).max()})"
    assert np.array_equal(v.Colors, oc), f"{name}: colours differ"
    assert (v.NX, v.NY, v.NZ) == dims
    assert np.array_equal(np.array([v.DX, v.DY, v.DZ]), O.cell_size(mn, mx, *dims))


@pytest.mark.parametrize("name", ["readme_repeat_xy", "sphere_w", "union8"])
@pytest.mark.parametrize("dims,clip", [((10, 12, 260), True), ((9, 7, 516), False), ((16, 5, 256), True), ((8, 6, 768), False),
                                       ((12, 9, 100), True), ((11, 3, 262), True), ((10, 12, 250), True), ((7, 6, 510), False),
                                       ((9, 4, 7), True), ((8, 8, 5), False), ((6, 2, 258), True), ((5, 3, 257), True)])
def test_sample_long_rows_every_kernel_shape(gpu, name, dims, clip):
    """The fused sampler has three shapes: z tiles of one row (nz % 256 == 0), 256-voxel chunks of the
    (y, z) plane (nz % 4 == 0; a chunk spans two rows when nz >= 256, several when nz < 256), rows of any length.
    Values, colours and the mesh built from the kernel's own sign bits must not depend on the shape."""
    scene, sdf = S.CATALOGUE[name]()
    mn, mx = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    v = sdf.ToVoxels(mn, mx, *dims, clipToBounds=clip)
    assert np.array_equal(v.Values, ov) and np.array_equal(v.Colors, oc)
    assert_mesh_equal(v.ToMesh(), O.march(ov, oc, mn, mx))
    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims, clipToBounds=clip), O.march(ov, oc, mn, mx))


def test_sample_instance_then_clip(gpu):
    scene, sdf = S.sphere_w(2.0)
    v = Voxels.SampleSdf(sdf, [-1] * 3, [1] * 3, 10, 10, 10)
    ov, _ = O.sample(scene, [-1] * 3, [1] * 3, 10, 10, 10)
    assert np.array_equal(v.Values, ov)
    v.ClipToBounds()
    O.clip_to_bounds(ov, [-1] * 3, [1] * 3)
    assert np.array_equal(v.Values, ov)


def test_sqrt_div_correctly_rounded(gpu):
    """the JIT kernel's sqrt and divide must be correctly rounded."""
    from sdfkit_amd import Sdf, Vec4, MathF
    sdf = Sdf(lambda p: Vec4(MathF.Sqrt(abs(p.x * p.y)), p.x / p.z, p.y / (p.z * p.z), MathF.Sqrt(p.z)), True)
    mn, mx = [0.001, 0.37, 1e-3], [977.0, 1.63, 3.0]
    v = sdf.ToVoxels(mn, mx, 64, 64, 256, clipToBounds=False)
    x = np.array([O.sample_position(mn, mx, 64, 64, 256, i)[0] for i in range(64)], np.float32)
    y = np.array([O.sample_position(mn, mx, 64, 64, 256, i * 64)[1] for i in range(64)], np.float32)
    z = np.array([O.sample_position(mn, mx, 64, 64, 256, i * 64 * 64)[2] for i in range(256)], np.float32)
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    assert np.array_equal(v.Values, np.sqrt(Z))
    assert np.array_equal(v.Colors[..., 0], np.sqrt(np.abs(X * Y)))
    assert np.array_equal(v.Colors[..., 1], X / Z)
    assert np.array_equal(v.Colors[..., 2], Y / (Z * Z))


def test_sqrt_whole_float_range(gpu):
    """The JIT's sqrt takes a short path when every lane of a wavefront has a normal operand >= 2^-96 and the
    general expansion otherwise: operands from negative / zero / denormal to overflow, mixed within wavefronts
    (lanes run along z)."""
    from sdfkit_amd import Sdf, Vec4, MathF

    def f(p):
        y2 = p.y * p.y; y4 = y2 * y2; y8 = y4 * y4
        z2 = p.z * p.z; z4 = z2 * z2; z8 = z4 * z4
        return Vec4(MathF.Sqrt(p.x * y8), MathF.Sqrt((y8 * 1e30) * (z8 * 1e12)), MathF.Sqrt(((y8 * y8) * z8) * abs(p.x)),
                    MathF.Sqrt((p.x * y8) * z8))
    sdf = Sdf(f, True)
    mn, mx, n = [-0.5, 0.0, 0.0], [7.5, 2.56, 0.512], (16, 128, 512)
    v = sdf.ToVoxels(mn, mx, *n, clipToBounds=False)
    D = [np.float32((np.float32(mx[a]) - np.float32(mn[a])) / np.float32(n[a])) for a in range(3)]
    g = [(np.float32(mn[a]) + np.float32(0.5) * D[a]) + np.arange(n[a], dtype=np.float32) * D[a] for a in range(3)]
    assert all(np.array_equal(g[a], [O.sample_position(mn, mx, *n, i * (1, n[0], n[0] * n[1])[a])[a] for i in range(n[a])]) for a in range(3))
    X, Y, Z = np.meshgrid(*g, indexing="ij")
    with np.errstate(all="ignore"):
        y8 = ((Y * Y) * (Y * Y)) * ((Y * Y) * (Y * Y))
        z8 = ((Z * Z) * (Z * Z)) * ((Z * Z) * (Z * Z))
        ops = [X * y8, (y8 * np.float32(1e30)) * (z8 * np.float32(1e12)), ((y8 * y8) * z8) * np.abs(X), (X * y8) * z8]
        want = [np.sqrt(o) for o in ops]
    allops = np.concatenate([o.ravel() for o in ops])
    tiny = np.float32(2.0) ** -96
    assert (allops == 0).any() and np.isinf(allops).any() and (allops < 0).any()
    assert ((allops > 0) & (allops < 1e-38)).any() and ((allops >= 1e-38) & (allops < tiny)).any() and (allops > 1e30).any()
    mixed = ((ops[3] > 0) & (ops[3] < tiny)).any(axis=2) & (ops[3] >= tiny).any(axis=2)
    assert mixed.any()                                     # rows (= wavefronts) holding both kinds of operand
    for w_, g_ in zip(want, [v.Colors[..., 0], v.Colors[..., 1], v.Colors[..., 2], v.Values]):
        np.testing.assert_array_equal(g_, w_)              # (NaN == NaN here)


@pytest.mark.parametrize("consts", [(1.125, 6.0, 0.1), (3.0, -7.0, 1e-3), (12345.678, 0.3, 255.0)])
def test_division_by_constants_correctly_rounded(gpu, consts):
    """x / c must be the correctly rounded quotient, signed zeros included: operands from zero / denormal to
    overflow, both signs, mixed within wavefronts (also the regression test for any future short form of the
    division by a program constant: DESIGN.md 5, "tried and dropped")."""
    from sdfkit_amd import Sdf, Vec4
    c0, c1, c2 = (np.float32(c) for c in consts)

    def f(p):
        z2 = p.z * p.z; z4 = z2 * z2; z8 = z4 * z4
        big = (p.y * z8) * 1e30
        return Vec4(((p.x - p.x) * p.y) / float(c0), (p.y * z8) / float(c1), big / float(c2), ((p.x * z8) * z8) / float(c0))
    sdf = Sdf(f, True)
    mn, mx, n = [-4.0, -0.5, 0.0], [4.0, 7.5, 0.512 * 40], (16, 16, 512)
    v = sdf.ToVoxels(mn, mx, *n, clipToBounds=False)
    D = [np.float32((np.float32(mx[a]) - np.float32(mn[a])) / np.float32(n[a])) for a in range(3)]
    g = [(np.float32(mn[a]) + np.float32(0.5) * D[a]) + np.arange(n[a], dtype=np.float32) * D[a] for a in range(3)]
    X, Y, Z = np.meshgrid(*g, indexing="ij")
    with np.errstate(all="ignore"):
        z8 = ((Z * Z) * (Z * Z)) * ((Z * Z) * (Z * Z))
        nums = [(X - X) * Y, Y * z8, (Y * z8) * np.float32(1e30), (X * z8) * z8]   # [0]: zeros of both signs
        want = [nums[0] / c0, nums[1] / c1, nums[2] / c2, nums[3] / c0]
    allx = np.abs(np.concatenate([x.ravel() for x in nums]))
    assert np.isinf(allx).any() and (allx > 2.0 ** 32).any() and ((allx > 0) & (allx < 2.0 ** -32)).any()
    assert ((allx >= 2.0 ** -32) & (allx <= 2.0 ** 32)).mean() > 0.2 and np.signbit(nums[0]).any() and not np.signbit(nums[0]).all()
    for w_, g_ in zip(want, [v.Colors[..., 0], v.Colors[..., 1], v.Colors[..., 2], v.Values]):
        np.testing.assert_array_equal(g_, w_)
        assert np.array_equal(np.signbit(g_), np.signbit(w_))   # signed zeros too


# ---------------------------------------------------------------------------
# marching cubes on the reference's own test scenes (golden counts + full oracle parity)
# ---------------------------------------------------------------------------
GOLDEN = [  # (name, scene, min, max, n, clip, expected vertex count), SURVEY.md 8c
    ("ColoredSpheres", S.colored_spheres, [-3] * 3, [3] * 3, 32, False, 104),
    ("Sphere5", lambda: S.sphere_w(1.0), [-1.5] * 3, [1.5] * 3, 5, False, 54),
    ("Sphere10", lambda: S.sphere_w(2.0), [-2.5] * 3, [2.5] * 3, 10, False, 312),
    ("UnclippedSphere10", lambda: S.sphere_w(2.0), [-1] * 3, [1] * 3, 10, False, 0),
    ("ClippedSphere10", lambda: S.sphere_w(2.0), [-1] * 3, [1] * 3, 10, True, 384),
    ("Box10", lambda: S.box_w(2.0), [-2.5] * 3, [2.5] * 3, 10, False, 384),
    ("Cylinder50", lambda: S.cylinder(1, 3), [-1.5, -3.5, -1.5], [1.5, 3.5, 1.5], 50, False, 7456),
    ("Sphere128", lambda: S.sphere_w(3.0), [-3.1] * 3, [3.1] * 3, 128, False, 72240),
    ("CreateMeshSphere", lambda: S.sphere_w(0.5), [-1] * 3, [1] * 3, 32, True, 1248),
    ("SolidSphere", lambda: S.solid_sphere(0.5), [-1] * 3, [1] * 3, 32, True, 1248),
    ("C1_Sphere64", lambda: S.sphere_w(1.0), [-1.5] * 3, [1.5] * 3, 64, False, 8616),
]


@pytest.mark.parametrize("case", GOLDEN, ids=[g[0] for g in GOLDEN])
def test_golden_scene(gpu, case):
    name, mk, mn, mx, n, clip, expect = case
    scene, sdf = mk()
    ov, oc = O.sample(scene, mn, mx, n, n, n)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) == expect
    # two-stage path: Voxels.SampleSdf -> (ClipToBounds) -> MarchingCubes.CreateMesh
    vol = Voxels.SampleSdf(sdf, mn, mx, n, n, n)
    if clip:
        vol.ClipToBounds()
    got = []
    m = MarchingCubes.CreateMesh(vol, 0.0, 1, got.append)
    assert len(m.Vertices) == expect
    assert_mesh_equal(m, om)
    assert m.ActiveCells == len(om.cells)
    if n > 2:
        assert all(0.0 <= f <= 1.0 for f in got) and min(got) < 1e-6 and 1 - max(got) < 1e-6
    # fused path: SdfEx.ToMesh
    m2 = sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip)
    assert_mesh_equal(m2, om)


@pytest.mark.parametrize("name", ["readme_repeat_xy", "union8", "repeat_xz_box", "sdf_with_color", "repeat_xy_plain"])
def test_coloured_scenes(gpu, name):
    scene, sdf = S.CATALOGUE[name]()
    mn, mx, n = [-2.8125] * 3, [2.8125] * 3, (48, 40, 52)
    ov, oc = O.sample(scene, mn, mx, *n)
    O.clip_to_bounds(ov, mn, mx)
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) > 0
    m = sdf.ToMesh(mn, mx, *n)
    assert_mesh_equal(m, om)


# ---------------------------------------------------------------------------
# host-array path and the ambiguous / centre-vertex tilings (random volumes)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(8))
def test_random_volume_all_cases(gpu, seed):
    """17^3 uniform(-1,1): hits all 14 cases, the face/interior tests, centre vertices
    (oracle-only pin: no reference test reaches these)."""
    rng = np.random.default_rng(seed)
    shape = (17, 17, 17) if seed < 4 else (9 + seed, 21, 14)
    v = rng.uniform(-1, 1, shape).astype(np.float32)
    c = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    mn, mx = [-1, -2, -3], [1.5, 2.5, 3.5]
    om = O.march(v, c, mn, mx)
    assert len(np.unique(om.cells[:, 1])) > 200  # most of the 254 active sign words occur
    m = MarchingCubes.CreateMesh(Voxels(v, c, mn, mx))
    assert_mesh_equal(m, om)
    assert m.ImpossibleCase13Cells == om.impossible13


@pytest.mark.parametrize("iso,step", [(0.25, 1), (-0.3, 1), (0.0, 2), (0.1, 3), (0.0, 5)])
def test_iso_and_step(gpu, iso, step):
    rng = np.random.default_rng(100 + step)
    v = rng.uniform(-1, 1, (26, 19, 23)).astype(np.float32)
    c = rng.uniform(0, 1, (26, 19, 23, 3)).astype(np.float32)
    mn, mx = [-1] * 3, [1] * 3
    om = O.march(v, c, mn, mx, iso, step)
    m = MarchingCubes.CreateMesh(Voxels(v, c, mn, mx), iso, step)
    assert_mesh_equal(m, om)
    scene, sdf = S.sphere_w(1.0)
    ov, oc = O.sample(scene, [-1.5] * 3, [1.5] * 3, 40, 40, 40)
    om = O.march(ov, oc, [-1.5] * 3, [1.5] * 3, iso, step)
    m = sdf.ToMesh([-1.5] * 3, [1.5] * 3, 40, 40, 40, clipToBounds=False, isoValue=iso, step=step)
    assert_mesh_equal(m, om)


# Corner values (v0..v7) of a case-13 cell whose face tests come out (1,0,1,0,0,0): face 1
# is degenerate (|v0*v5 - v4*v1| = 3e-8 < 1e-7 -> true), face 3 is true, faces 2,4,5,6 false;
# Luts.subconfig13[5] = -1, i.e. "Marching Cubes: Impossible case 13?" (MarchingCubes.cs:365).
DEAD13 = np.array([1e-4, -2e-4, 1.5, -1.0, -2e-4, 1e-4, -1.0, 1.5], np.float32)
CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def plant_dead_cells(v, cells, flip=()):
    for n, (x, y, z) in enumerate(cells):
        sgn = -1.0 if n in flip else 1.0   # flipped: sign word 0x5A instead of 0xA5
        for k, (dx, dy, dz) in enumerate(CORNER):
            v[x + dx, y + dy, z + dz] = sgn * DEAD13[k]
    return v


def test_impossible_case13_cells(gpu):
    """Cells that resolve to no tiling emit nothing, so vertices on their edges are created
    by LATER cells of the sweep (or never): exercises the vertex-creation rule when an
    earlier cell is dead, at the volume corner, in the interior and on the far faces."""
    assert O.resolve_tiling(DEAD13.astype(np.float64))[1:] == (-1, 0)
    rng = np.random.default_rng(7)
    v = rng.uniform(-1, 1, (14, 12, 13)).astype(np.float32)
    c = rng.uniform(0, 1, v.shape + (3,)).astype(np.float32)
    cells = [(0, 0, 0), (5, 5, 5), (12, 10, 11), (0, 6, 3), (8, 0, 9), (3, 3, 0), (9, 7, 2), (2, 9, 8)]
    plant_dead_cells(v, cells, flip=(1, 4))
    om = O.march(v, c, [-1] * 3, [1] * 3)
    assert om.impossible13 >= 6
    m = MarchingCubes.CreateMesh(Voxels(v, c, [-1] * 3, [1] * 3))
    assert m.ImpossibleCase13Cells == om.impossible13
    assert_mesh_equal(m, om)
    # a volume made only of dead cells and empty space
    w = np.full((6, 6, 6), -1.0, np.float32)
    plant_dead_cells(w, [(2, 2, 2)])
    om = O.march(w, None, [-1] * 3, [1] * 3)
    m = MarchingCubes.CreateMesh(Voxels(w, None, [-1] * 3, [1] * 3))
    assert om.impossible13 == 1
    assert_mesh_equal(m, om)


def test_speculative_sizes_too_small_are_recovered(gpu):
    """A repeat call with the same grid shape is launched speculatively with buffers sized from
    the previous mesh; when the new mesh is much larger the library must notice and redo it."""
    mn, mx, n = [-1.5] * 3, [1.5] * 3, 56
    small = Sdfs.Sphere(0.2).ToMesh(mn, mx, n, n, n, clipToBounds=False)           # sets the size hint
    assert 0 < len(small.Vertices) < 2000
    for name in ("union8", "readme_repeat_xy"):
        scene, sdf = S.CATALOGUE[name]()
        ov, oc = O.sample(scene, [-2.8125] * 3, [2.8125] * 3, n, n, n)
        O.clip_to_bounds(ov, [-2.8125] * 3, [2.8125] * 3)
        om = O.march(ov, oc, [-2.8125] * 3, [2.8125] * 3)
        assert len(om.vertices) > 4 * len(small.Vertices)
        assert_mesh_equal(sdf.ToMesh([-2.8125] * 3, [2.8125] * 3, n, n, n), om)
        small = Sdfs.Sphere(0.2).ToMesh(mn, mx, n, n, n, clipToBounds=False)       # shrink the hint again
    v = Voxels.SampleSdf(Sdfs.Sphere(0.2), mn, mx, n, n, n)                           # and the volume is intact
    ov, _ = O.sample(S.sphere_w(0.2)[0], mn, mx, n, n, n)
    assert np.array_equal(v.Values, ov)


def test_sparse_rows_volume(gpu):
    """Isolated blobs far apart: a 256-cell chunk of the sweep spans hundreds of (mostly empty)
    cell rows and several layers, so neighbour windows are cut and the slow path is used."""
    rng = np.random.default_rng(5)
    shape = (70, 66, 41)
    v = np.full(shape, -1.0, np.float32) - rng.uniform(0, 1, shape).astype(np.float32)
    c = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    for x in range(3, shape[0] - 3, 9):
        for y in range(2, shape[1] - 3, 11):
            for z in range(2, shape[2] - 3, 6):
                v[x:x + 2, y, z] = rng.uniform(0.2, 2.0, 2)
    om = O.march(v, c, [-1] * 3, [1] * 3)
    assert len(om.vertices) > 2000
    m = MarchingCubes.CreateMesh(Voxels(v, c, [-1] * 3, [1] * 3))
    assert_mesh_equal(m, om)
    # ... and with cell rows by the thousand: k_vertices stages 1152 rowstart entries per window (mc_kernels.hip), so only a
    # chunk whose records are spread over more rows than that still cuts its windows and takes the global-memory path
    shape = (19, 2000, 9)
    v = np.full(shape, -1.0, np.float32) - rng.uniform(0, 1, shape).astype(np.float32)
    c = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    for x in range(2, shape[0] - 3, 4):
        for y in range(2, shape[1] - 3, 150):
            for z in range(1, shape[2] - 2, 2):
                v[x:x + 2, y:y + 2, z] = rng.uniform(0.2, 2.0, (2, 2))
    om = O.march(v, c, [-1] * 3, [1] * 3)
    assert len(om.vertices) > 2000
    assert_mesh_equal(MarchingCubes.CreateMesh(Voxels(v, c, [-1] * 3, [1] * 3)), om)
    # and a surface with exactly one active cell per cell row (a plane x = const)
    scene, sdf = S.plane_w((1.0, 0.0, 0.0), 0.013)
    mn, mx, n = [-1] * 3, [1] * 3, (33, 90, 47)
    ov, oc = O.sample(scene, mn, mx, *n)
    om = O.march(ov, oc, mn, mx)
    assert_mesh_equal(sdf.ToMesh(mn, mx, *n, clipToBounds=False), om)


def test_many_layers_many_chunks(gpu):
    """More than MC_SCAN_BLOCKS (8192) logical blocks in the compaction and more than MC_SCAN_CHUNKS (8192) chunks of records: their
    prefixes then come from k_blockscan / k_chunkscan (one workgroup between two kernels of the chain) instead of every consumer
    workgroup summing its predecessors (mc_kernels.hip).  A tall grid of 8300 planes with a busy surface, as a host volume (first
    call: capacity too small, redone exactly; second call: speculative sizes) and as a sampled program (the volume-less default path)."""
    shape = (32, 28, 8300)
    X, Y, Z = np.meshgrid(*[np.arange(n, dtype=np.float32) for n in shape], indexing="ij")
    v = (np.sin(0.9 * X) * np.cos(0.8 * Y) + 0.6 * np.sin(0.11 * Z) - 0.1).astype(np.float32)
    rng = np.random.default_rng(11)
    c = rng.uniform(0, 1, shape + (3,)).astype(np.float32)
    om = O.march(v, c, [-1] * 3, [1] * 3)
    assert len(om.cells) > 8192 * 240
    for rep in range(2):
        assert_mesh_equal(MarchingCubes.CreateMesh(Voxels(v, c, [-1] * 3, [1] * 3)), om)
    scene, sdf = S.CATALOGUE["readme_repeat_xy"]()
    mn, mx, dims = [-2.8125, -2.8125, -0.6], [2.8125, 2.8125, 0.6], (40, 36, 8300)
    ov, oc = O.sample(scene, mn, mx, *dims)
    O.clip_to_bounds(ov, mn, mx)
    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims, clipToBounds=True), O.march(ov, oc, mn, mx))


@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 2, 2), (1, 5, 5), (5, 1, 5), (5, 5, 1), (2, 3, 65), (65, 2, 3), (3, 70, 2), (130, 3, 4)])
def test_degenerate_and_ragged_shapes(gpu, shape):
    rng = np.random.default_rng(sum(shape))
    v = rng.uniform(-1, 1, shape).astype(np.float32)
    om = O.march(v, None, [-1] * 3, [1] * 3)
    m = MarchingCubes.CreateMesh(Voxels(v, None, [-1] * 3, [1] * 3))
    assert_mesh_equal(m, om)


def test_empty_and_full_volumes(gpu):
    for fill in (1.0, -1.0, 0.0):
        v = np.full((12, 12, 12), fill, np.float32)
        m = MarchingCubes.CreateMesh(Voxels(v, None, [-1] * 3, [1] * 3))
        assert len(m.Vertices) == 0 and len(m.Triangles) == 0


def test_nan_and_inf_voxels(gpu):
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, (12, 13, 16)).astype(np.float32)
    v[3, 4, 5] = np.inf
    v[7, 7, 7] = -np.inf
    om = O.march(v, None, [-1] * 3, [1] * 3)
    m = MarchingCubes.CreateMesh(Voxels(v, None, [-1] * 3, [1] * 3))
    assert_mesh_equal(m, om, exact=False)
    assert np.array_equal(m.Triangles, om.triangles)


# ---------------------------------------------------------------------------
# full-size properties (BASELINE configs C2 / C3): size-independent invariants
# ---------------------------------------------------------------------------
def _sign_change_edges(v, iso=0.0):
    s = v > iso
    return int((s[1:] != s[:-1]).sum() + (s[:, 1:] != s[:, :-1]).sum() + (s[:, :, 1:] != s[:, :, :-1]).sum())


def test_c3s_sphere_512(gpu):
    """The headline workload itself (BASELINE C3s: 512^3 `Sdfs.Sphere(1)`, bounds -1.5..1.5, no clip):
    the mesh bench.py times, compared array by array with the oracle's (CPU marching cubes: ~15 s)."""
    scene, sdf = S.sphere_w(1.0)
    mn, mx, n = [-1.5] * 3, [1.5] * 3, 512
    m = sdf.ToMesh(mn, mx, n, n, n, clipToBounds=False)       # the fused path bench.py drives (sdfk_sample_march)
    assert len(m.Vertices) == 549144 and len(m.Triangles) == 3 * 1098284   # SURVEY.md 8d / BASELINE.md C3s
    ov, oc = O.sample(scene, mn, mx, n, n, n)
    om = O.march(ov, oc, mn, mx)
    assert_mesh_equal(m, om)
    # ... and the two-call form through a resident volume (Voxels.SampleSdf -> CreateMesh)
    vol = Voxels.SampleSdf(sdf, mn, mx, n, n, n)
    assert_mesh_equal(vol.ToMesh(), om)
    assert np.array_equal(vol.Values, ov)


def test_c2_sphere_256(gpu):
    scene, sdf = S.sphere_w(1.0)
    vol = Voxels.SampleSdf(sdf, [-1.5] * 3, [1.5] * 3, 256, 256, 256)
    m = vol.ToMesh()
    assert len(m.Vertices) == 137232  # sign-changing edges at 256^3
    assert len(m.Vertices) == _sign_change_edges(vol.Values)
    t = m.Triangles.reshape(-1, 3)
    # closed 2-manifold: every undirected edge is shared by exactly two triangles, Euler = 2
    e = np.sort(np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert np.all(cnt == 2)
    assert len(m.Vertices) - len(cnt) + len(t) == 2
    # vertices numbered in order of first reference (serial sweep property)
    first = np.full(len(m.Vertices), len(m.Triangles), np.int64)
    np.minimum.at(first, m.Triangles, np.arange(len(m.Triangles)))
    assert np.all(np.diff(first) > 0)
    # all vertices within half a cell of the unit sphere, normals point outwards
    r = np.linalg.norm(m.Vertices.astype(np.float64), axis=1)
    assert np.abs(r - 1.0).max() < 3.0 / 256
    assert np.all(np.einsum("ij,ij->i", m.Normals, m.Vertices) > 0.9 * r)
    assert np.abs(np.linalg.norm(m.Normals.astype(np.float64), axis=1) - 1).max() < 1e-6
    assert np.abs(m.Center).max() <= 1e-6
    # oracle parity at this size too (about 1 s of CPU)
    ov, oc = O.sample(scene, [-1.5] * 3, [1.5] * 3, 256, 256, 256)
    assert np.array_equal(vol.Values, ov)
    om = O.march(ov, oc, [-1.5] * 3, [1.5] * 3)
    assert_mesh_equal(m, om)


def test_c3_repeat_xy_512_properties(gpu):
    scene, sdf = S.readme_repeat_xy()
    mn, mx = [-2.8125] * 3, [2.8125] * 3
    vol = sdf.ToVoxels(mn, mx, 512, 512, 512)
    m = vol.ToMesh()
    vals = vol.Values
    assert len(m.Vertices) == _sign_change_edges(vals)
    t = m.Triangles
    assert t.min() == 0 and t.max() == len(m.Vertices) - 1
    first = np.full(len(m.Vertices), len(t), np.int64)
    np.minimum.at(first, t, np.arange(len(t)))
    assert np.all(np.diff(first) > 0)
    # spot-check the volume against the oracle on a thin x-slab (full 512^3 on CPU is slow)
    ov, oc = O.sample(scene, mn, mx, 512, 512, 512, threads=0)
    O.clip_to_bounds(ov, mn, mx)
    assert np.array_equal(vals, ov)
    assert np.array_equal(vol.Colors, oc)
    # colour blend stays inside the palette range of the scene
    assert m.Colors.min() >= 0.9 - 3.0 / 6 - 1e-6 and m.Colors.max() <= 0.9 + 1e-6
    # ... and the whole mesh against the reference's serial sweep at FULL size (MarchingCubes.cs:39-92 on the README scene,
    # README.md:24-30): vertices, colours, normals, triangles, AABB -- about 30 s of one CPU core
    del vals
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) > 900_000
    assert_mesh_equal(m, om)
    assert m.ActiveCells == len(om.cells)


def test_c4_union8_1024_windows_equal_oracle(gpu):
    """BASELINE config C4 at its full size, 1024^3 CSG union of 8 primitives, against the reference's serial sweep on z
    WINDOWS (the whole sweep of 10^9 cells does not fit a test; the window form of the oracle is pinned against the
    whole-volume oracle by tests/test_oracle_window.py): the slab the 8-GPU run gives rank 3 right at its seam with rank 2,
    the layers through the centre plane of the upper primitives, and the two clip faces of the grid.  For each window the
    product meshes exactly those layers of a slab volume (sdfk_march_slab, slab-local vertex ids) and every array must be
    the oracle's, bit for bit -- including the negative ids that point across the seam."""
    from tests import slab_worker as W
    scene, sdf = S.CATALOGUE["union8"]()
    n = 1024
    mn, mx = [-2.0] * 3, [2.0] * 3        # bounds -2..2, clipToBounds
    L = N.lib()
    seam = W.slab_layers(n - 1, 8, 3)[0]
    # (the primitives span z indices 102..410 and 614..922: the last two windows cut their lowest tips and hold nothing at all)
    windows = [(seam, seam + 12), (seam - 6, seam + 6), (762, 774), (100, 112), (914, 926), (0, 10)]
    prog = sdf.program()
    total_v = 0
    for lb, le in windows:
        z0, nzl = W.slab_planes(lb, le, n)
        ov, oc = O.sample_window(scene, mn, mx, n, n, n, z0, nzl, clip=True)
        wm = O.march_window(ov, oc, z0, n, mn, mx)
        want = O.window_part(wm, n, n, z0, lb, le)
        vol, mesh = C.c_void_p(), C.c_void_p()
        N.check(L.sdfk_volume_create_slab(n, n, n, N.f3(mn), N.f3(mx), z0, nzl, 1, C.byref(vol)))
        try:
            for rep in range(2):      # exact path first, then the speculative one (hints of this slab shape)
                N.check(L.sdfk_sample(prog, vol, 1))
                if rep == 0:          # the slab's voxels are the oracle's
                    gv, gc = np.empty((n, n, nzl), np.float32), np.empty((n, n, nzl, 3), np.float32)
                    N.check(L.sdfk_volume_download(vol, gv.ctypes.data, gc.ctypes.data))
                    assert np.array_equal(gv, ov) and np.array_equal(gc, oc)
                    del gv, gc
                N.check(L.sdfk_march_slab(vol, C.c_float(0.0), lb, le, 0, C.byref(mesh)))
                nv, ni = C.c_int64(), C.c_int64()
                N.check(L.sdfk_mesh_counts(mesh, C.byref(nv), C.byref(ni)))
                assert (nv.value, ni.value) == (len(want[0]), len(want[3])), (lb, le, nv.value, ni.value)
                got = [np.empty((nv.value, 3), np.float32) for _ in range(3)] + [np.empty(ni.value, np.int32)]
                N.check(L.sdfk_mesh_copy(mesh, *[a.ctypes.data for a in got]))
                L.sdfk_mesh_free(mesh)
                for a, b, what in zip(got, want, ("vertices", "colours", "normals", "triangles")):
                    assert np.array_equal(a, b, equal_nan=True), f"window [{lb}, {le}) pass {rep}: {what} differ"
                if lb > 0 and len(got[3]) and lb not in (100,):
                    assert got[3].min() < 0          # triangles of the first layer reach across the seam
        finally:
            L.sdfk_volume_free(vol)
        total_v += len(want[0])
        del ov, oc, wm
    assert total_v > 100_000


def _exclusive_prefix(counts):
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += int(c)
    return out, acc


# ---------------------------------------------------------------------------
# Z-slab sharding: slabs meshed one after another on this GPU must concatenate to exactly
# the single-volume mesh (what the 8-GPU path relies on; sdfkit_amd/dist.py)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3, 5, 8])
@pytest.mark.parametrize("scene_name,dims", [("readme_repeat_xy", (40, 36, 44)), ("union8", (33, 30, 26))])
def test_slab_concatenation_equals_whole(gpu, world, scene_name, dims):
    from sdfkit_amd import dist as D
    from tests import slab_worker as W
    scene, sdf = S.CATALOGUE[scene_name]()
    mn, mx = [-2.8125] * 3, [2.8125] * 3
    whole = sdf.ToMesh(mn, mx, *dims)
    workers = [W.GpuSlabWorker(sdf, mn, mx, *dims, r, world, True, 0.0) for r in range(world)]
    try:
        counts = [w.begin() for w in workers]
        bases, total = _exclusive_prefix([c[0] for c in counts])
        assert total == len(whole.Vertices)
        V, Cc, Nn, T, mins, maxs = [], [], [], [], [], []
        for w, (nv, ni), base in zip(workers, counts, bases):
            h = w.finish(base)
            v = np.empty((nv, 3), np.float32); c = np.empty((nv, 3), np.float32)
            n = np.empty((nv, 3), np.float32); t = np.empty((ni,), np.int32)
            N.check(N.lib().sdfk_mesh_copy(h, v.ctypes.data, c.ctypes.data, n.ctypes.data, t.ctypes.data))
            V.append(v); Cc.append(c); Nn.append(n); T.append(t)
    finally:
        for w in workers:
            w.close()
    assert np.array_equal(np.concatenate(T), whole.Triangles)
    assert np.array_equal(np.concatenate(V), whole.Vertices)
    assert np.array_equal(np.concatenate(Cc), whole.Colors)
    assert np.array_equal(np.concatenate(Nn), whole.Normals, equal_nan=True)


def test_slab_with_dead_cells_at_the_seam(gpu):
    """'Impossible 13' cells right below / at a slab seam: the lower slab's last layer and the
    upper slab's recount of it must agree on who creates the seam vertices."""
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, (10, 9, 16)).astype(np.float32)
    plant_dead_cells(v, [(2, 2, 6), (5, 4, 7), (7, 1, 5), (1, 6, 8), (4, 4, 3)])
    mn, mx = [-1] * 3, [1] * 3
    om = O.march(v, None, mn, mx)
    assert om.impossible13 >= 4
    L = N.lib()
    world = 2
    from sdfkit_amd import dist as D
    from tests import slab_worker as W
    got_t, got_v, base = [], [], 0
    for r in range(world):
        lb, le = W.slab_layers(15, world, r)   # seam at layer 8 (rank 0: [0,8), rank 1: [8,15))
        z0, nzl = W.slab_planes(lb, le, 16)
        vol = C.c_void_p()
        N.check(L.sdfk_volume_create_slab(10, 9, 16, N.f3(mn), N.f3(mx), z0, nzl, 0, C.byref(vol)))
        sub = np.ascontiguousarray(v[:, :, z0:z0 + nzl])
        N.check(L.sdfk_volume_upload(vol, sub.ctypes.data, None))
        job, nv, ni = C.c_void_p(), C.c_int64(), C.c_int64()
        N.check(L.sdfk_march_begin(vol, C.c_float(0.0), lb, le, C.byref(job), C.byref(nv), C.byref(ni)))
        m = C.c_void_p()
        N.check(L.sdfk_march_finish(job, base, C.byref(m)))
        vv = np.empty((nv.value, 3), np.float32); tt = np.empty((ni.value,), np.int32)
        N.check(L.sdfk_mesh_copy(m, vv.ctypes.data, None, None, tt.ctypes.data))
        got_v.append(vv); got_t.append(tt); base += nv.value
        L.sdfk_mesh_free(m); L.sdfk_march_job_free(job); L.sdfk_volume_free(vol)
    assert np.array_equal(np.concatenate(got_t), om.triangles)
    assert np.array_equal(np.concatenate(got_v), om.vertices)


@pytest.mark.parametrize("world", [2, 4])
def test_slab_one_call_pack_and_rebase(gpu, world):
    """The steady-state sharded form (dist.SlabSession minus the collective): one-call slab
    marches with slab-local ids, self-describing payloads placed side by side as an all-gather
    would, one rebase launch -> identical to the single-volume mesh."""
    import torch
    from sdfkit_amd import dist as D
    from tests import slab_worker as W
    scene, sdf = S.readme_repeat_xy()
    mn, mx, dims = [-2.8125] * 3, [2.8125] * 3, (44, 40, 48)
    whole = sdf.ToMesh(mn, mx, *dims)
    N.bind_torch_stream()
    try:
        for _ in range(2):   # second round runs on the speculative (hinted) path
            workers = [W.GpuSlabWorker(sdf, mn, mx, *dims, r, world, True, 0.0) for r in range(world)]
            counts = [w.run_local() for w in workers]
            stride = max(D.SLAB_HEADER_BYTES + 36 * a + 4 * b for a, b in counts) + 512
            g = torch.zeros((world, stride), dtype=torch.uint8, device="cuda")
            for r, w in enumerate(workers):
                assert w.pack_self_describing(g[r]) <= stride
            N.check(N.lib().sdfk_slabs_rebase(C.c_void_p(g.data_ptr()), world, stride))
            torch.cuda.synchronize()
            V, Cc, Nn, T, bmin, bmax = D.unpack_self_describing(g.cpu().numpy())
            for w in workers:
                w.close()
            assert np.array_equal(T, whole.Triangles)
            assert np.array_equal(V, whole.Vertices) and np.array_equal(Cc, whole.Colors)
            assert np.array_equal(Nn, whole.Normals, equal_nan=True)
            assert np.array_equal(bmin, whole.Min) and np.array_equal(bmax, whole.Max)
    finally:
        torch.cuda.synchronize()
        N.check(N.lib().sdfk_set_stream(None))
        torch.cuda.set_stream(torch.cuda.default_stream())


# ---------------------------------------------------------------------------
# random compositions of the SDF catalogue: lowering + JIT against the oracle's interpreter
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(16))
def test_random_scene_compositions(gpu, seed):
    scene, sdf = S.random_scene(seed, depth=3 + seed % 2)
    mn, mx, dims = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5], (28, 24, 32)
    ov, oc = O.sample(scene, mn, mx, *dims)
    O.clip_to_bounds(ov, mn, mx)
    v = sdf.ToVoxels(mn, mx, *dims)
    assert np.array_equal(v.Values, ov, equal_nan=True), f"seed {seed}: values differ"
    assert np.array_equal(v.Colors, oc, equal_nan=True), f"seed {seed}: colours differ"
    om = O.march(ov, oc, mn, mx)
    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims), om)
    assert_mesh_equal(MarchingCubes.CreateMesh(v), om)


def test_ieee_min_max_semantics(gpu):
    """Math.Max / Math.Min (SDFK_OP_MAX_IEEE / MIN_IEEE): -0 < +0 and NaN propagates from either side."""
    from sdfkit_amd import MathF, Sdf, Vec4
    def run(fn):
        v = Sdf(lambda p: Vec4.of((0.0, 0.0, 0.0), fn(p)), False).ToVoxels([-1] * 3, [1] * 3, 8, 4, 4, clipToBounds=False)
        return v.Values[:, 0, 0]
    xs_negative = np.arange(8) < 4                      # sample x < 0 for the first half of the row
    z = lambda p: p.x * 0.0                             # -0 where x < 0, +0 elsewhere
    mx, mn = run(lambda p: MathF.Max(z(p), -z(p))), run(lambda p: MathF.Min(z(p), -z(p)))
    assert np.all(mx == 0) and not np.signbit(mx).any()                 # max(-0, +0) = max(+0, -0) = +0
    assert np.all(mn == 0) and np.signbit(mn).all()                     # min = -0
    assert np.array_equal(np.signbit(run(z)), xs_negative)              # (the inputs really were -0 / +0)
    nan = lambda p: z(p) / z(p)
    for f in (lambda p: MathF.Max(nan(p), p.x), lambda p: MathF.Max(p.x, nan(p)),
              lambda p: MathF.Min(nan(p), p.x), lambda p: MathF.Min(p.x, nan(p))):
        assert np.isnan(run(f)).all()
    assert np.array_equal(run(lambda p: MathF.Max(p.x, 0.25)), np.maximum(run(lambda p: p.x), np.float32(0.25)))


# ---------------------------------------------------------------------------
# the JIT code generator against a numpy interpreter of the program IR (second oracle)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(24))
def test_random_programs_match_ir_interpreter(gpu, seed):
    """Random op DAGs over all 17 opcodes (NaN, inf, signed zeros and denormals flow through):
    sampled values and colours must equal the numpy float32 interpretation op for op."""
    from oracle import ir_interp as I
    ops, out = I.random_program(seed)
    L = N.lib()
    arr = (N.Op * len(ops))()
    for i, (op, a, b, c, d, imm) in enumerate(ops):
        arr[i].opcode, arr[i].a, arr[i].b, arr[i].c, arr[i].d, arr[i].imm = op, a, b, c, d, imm
    prog = C.c_void_p()
    N.check(L.sdfk_program_create(arr, len(ops), (C.c_int32 * 4)(*out), 1, C.byref(prog)))
    try:
        for dims in ((16, 12, 20), (9, 7, 18)):        # nz % 4 == 0 and the any-nz instantiation
            mn, mx = [-2.0, -1.0, -0.5], [1.0, 2.0, 3.5]
            ev, ec = I.sample(ops, out, True, mn, mx, *dims)
            vol = C.c_void_p()
            N.check(L.sdfk_volume_create(*dims, N.f3(mn), N.f3(mx), 1, C.byref(vol)))
            N.check(L.sdfk_sample(prog, vol, 0))
            v = np.empty(dims, np.float32)
            c = np.empty(dims + (3,), np.float32)
            N.check(L.sdfk_volume_download(vol, v.ctypes.data, c.ctypes.data))
            L.sdfk_volume_free(vol)
            assert np.array_equal(v, ev, equal_nan=True), f"seed {seed} {dims}: values"
            assert np.array_equal(c, ec, equal_nan=True), f"seed {seed} {dims}: colours"
            finite = np.isfinite(ev)
            assert np.array_equal(np.signbit(v[finite]), np.signbit(ev[finite]))       # -0 / +0 too
    finally:
        L.sdfk_program_destroy(prog)


# ---------------------------------------------------------------------------
# isolated cells covering every sign word with wildly different corner magnitudes: reaches the rare
# face-test / interior-test sub-tilings that uniform random volumes seldom produce
# ---------------------------------------------------------------------------
_CORNERS = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def _cell_lattice(seed, reps=40):
    rng = np.random.default_rng(5000 + seed)
    n = 256 * reps
    side = int(np.ceil(n ** (1 / 3)))
    v = np.full((3 * side + 1,) * 3, -1.0, np.float32)
    w = np.arange(n) % 256
    mag = (10.0 ** rng.uniform(-3, 1.5, (n, 8))).astype(np.float32)
    vals = mag * np.where((w[:, None] >> np.arange(8)) & 1, 1.0, -1.0).astype(np.float32)
    ii, jj, kk = np.unravel_index(np.arange(n), (side, side, side))
    for c, (dx, dy, dz) in enumerate(_CORNERS):
        v[3 * ii + 1 + dx, 3 * jj + 1 + dy, 3 * kk + 1 + dz] = vals[:, c]
    return v


@pytest.mark.parametrize("seed", range(5))
def test_cell_lattice_rare_subtilings(gpu, seed):
    v = _cell_lattice(seed)
    c = np.random.default_rng(seed).uniform(0, 1, v.shape + (3,)).astype(np.float32)
    mn, mx = [-1, -2, -3], [1.5, 2.5, 3.5]
    om = O.march(v, c, mn, mx)
    assert len({tuple(r) for r in om.cells[:, 2:4].tolist()}) > 560      # distinct tiling rows in ONE volume
    m = MarchingCubes.CreateMesh(Voxels(v, c, mn, mx))
    assert_mesh_equal(m, om)
    assert m.ActiveCells == len(om.cells)


@pytest.mark.parametrize("name,dims", [("plane_w", (70001, 3, 5)), ("sphere_w", (3, 66001, 4)), ("readme_repeat_xy", (66000, 4, 3)),
                                      ("union8", (5, 4, 70003))])
def test_extents_beyond_16_bits(gpu, name, dims):
    """The reference only caps nx * ny * nz (int32 linear index, Voxels.cs:82): one extent may well exceed 65535.  Cell
    coordinates are packed with a variable bit split then, and extents that do not fit a 16-bit grid dimension are
    folded into another one.  Sampling (values, colours) and the mesh, fused and two-stage, against the oracle."""
    scene, sdf = S.CATALOGUE[name]()
    mn, mx = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]
    ov, oc = O.sample(scene, mn, mx, *dims)
    vol = Voxels.SampleSdf(sdf, mn, mx, *dims)
    assert np.array_equal(vol.Values, ov) and np.array_equal(vol.Colors, oc)
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) > 1000
    assert_mesh_equal(vol.ToMesh(), om)
    O.clip_to_bounds(ov, mn, mx)
    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims), O.march(ov, oc, mn, mx))
    # an uploaded volume takes the load-based sign-bit and corner kernels
    assert_mesh_equal(MarchingCubes.CreateMesh(Voxels(ov, oc, mn, mx)), O.march(ov, oc, mn, mx))


@pytest.mark.parametrize("dims", [(12, 10, 256), (9, 14, 38), (8, 8, 513)])
def test_w_only_program_zeroes_the_colors_of_a_colored_volume(gpu, dims):
    """Voxels.cs:88-92,117-119: a delegate that only assigns .W scatters the zero-initialised scratch colours -- a
    volume that held colours before holds (0,0,0) everywhere afterwards (the sampler of such a program has no colour
    staging buffer; it stores the zeros directly)."""
    mn, mx = [-1.5] * 3, [1.5] * 3
    rng = np.random.default_rng(5)
    v = Voxels(rng.standard_normal(dims).astype(np.float32), rng.random(dims + (3,), dtype=np.float32) + 0.25, mn, mx)
    v._sync_to_device()
    assert v._has_colors
    scene, sdf = S.sphere_w(1.0)
    v.SampleSdf(sdf)
    ov, _ = O.sample(scene, mn, mx, *dims)
    np.testing.assert_array_equal(v.Values, ov)
    assert v._has_colors and not v.Colors.any()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_special_values_in_host_volumes(gpu, seed):
    """Garbage in, the reference's garbage out: uploaded volumes with NaN, +-inf, signed zeros, denormals and huge
    magnitudes sprinkled in (sign test `v - iso > 0` is false for NaN, weights 1 / (1e-7 + |v|) become 0 or NaN,
    positions and normals go NaN / inf) -- topology, vertex count and every finite or non-finite output bit must
    still be the oracle's."""
    rng = np.random.default_rng(100 + seed)
    dims = (19, 17, 22)
    v = rng.uniform(-1, 1, dims).astype(np.float32)
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 3e38, -3e38, 1e-30], np.float32)
    idx = rng.integers(0, v.size, v.size // 40)
    v.ravel()[idx] = specials[rng.integers(0, len(specials), len(idx))]
    c = rng.uniform(0, 1, dims + (3,)).astype(np.float32)
    mn, mx = [-1.0, -1.25, -1.5], [1.0, 1.25, 1.5]
    for iso in (0.0, 0.125, -0.0):   # (+0.0 takes k_vertices<ISO0>, which leaves `- iso` out; -0.0 must not: -0.0 - -0.0 = +0.0)
        om = O.march(v, c, mn, mx, iso=iso)
        m = MarchingCubes.CreateMesh(Voxels(v.copy(), c.copy(), mn, mx), iso)
        assert len(m.Vertices) == len(om.vertices) and np.array_equal(m.Triangles, om.triangles)
        for a, b in ((m.Vertices, om.vertices), (m.Colors, om.colors), (m.Normals, om.normals)):
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isinf(a), np.isinf(b))
            assert np.array_equal(a, b, equal_nan=True)
