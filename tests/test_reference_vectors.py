"""The oracle against vectors produced by EXECUTING the reference's own source.

tests/golden/reference_meshes.npz was written by tools/gen_reference_vectors.py in the build container: MarchingCubes.CreateMesh
(MarchingCubes.cs:39-92 with TheBigSwitch, TestFace, TestInternal) and all of Cell.cs, parsed where they lie under
/root/reference and run by tools/cs_subset.py (an interpreter of the C# subset the two files are written in) on seeded volumes.
Recorded: what the Mesh constructor receives at MarchingCubes.cs:84 -- vertices in voxel-index units, colours, negative normals,
faces -- and the number of "Impossible case 13?" messages.  This pins what the reference's NUnit tests do not:
positions, colours, normals, index contents, every ambiguous and centre-vertex tiling, step > 1, iso != 0 -- bit for bit.
(The interpreter's arithmetic is IEEE float32 / float64 with C#'s promotions; Vector3.Normalize is the BCL restatement the
oracle documents.  Mesh.Transform, BCL matrix arithmetic, is outside the vectors.)"""
import os

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = np.load(os.path.join(ROOT, "tests", "golden", "reference_meshes.npz"))
NAMES = [str(n) for n in REF["names"]]


def case(name):
    g = lambda k: REF[f"{name}/{k}"]
    iso, step = g("iso_step")
    return g("values"), g("colors"), float(iso), int(step), g("vertices"), g("out_colors"), g("normals"), g("faces"), int(g("console_lines")[0])


def identity_bounds(shape):
    """min / max for which CreateMesh's T(-(n-1)/2) S(size/(n-1)) T(center) is the identity: positions stay in voxel units."""
    return [0.0, 0.0, 0.0], [float(n - 1) for n in shape]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_the_executed_reference(name):
    values, colors, iso, step, rv, rc, rn, rf, lines = case(name)
    mn, mx = identity_bounds(values.shape)
    m = O.march(values, colors, mn, mx, iso=iso, step=step)
    assert len(m.grid_vertices) == len(rv) and len(m.triangles) == len(rf)
    assert np.array_equal(m.triangles, rf)
    assert np.array_equal(m.grid_vertices, rv)
    assert np.array_equal(m.colors, rc)
    assert np.array_equal(m.grid_normals, rn, equal_nan=True)
    assert m.impossible13 == lines
    assert np.array_equal(m.vertices, rv)   # (identity transform: x * 1 + 0)


def test_the_vectors_reach_the_hard_parts():
    """How much of the algorithm the vectors exercise, counted on the oracle's own record of the cells it resolved for these
    inputs: distinct (tiling row, triangle count) pairs over all 33 tiling tables -- the reference's own tests reach none of the
    ambiguous ones --, dead case-13 cells, vertices."""
    rows = set()
    for name in NAMES:
        values, colors, iso, step = case(name)[:4]
        mn, mx = identity_bounds(values.shape)
        for _cell, _index, lutoff, nt in O.march(values, colors, mn, mx, iso=iso, step=step).cells:
            rows.add((int(lutoff), int(nt)))
    assert len(rows) >= 570, len(rows)
    assert sum(case(n)[8] for n in NAMES) == 4
    assert sum(len(case(n)[4]) for n in NAMES) == 19568


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_path_equals_the_executed_reference(gpu, name):
    from sdfkit_amd import MarchingCubes, Voxels
    values, colors, iso, step, rv, rc, rn, rf, lines = case(name)
    mn, mx = identity_bounds(values.shape)
    m = MarchingCubes.CreateMesh(Voxels(values, colors, mn, mx), iso, step)
    assert np.array_equal(m.Triangles, rf)
    assert np.array_equal(m.Vertices, rv)
    assert np.array_equal(m.Colors, rc)
    # Mesh.Transform with the identity still re-normalises (Mesh.cs:61): n / |n| of the reference's negative normal, float32
    with np.errstate(all="ignore"):
        ln = np.sqrt((rn[:, 0] * rn[:, 0] + rn[:, 1] * rn[:, 1]) + rn[:, 2] * rn[:, 2]).astype(np.float32)
        want = (rn / ln[:, None]).astype(np.float32)
    assert np.array_equal(m.Normals, want, equal_nan=True)
    assert m.ImpossibleCase13Cells == lines


@pytest.mark.skipif(not os.path.exists("/root/reference/SdfKit/Cell.cs"), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("name", ["random0", "random4", "dead13", "sphere12_step3"])
def test_vectors_regenerate_from_the_reference_source(name):
    """In the build container: parse MarchingCubes.cs / Cell.cs again, execute CreateMesh on the stored inputs with
    tools/cs_subset.py and get the stored outputs -- the fixture is what the generator says it is."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_reference_vectors as G
    classes, luts = G.load()
    values, colors, iso, step, rv, rc, rn, rf, lines = case(name)
    v, c, n, f, console = G.create_mesh(classes, luts, values, colors, iso, step)
    assert np.array_equal(v, rv) and np.array_equal(c, rc) and np.array_equal(n, rn, equal_nan=True) and np.array_equal(f, rf)
    assert console == lines


# ---- the SDF catalogue, executed (tools/gen_reference_sdf_vectors.py: SdfFuncs / SdfFuncEx of Sdf.cs:217-341) ----------
SDF = np.load(os.path.join(ROOT, "tests", "golden", "reference_sdf_points.npz"))
SDF_NAMES = [str(n) for n in SDF["names"]]


def oracle_scene(descr):
    sc = O.Scene()

    def build(d):
        kind = d[0]
        if kind == "exprs_solid_sphere":
            return sc.f_sphere(d[1])
        if kind == "exprs_union":
            return sc.f_union(build(d[1]), build(d[2]))
        if kind == "exprs_color":
            return sc.f_with_color(build(d[1]), d[2], d[3], d[4])
        if kind == "exprs_translate":
            return sc.f_translate(build(d[1]), d[2], d[3], d[4])
        if kind == "exprs_repeat_x":
            return sc.f_repeat_x(build(d[1]), d[2])
        if kind == "exprs_repeat_y":
            return sc.f_repeat_y(build(d[1]), d[2])
        if kind == "exprs_repeat_xy":
            return sc.f_repeat_xy(build(d[1]), d[2], d[3])
        if kind == "exprs_repeat_xy_idx":
            return sc.f_repeat_xy_idx(build(d[1]), d[2], d[3])
        if kind == "exprs_repeat_xz_idx":
            return sc.f_repeat_xz_idx(build(d[1]), d[2], d[3])
        if kind == "exprs_cylinder":
            return sc.f_cylinder(d[1], d[2], tuple(d[3]) if len(d) > 3 else (1, 1, 1))
        if kind == "exprs_sphere":
            return sc.f_sphere(d[1], tuple(d[2]) if len(d) > 2 else (1, 1, 1))
        if kind == "exprs_box":
            return sc.f_box(d[1], d[2], d[3])
        if kind == "sdfs_sphere":
            return sc.sphere_w(d[1])
        if kind == "sdfs_box":
            return sc.box_w(d[1])
        if kind == "sdfs_plane":
            return sc.plane_w(d[1], d[2], d[3], d[4])
        if kind == "sphere":
            return sc.f_sphere(d[1])
        if kind == "box":
            return sc.f_box(d[1], d[2], d[3])
        if kind == "with_color":
            return sc.f_with_color(build(d[1]), d[2], d[3], d[4])
        if kind == "translate":
            return sc.f_translate(build(d[1]), d[2], d[3], d[4])
        if kind == "union":
            return sc.f_union(build(d[1]), build(d[2]))
        if kind == "repeat_xy_idx":
            return sc.f_repeat_xy_idx(build(d[1]), d[2], d[3])
        if kind == "repeat_xz_idx":
            return sc.f_repeat_xz_idx(build(d[1]), d[2], d[3])
        raise KeyError(kind)

    sc.root = build(descr)
    return sc


@pytest.mark.parametrize("name", SDF_NAMES)
def test_oracle_sdf_catalogue_equals_the_executed_reference(name):
    """orc_eval against what the reference's own lambdas return (SdfFuncs.Sphere / Box / Union, SdfFuncEx.Translate / WithColor /
    RepeatXY / RepeatXZ with the README's colour lambda), at 611 points per scene incl. period boundaries, signed zeros and
    denormals: colour and distance, bit for bit."""
    import json
    descr = json.loads(str(SDF["scenes_json"]))[name]
    sc = oracle_scene(descr)
    pts, want = SDF[f"{name}/points"], SDF[f"{name}/rgbw"]
    got = np.stack([O.eval_point(sc, p) for p in pts])
    same = got.view(np.uint32) == want.view(np.uint32)
    assert same.all(), (np.argwhere(~same)[:5], got[~same.all(axis=1)][:3], want[~same.all(axis=1)][:3])


def mirror_sdf(descr):
    """The same scene through the product's host mirror (sdfkit_amd.api: what becomes the sdfk_op list the GPU compiles)."""
    from sdfkit_amd import SdfFuncs
    from tests.scenes import _readme_color
    kind = descr[0]
    if kind.startswith("sdfs_"):   # the batched catalogue: already an Sdf (ToSdf below is then the identity)
        from sdfkit_amd import Sdfs
        sdf = {"sdfs_sphere": lambda: Sdfs.Sphere(descr[1]), "sdfs_box": lambda: Sdfs.Box(descr[1]),
               "sdfs_plane": lambda: Sdfs.Plane((descr[1], descr[2], descr[3]), descr[4])}[kind]()
        sdf.ToSdf = lambda: sdf
        return sdf
    if kind in ("exprs_union", "exprs_color", "exprs_translate", "exprs_repeat_x", "exprs_repeat_y", "exprs_repeat_xy", "exprs_repeat_xy_idx", "exprs_repeat_xz_idx"):
        from sdfkit_amd import SdfExprs, Vec3
        from tests.scenes import _readme_color
        if kind == "exprs_union":
            return SdfExprs.Union(mirror_sdf(descr[1]), mirror_sdf(descr[2]))
        child = mirror_sdf(descr[1])
        if kind == "exprs_color":
            return child.Color(descr[2], descr[3], descr[4])
        if kind == "exprs_translate":
            off = (descr[2], descr[3], descr[4])
            return child.ModifyInput(lambda p: p - Vec3.of(p.x.b, off))
        if kind == "exprs_repeat_x":
            return child.RepeatX(descr[2])
        if kind == "exprs_repeat_y":
            return child.RepeatY(descr[2])
        if kind == "exprs_repeat_xy":
            return child.RepeatXY(descr[2], descr[3])
        if kind == "exprs_repeat_xy_idx":
            return child.RepeatXY(descr[2], descr[3], _readme_color)
        return child.RepeatXZ(descr[2], descr[3], _readme_color)
    if kind.startswith("exprs_"):
        from sdfkit_amd import SdfExprs
        if kind == "exprs_solid_sphere":
            r = np.float32(descr[1])
            return SdfExprs.Solid(lambda p: p.Length() - r)
        if kind == "exprs_cylinder":
            return SdfExprs.Cylinder(descr[1], descr[2], *( [tuple(descr[3])] if len(descr) > 3 else []))
        if kind == "exprs_sphere":
            return SdfExprs.Sphere(descr[1], *([tuple(descr[2])] if len(descr) > 2 else []))
        return SdfExprs.Box((descr[1], descr[2], descr[3]))
    if kind == "sphere":
        return SdfFuncs.Sphere(descr[1])
    if kind == "box":
        return SdfFuncs.Box((descr[1], descr[2], descr[3]))
    if kind == "with_color":
        return mirror_sdf(descr[1]).WithColor(descr[2], descr[3], descr[4])
    if kind == "translate":
        return mirror_sdf(descr[1]).Translate(descr[2], descr[3], descr[4])
    if kind == "union":
        return SdfFuncs.Union(mirror_sdf(descr[1]), mirror_sdf(descr[2]))
    if kind == "repeat_xy_idx":
        return mirror_sdf(descr[1]).RepeatXY(descr[2], descr[3], _readme_color)
    if kind == "repeat_xz_idx":
        return mirror_sdf(descr[1]).RepeatXZ(descr[2], descr[3], _readme_color)
    raise KeyError(kind)


@pytest.mark.parametrize("name", SDF_NAMES)
def test_lowered_programs_equal_the_executed_reference(name):
    """The op list the host mirror lowers each of these scenes to -- the program hiprtc compiles for the GPU -- evaluated by the
    numpy interpreter of the IR (oracle/ir_interp.py: the semantics the sampling kernels are held to on the GPU,
    test_random_programs_match_ir_interpreter) gives the reference's own (r, g, b, w) at every point, bit for bit."""
    import json
    from oracle import ir_interp
    sdf = mirror_sdf(json.loads(str(SDF["scenes_json"]))[name]).ToSdf()
    arr, n, out = sdf.ir()
    ops = [(arr[i].opcode, arr[i].a, arr[i].b, arr[i].c, arr[i].d, arr[i].imm) for i in range(n)]
    pts, want = SDF[f"{name}/points"], SDF[f"{name}/rgbw"]
    got = ir_interp.run(ops, list(out), pts)
    for k in range(4):
        assert got[k] is not None
        same = np.asarray(got[k], np.float32).view(np.uint32) == np.ascontiguousarray(want[:, k]).view(np.uint32)
        assert same.all(), (name, k, np.argwhere(~same)[:5])


# ---- the whole path, executed (tools/gen_reference_path_vectors.py: Voxels ctor + SampleSdf + ClipToBounds + CreateMesh) ------
PATH = np.load(os.path.join(ROOT, "tests", "golden", "reference_path.npz"))
PATH_META = __import__("json").loads(str(PATH["meta_json"]))


@pytest.mark.parametrize("name", sorted(PATH_META))
def test_oracle_path_equals_the_executed_reference(name):
    """Sample points, index mapping and scatter of Voxels.SampleSdf (Voxels.cs:72-125), the clip value of ClipToBounds
    (:133-167) and the mesh CreateMesh builds from that volume: the oracle's volumes and meshes against the ones the reference's
    own source produced when executed -- bit for bit.  (Nine of the cases are scenes of the reference's NUnit tests; the executed
    source gave every vertex count they assert -- 104, 54, 312, 0, 384, 384, 7456, 1248, 1248 -- checked by the generator and again here.)"""
    md = PATH_META[name]
    sc = oracle_scene(md["scene"])
    nx, ny, nz = md["grid"]
    v, c = O.sample(sc, md["min"], md["max"], nx, ny, nz)
    if md["clip"]:
        O.clip_to_bounds(v, md["min"], md["max"])
    assert np.array_equal(v.view(np.uint32), PATH[f"{name}/values"].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), PATH[f"{name}/colors"].view(np.uint32))
    m = O.march(v, c, md["min"], md["max"], iso=md["iso"], step=md["step"])
    assert np.array_equal(m.triangles, PATH[f"{name}/faces"])
    assert np.array_equal(m.grid_vertices, PATH[f"{name}/vertices"])
    assert np.array_equal(m.colors, PATH[f"{name}/out_colors"])
    assert np.array_equal(m.grid_normals, PATH[f"{name}/normals"], equal_nan=True)
    assert m.impossible13 == md["console_lines"]
    # ... and the mesh CreateMesh RETURNS: Mesh.cs (constructor, Measure, Transform) and the T S T composition of MarchingCubes.cs:85-90
    # executed too, over a float32 restatement of the BCL's Matrix4x4 / Vector3.Transform (tools/gen_reference_path_vectors.py)
    assert np.array_equal(m.vertices, PATH[f"{name}/final_vertices"])
    assert np.array_equal(m.normals, PATH[f"{name}/final_normals"], equal_nan=True)
    if len(m.vertices):
        assert np.array_equal(m.min, PATH[f"{name}/final_min"]) and np.array_equal(m.max, PATH[f"{name}/final_max"])
    # the vertex counts the reference's own NUnit tests assert for these scenes (Tests/MarchingCubesTests.cs:11-115, Tests/SdfTests.cs:29-52)
    nunit = {"colored_spheres_32": 104, "sphere_32_clipped": 1248, "nunit_sphere5": 54, "nunit_sphere10": 312, "nunit_unclipped_sphere10": 0,
             "nunit_clipped_sphere10": 384, "nunit_box10": 384, "nunit_create_mesh_sphere": 1248, "nunit_cylinder50": 7456, "nunit_solid_sphere": 1248}
    if name in nunit:
        assert len(m.vertices) == len(PATH[f"{name}/vertices"]) == nunit[name]
    if name == "colored_spheres_32":
        assert m.colors[0][0] > 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PATH_META))
def test_hip_path_volume_and_mesh_equal_the_executed_reference(gpu, name):
    from sdfkit_amd import MarchingCubes, Voxels
    md = PATH_META[name]
    sdf = mirror_sdf(md["scene"]).ToSdf()
    nx, ny, nz = md["grid"]
    vol = Voxels.SampleSdf(sdf, md["min"], md["max"], nx, ny, nz)
    if md["clip"]:
        vol.ClipToBounds()
    assert np.array_equal(np.asarray(vol.Values).view(np.uint32), PATH[f"{name}/values"].view(np.uint32))
    assert np.array_equal(np.asarray(vol.Colors).view(np.uint32), PATH[f"{name}/colors"].view(np.uint32))
    m = MarchingCubes.CreateMesh(vol, md["iso"], md["step"])
    assert np.array_equal(m.Triangles, PATH[f"{name}/faces"])
    assert np.array_equal(m.Colors, PATH[f"{name}/out_colors"])
    assert len(m.Vertices) == len(PATH[f"{name}/vertices"])
    # the final arrays: what the reference's CreateMesh returns after Mesh.Transform and Measure (executed: Mesh.cs, MarchingCubes.cs:85-90)
    assert np.array_equal(m.Vertices, PATH[f"{name}/final_vertices"])
    assert np.array_equal(m.Normals, PATH[f"{name}/final_normals"], equal_nan=True)
    if len(m.Vertices):
        assert np.array_equal(np.asarray(m.Min, np.float32), PATH[f"{name}/final_min"]) and np.array_equal(np.asarray(m.Max, np.float32), PATH[f"{name}/final_max"])


@pytest.mark.skipif(not os.path.exists("/root/reference/SdfKit/Voxels.cs"), reason="the reference tree only exists in the build container")
def test_path_vectors_regenerate_from_the_reference_source():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_reference_path_vectors as G
    it, console = G.load()
    name = "box_16_clipped_step2"
    descr, mn, mx, grid, clip, iso, step = G.CASES[name]
    values, colors, v, c, n, f, lines, final = G.run_case(it, console, descr, mn, mx, grid, clip, iso, step)
    assert np.array_equal(final["vertices"], PATH[f"{name}/final_vertices"]) and np.array_equal(final["max"], PATH[f"{name}/final_max"])
    assert np.array_equal(values, PATH[f"{name}/values"]) and np.array_equal(colors, PATH[f"{name}/colors"])
    assert np.array_equal(v, PATH[f"{name}/vertices"]) and np.array_equal(f, PATH[f"{name}/faces"])
    assert np.array_equal(n, PATH[f"{name}/normals"], equal_nan=True)


def _digest(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


BIG_PATH = os.path.join(ROOT, "tests", "golden", "reference_path_big.json")


@pytest.mark.skipif(not os.path.exists(BIG_PATH), reason="tests/golden/reference_path_big.json not generated")
def test_oracle_equals_the_executed_reference_at_128_cubed():
    """Tests/MarchingCubesTests.cs:141-171 (Sdfs.Sphere(3) in 128^3): the reference's source, executed, gives the asserted 72 240
    vertices; its arrays are kept as SHA-256 digests (2.1 M voxels, 72 k vertices), and the oracle's arrays have the same digests."""
    import json
    for name, md in json.load(open(BIG_PATH)).items():
        sc = oracle_scene(md["scene"])
        nx, ny, nz = md["grid"]
        v, c = O.sample(sc, md["min"], md["max"], nx, ny, nz)
        if md["clip"]:
            O.clip_to_bounds(v, md["min"], md["max"])
        m = O.march(v, c, md["min"], md["max"], iso=md["iso"], step=md["step"])
        assert len(m.vertices) == md["vertices"] == 72240 and len(m.triangles) == md["indices"]
        got = {"values": _digest(v), "colors": _digest(c), "vertices": _digest(m.grid_vertices), "out_colors": _digest(m.colors),
               "normals": _digest(m.grid_normals), "faces": _digest(m.triangles), "final_vertices": _digest(m.vertices),
               "final_normals": _digest(m.normals)}
        assert got == md["sha256"]
        assert np.array_equal(m.min, np.array(md["final_min"], np.float32)) and np.array_equal(m.max, np.array(md["final_max"], np.float32))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIG_PATH), reason="tests/golden/reference_path_big.json not generated")
def test_hip_path_equals_the_executed_reference_at_128_cubed(gpu):
    import json
    from sdfkit_amd import MarchingCubes, Voxels
    for name, md in json.load(open(BIG_PATH)).items():
        sdf = mirror_sdf(md["scene"]).ToSdf()
        nx, ny, nz = md["grid"]
        vol = Voxels.SampleSdf(sdf, md["min"], md["max"], nx, ny, nz)
        if md["clip"]:
            vol.ClipToBounds()
        assert _digest(np.asarray(vol.Values, np.float32)) == md["sha256"]["values"]
        assert _digest(np.asarray(vol.Colors, np.float32)) == md["sha256"]["colors"]
        m = MarchingCubes.CreateMesh(vol, md["iso"], md["step"])
        assert len(m.Vertices) == md["vertices"] == 72240
        assert _digest(np.asarray(m.Triangles, np.int32)) == md["sha256"]["faces"]
        assert _digest(np.asarray(m.Colors, np.float32)) == md["sha256"]["out_colors"]
        assert _digest(np.asarray(m.Vertices, np.float32)) == md["sha256"]["final_vertices"]
        assert _digest(np.asarray(m.Normals, np.float32)) == md["sha256"]["final_normals"]
