#!/usr/bin/env python3
"""Golden vectors for the WHOLE path, produced by EXECUTING the reference's own source (build container only).

`new Voxels(min, max, nx, ny, nz)` -> `voxels.SampleSdf(sdf)` (Voxels.cs:23-40, 72-125: cell-centred sample points, the index
mapping of the batches, the scatter into Values / Colors) -> `voxels.ClipToBounds()` (Voxels.cs:133-167) ->
`MarchingCubes.CreateMesh(voxels, iso, step)` (MarchingCubes.cs, Cell.cs), with the scene a composition of the per-point
catalogue `SdfFuncs` / `SdfFuncEx` turned into a batched `Sdf` by `SdfFuncEx.ToSdf` (Sdf.cs:301-313) -- all of it parsed where
it lies under /root/reference and run by tools/cs_subset.py.  `Parallel.For` runs its batches in order (the reference's result
does not depend on the order: every voxel is written once); `Memory<T>` / `Span<T>` are views of the arrays.

Output, committed: tests/golden/reference_path.npz -- per case the scene description, bounds, grid, clip / iso / step, the sampled
Values and Colors, and what `new Mesh(...)` receives (MarchingCubes.cs:84).  Nine cases are scenes of the reference's own NUnit
tests (the batched `Sdfs.Sphere` / `Sdfs.Box` of Sdf.cs:118-214 and `Sdfs.Cylinder` = `SdfExprs.Cylinder(...).ToSdf()` among them), and
the executed source gives every vertex count those tests assert: 104, 54, 312, 0, 384, 384, 7456, 1248, 1248 (the generator refuses to
write the file otherwise); `--big` adds the tenth, the 128^3 sphere with 72 240 vertices, as digests.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cs_subset as CS                      # noqa: E402
import gen_luts                             # noqa: E402
import gen_reference_sdf_vectors as GS      # noqa: E402
import gen_reference_vectors as GM          # noqa: E402

REF = "/root/reference/SdfKit"
F32 = np.float32


class ParallelHost:
    @staticmethod
    def For(lo, hi, options, init, body, final):
        local = init()
        for i in range(lo, hi):
            local = body(i, None, local)
        final(local)


class ParallelOptions:
    MaxDegreeOfParallelism = -1


class Namespace:
    pass


class MathHost(GS.MathHost):
    @staticmethod
    def Min(a, b):
        return min(a, b)


def load():
    sdf_cs = open(os.path.join(REF, "Sdf.cs")).read()
    vec_cs = open(os.path.join(REF, "VectorData.cs")).read()
    expr_cs = open(os.path.join(REF, "SdfExpr.cs")).read()
    vox_cs = open(os.path.join(REF, "Voxels.cs")).read()
    head = vox_cs[vox_cs.index("{", vox_cs.index("public class Voxels")) + 1:vox_cs.index("public Voxels(float[,,] values")]
    voxels = "\n".join([
        "public class Voxels {", head,
        GS.cut_braced(vox_cs, r"public Voxels\(float\[,,\] values"),
        GS.cut_braced(vox_cs, r"public Voxels\(Vector3 min, Vector3 max, int nx"),
        GS.cut_braced(vox_cs, r"public void SampleSdf\(Sdf sdf"),
        GS.cut_braced(vox_cs, r"public void ClipToBounds\(\)"),
        "}",
    ])
    mesh_cs = open(os.path.join(REF, "Mesh.cs")).read()
    mesh_head = mesh_cs[mesh_cs.index("{", mesh_cs.index("public class Mesh")) + 1:mesh_cs.index("public Mesh(Vector3[] vertices")]
    mesh = "\n".join(["public class Mesh {", mesh_head, GS.cut_braced(mesh_cs, r"public Mesh\(Vector3\[\] vertices"),
                      GS.cut_braced(mesh_cs, r"void Measure\(\)"), GS.cut_braced(mesh_cs, r"public void Transform\(Matrix4x4 transform\)"), "}"])
    text = "\n".join([
        GS.cut_braced(sdf_cs, r"public class SdfConfig\b"),
        GS.cut_braced(sdf_cs, r"public static class Sdfs\b"),
        GS.cut_braced(sdf_cs, r"public static class SdfFuncs\b"),
        GS.cut_braced(sdf_cs, r"public static class SdfFuncEx\b"),
        GS.cut_braced(expr_cs, r"public struct SdfIndexedInput\b"),
        # the primitives of the expression-tree catalogue are plain expression lambdas (SdfExpr.cs:18-51); what is NOT executed is
        # the LINQ plumbing that batches them (SdfExprCompiler, SdfExpr.cs:229-273): SdfFuncEx.ToSdf's loop stands in for it
        "public static class SdfExprs {",
        GS.cut_braced(expr_cs, r"public static SdfExpr Box\(Vector3 bounds\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Box\(float bounds\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Cylinder\(float r, float h, Vector3 color\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Cylinder\(float r, float h\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Sphere\(float r, Vector3 color\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Sphere\(float r\)"),
        GS.cut_braced(expr_cs, r"public static SdfExpr Union\(SdfExpr a, SdfExpr b\)"),
        GS.cut_braced(expr_cs, r"public static SdfExpr Solid\(SdfDistExpr sdf, Vector3 color\)"),
        GS.cut_expression_bodied(expr_cs, r"public static SdfExpr Solid\(SdfDistExpr sdf\)"),
        "}",
        # the combinators: LINQ expression trees (SdfExpr.cs:76-212), built through ExpressionHost and evaluated as built
        GS.cut_braced(expr_cs, r"public static class SdfExprEx\b"),
        "public static class VectorOps {",
        GS.cut_expression_bodied(vec_cs, r"public static float Mod\(float a, float b\)"),
        GS.cut_expression_bodied(vec_cs, r"public static float VMax\(Vector3 v\)"),
        "}",
        voxels,
        mesh,
        open(os.path.join(REF, "MarchingCubes.cs")).read(),
        open(os.path.join(REF, "Cell.cs")).read(),
    ])
    classes = CS.parse(text)
    luts = type("LutsHost", (), {})()
    for name, shape, vals in gen_luts.parse_tables(open(os.path.join(REF, "Luts.cs")).read()):
        setattr(luts, name, np.array(vals, dtype=np.int8).reshape(shape))
    system = Namespace()
    system.Threading = Namespace()
    system.Threading.Tasks = Namespace()
    system.Threading.Tasks.Parallel = ParallelHost
    console = GM.ConsoleHost()
    hosts = {"Luts": luts, "Math": MathHost, "MathF": GS.MathFHost, "Console": console, "Vector3": Vector3Host, "Matrix4x4": Matrix4x4,
             "System": system, "ParallelOptions": ParallelOptions, "Expression": ExpressionHost,
             # the reflection handles of SdfExprMembers / SdfExprs (SdfExpr.cs:33, 216-225): what they denote
             "Vector4Ctor": lambda v, w: CS.Vec4(v, w), "WOfVector4": "W", "PositionOfInstance": "Position", "CellOfInstance": "Index"}
    it = CS.Interp(classes, hosts)
    it.static_imports = ["VectorOps"]
    # what `new Mesh(...)` receives at MarchingCubes.cs:84, before Mesh.Transform overwrites the arrays in place
    it.on_new = {"Mesh": lambda o: PRE.update(vertices=[CS.value_copy(q) for q in o.f["Vertices"]], colors=[CS.value_copy(q) for q in o.f["Colors"]],
                                              normals=[CS.value_copy(q) for q in o.f["Normals"]], faces=list(o.f["Triangles"]))}
    return it, console


PRE = {}


class ParamEx:
    def __init__(self, name):
        self.name = name


class LambdaEx:
    """System.Linq.Expressions.Expression<TDelegate> built by Expression.Lambda: evaluated directly (Compile() is the identity)."""
    def __init__(self, body, params):
        self.body, self.params = body, params

    def __call__(self, *args):
        return ev(self.body, dict(zip(self.params, args)))

    def Compile(self, *a):
        return self


def ev(n, env):
    if isinstance(n, ParamEx):
        return env[n]
    if not (isinstance(n, tuple) and n and isinstance(n[0], str) and n[0].startswith("ex:")):
        return n   # a constant that was passed as it is
    k = n[0]
    if k == "ex:invoke":
        return n[1](*[ev(a, env) for a in n[2]])
    if k == "ex:block":
        r = None
        for q in n[2]:
            r = ev(q, env)
        return r
    if k == "ex:assign":
        env[n[1]] = CS.value_copy(ev(n[2], env))
        return env[n[1]]
    if k == "ex:cond":
        return ev(n[2], env) if ev(n[1], env) else ev(n[3], env)
    if k == "ex:lt":
        return CS.compare("<", ev(n[1], env), ev(n[2], env))
    if k == "ex:field":
        o = ev(n[1], env)
        return o.f[n[2]] if isinstance(o, CS.Instance) else getattr(o, n[2])
    if k == "ex:new":
        return n[1](*[ev(a, env) for a in n[2]])
    if k == "ex:const":
        return n[1]
    raise NotImplementedError(k)


class ExpressionHost:
    """the handful of System.Linq.Expressions factory methods SdfExpr.cs:16-212 calls (BCL, not reference code)"""
    @staticmethod
    def Parameter(ty, name):
        return ParamEx(name)

    Variable = Parameter

    @staticmethod
    def Invoke(target, *args):
        return ("ex:invoke", target, args)

    @staticmethod
    def Lambda(body, *params):
        flat = []
        for q in params:
            flat += list(q) if isinstance(q, list) else [q]
        return LambdaEx(body, flat)

    @staticmethod
    def Block(*a):
        vars_, exprs = (a[0], a[1:]) if isinstance(a[0], list) else ([], a)
        return ("ex:block", vars_, exprs)

    @staticmethod
    def Assign(v, e):
        return ("ex:assign", v, e)

    @staticmethod
    def Condition(t, a, b):
        return ("ex:cond", t, a, b)

    @staticmethod
    def LessThan(a, b):
        return ("ex:lt", a, b)

    @staticmethod
    def Field(e, info):
        return ("ex:field", e, info)

    @staticmethod
    def New(ctor, *args):
        return ("ex:new", ctor, args)

    @staticmethod
    def Constant(v):
        return ("ex:const", v)


class Matrix4x4:
    """System.Numerics.Matrix4x4 (BCL, not reference code), float32, software path, as oracle/sdfk_oracle.h documents it: row-vector
    convention, every product summed left to right with one rounding per operation, Invert by cofactor expansion."""
    NAMES = [f"M{r}{c}" for r in range(1, 5) for c in range(1, 5)]

    def __init__(self, vals=None):
        vals = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1] if vals is None else vals
        for n, v in zip(self.NAMES, vals):
            object.__setattr__(self, n, F32(v))

    def __setattr__(self, n, v):
        if CS.is_f64(v):
            raise TypeError("double into a float field")
        object.__setattr__(self, n, F32(v))

    def vals(self):
        return [getattr(self, n) for n in self.NAMES]

    def cs_copy(self):
        return Matrix4x4(self.vals())

    def cs_mul(self, o):
        a, b = np.array(self.vals(), dtype=np.float32).reshape(4, 4), np.array(o.vals(), dtype=np.float32).reshape(4, 4)
        out = []
        for r in range(4):
            for c in range(4):
                out.append(F32(F32(F32(F32(a[r, 0] * b[0, c]) + F32(a[r, 1] * b[1, c])) + F32(a[r, 2] * b[2, c])) + F32(a[r, 3] * b[3, c])))
        return Matrix4x4(out)

    @staticmethod
    def CreateTranslation(*a):
        x, y, z = (a[0].X, a[0].Y, a[0].Z) if len(a) == 1 else a
        if any(CS.is_f64(q) for q in (x, y, z)):
            raise TypeError("double argument")
        return Matrix4x4([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1])

    @staticmethod
    def CreateScale(x, y, z):
        if any(CS.is_f64(q) for q in (x, y, z)):
            raise TypeError("double argument")
        return Matrix4x4([x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1])

    @staticmethod
    def Transpose(m):
        v = np.array(m.vals(), dtype=np.float32).reshape(4, 4).T
        return Matrix4x4(list(v.ravel()))

    @staticmethod
    def Invert(mat, out):
        a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p = mat.vals()
        with np.errstate(all="ignore"):
            kp_lo, jp_ln, jo_kn = F32(F32(k * p) - F32(l * o)), F32(F32(j * p) - F32(l * n)), F32(F32(j * o) - F32(k * n))
            ip_lm, io_km, in_jm = F32(F32(i * p) - F32(l * m)), F32(F32(i * o) - F32(k * m)), F32(F32(i * n) - F32(j * m))
            a11 = F32(F32(F32(f * kp_lo) - F32(g * jp_ln)) + F32(h * jo_kn))
            a12 = F32(-F32(F32(F32(e * kp_lo) - F32(g * ip_lm)) + F32(h * io_km)))
            a13 = F32(F32(F32(e * jp_ln) - F32(f * ip_lm)) + F32(h * in_jm))
            a14 = F32(-F32(F32(F32(e * jo_kn) - F32(f * io_km)) + F32(g * in_jm)))
            det = F32(F32(F32(F32(a * a11) + F32(b * a12)) + F32(c * a13)) + F32(d * a14))
            if abs(det) < np.finfo(np.float32).smallest_subnormal:
                out.value = Matrix4x4([np.nan] * 16)
                return False
            inv = F32(F32(1) / det)
            gp_ho, fp_hn, fo_gn = F32(F32(g * p) - F32(h * o)), F32(F32(f * p) - F32(h * n)), F32(F32(f * o) - F32(g * n))
            ep_hm, eo_gm, en_fm = F32(F32(e * p) - F32(h * m)), F32(F32(e * o) - F32(g * m)), F32(F32(e * n) - F32(f * m))
            gl_hk, fl_hj, fk_gj = F32(F32(g * l) - F32(h * k)), F32(F32(f * l) - F32(h * j)), F32(F32(f * k) - F32(g * j))
            el_hi, ek_gi, ej_fi = F32(F32(e * l) - F32(h * i)), F32(F32(e * k) - F32(g * i)), F32(F32(e * j) - F32(f * i))
            t = lambda x, y, z, u, v, w, sign: F32(F32(sign * F32(F32(F32(x * u) - F32(y * v)) + F32(z * w))) * inv)
            r = [F32(a11 * inv), t(b, c, d, kp_lo, jp_ln, jo_kn, -1), t(b, c, d, gp_ho, fp_hn, fo_gn, 1), t(b, c, d, gl_hk, fl_hj, fk_gj, -1),
                 F32(a12 * inv), t(a, c, d, kp_lo, ip_lm, io_km, 1), t(a, c, d, gp_ho, ep_hm, eo_gm, -1), t(a, c, d, gl_hk, el_hi, ek_gi, 1),
                 F32(a13 * inv), t(a, b, d, jp_ln, ip_lm, in_jm, -1), t(a, b, d, fp_hn, ep_hm, en_fm, 1), t(a, b, d, fl_hj, el_hi, ej_fi, -1),
                 F32(a14 * inv), t(a, b, c, jo_kn, io_km, in_jm, 1), t(a, b, c, fo_gn, eo_gm, en_fm, -1), t(a, b, c, fk_gj, ek_gi, ej_fi, 1)]
        out.value = Matrix4x4(r)
        return True


class Vector3Host(GS.Vector3Host):
    Normalize = staticmethod(GM.Vector3Host.Normalize)

    @staticmethod
    def Transform(v, m):   # BCL: position . matrix, row vector, products summed left to right, + the translation row
        with np.errstate(all="ignore"):
            return CS.Vec3(F32(F32(F32(F32(v.X * m.M11) + F32(v.Y * m.M21)) + F32(v.Z * m.M31)) + m.M41),
                           F32(F32(F32(F32(v.X * m.M12) + F32(v.Y * m.M22)) + F32(v.Z * m.M32)) + m.M42),
                           F32(F32(F32(F32(v.X * m.M13) + F32(v.Y * m.M23)) + F32(v.Z * m.M33)) + m.M43))

    @staticmethod
    def TransformNormal(v, m):
        with np.errstate(all="ignore"):
            return CS.Vec3(F32(F32(F32(v.X * m.M11) + F32(v.Y * m.M21)) + F32(v.Z * m.M31)),
                           F32(F32(F32(v.X * m.M12) + F32(v.Y * m.M22)) + F32(v.Z * m.M32)),
                           F32(F32(F32(v.X * m.M13) + F32(v.Y * m.M23)) + F32(v.Z * m.M33)))

    @staticmethod
    def Dot(a, b):   # BCL: the three products summed left to right in float32
        with np.errstate(all="ignore"):
            return F32(F32(F32(a.X * b.X) + F32(a.Y * b.Y)) + F32(a.Z * b.Z))


def run_case(it, console, descr, mn, mx, grid, clip, iso, step):
    if descr[0].startswith("sdfs_"):   # the batched catalogue (Sdf.cs:118-214): already an `Sdf`
        sdf = build(it, descr)
    else:
        sdf = it.call_extension(build(it, descr), "ToSdf", [])
    vox = it.new("Voxels", [CS.Vec3(*mn), CS.Vec3(*mx), int(grid[0]), int(grid[1]), int(grid[2])])
    it.invoke(vox, "Voxels", it.pick("Voxels", "SampleSdf", [sdf, 2048, -1]), [sdf, 2048, -1])
    if clip:
        it.invoke(vox, "Voxels", it.pick("Voxels", "ClipToBounds", []), [])
    values = np.array(vox.f["Values"], dtype=np.float32)
    colors = np.zeros(values.shape + (3,), dtype=np.float32)
    for ix in np.ndindex(*values.shape):
        c = vox.f["Colors"][ix]
        colors[ix] = (c.X, c.Y, c.Z)
    console.lines.clear()
    PRE.clear()
    mesh = it.call_static("MarchingCubes", "CreateMesh", [vox, F32(iso), int(step), None])
    v3 = lambda lst: np.array([[q.X, q.Y, q.Z] for q in lst], dtype=np.float32).reshape(-1, 3)
    final = {"vertices": v3(mesh.f["Vertices"]), "normals": v3(mesh.f["Normals"]),
             "min": np.array([mesh.f["Min"].X, mesh.f["Min"].Y, mesh.f["Min"].Z], dtype=np.float32),
             "max": np.array([mesh.f["Max"].X, mesh.f["Max"].Y, mesh.f["Max"].Z], dtype=np.float32)}
    return values, colors, v3(PRE["vertices"]), v3(PRE["colors"]), v3(PRE["normals"]), np.array(PRE["faces"], dtype=np.int32), len(console.lines), final


def build(it, d):
    f = F32
    S = lambda *a: it.call_static("SdfFuncs", *a)
    X = lambda recv, name, *a: it.call_extension(recv, name, list(a))
    kind = d[0]
    if kind in ("exprs_union", "exprs_color", "exprs_translate", "exprs_repeat_x", "exprs_repeat_y", "exprs_repeat_xy", "exprs_repeat_xy_idx", "exprs_repeat_xz_idx"):
        E = lambda recv, name, *a: it.call_extension(recv, name, list(a))
        if kind == "exprs_union":
            return it.call_static("SdfExprs", "Union", [build(it, d[1]), build(it, d[2])])
        if kind == "exprs_color":
            return E(build(it, d[1]), "Color", f(d[2]), f(d[3]), f(d[4]))
        if kind == "exprs_translate":   # sdf.ModifyInput(p => p - offset): the caller's lambda (BASELINE config C4), a closure over `offset`
            off = CS.Vec3(d[2], d[3], d[4])
            return E(build(it, d[1]), "ModifyInput", lambda p: CS.arith("-", p, off))
        if kind == "exprs_repeat_x":
            return E(build(it, d[1]), "RepeatX", f(d[2]))
        if kind == "exprs_repeat_y":
            return E(build(it, d[1]), "RepeatY", f(d[2]))
        if kind == "exprs_repeat_xy":
            return E(build(it, d[1]), "RepeatXY", f(d[2]), f(d[3]))
        user = CS.Interp(CS.parse(GS.README_COLOUR), {"Vector3": GS.Vector3Host})
        colour = lambda i, p, q: user.call_static("UserCode", "Colour", [i, p, q])
        return E(build(it, d[1]), "RepeatXY" if kind == "exprs_repeat_xy_idx" else "RepeatXZ", f(d[2]), f(d[3]), colour)
    if kind == "exprs_solid_sphere":   # SdfExprs.Solid(p => p.Length() - r): the caller's distance lambda (Tests/SdfTests.cs:42-52)
        user = CS.Interp(CS.parse("public static class UserDist { public static float D(Vector3 p, float r) => p.Length() - r; }"), {})
        return it.call_static("SdfExprs", "Solid", [lambda p: user.call_static("UserDist", "D", [p, f(d[1])])])
    if kind == "exprs_cylinder":
        return it.call_static("SdfExprs", "Cylinder", [f(d[1]), f(d[2])] + ([CS.Vec3(*d[3])] if len(d) > 3 else []))
    if kind == "exprs_sphere":
        return it.call_static("SdfExprs", "Sphere", [f(d[1])] + ([CS.Vec3(*d[2])] if len(d) > 2 else []))
    if kind == "exprs_box":
        return it.call_static("SdfExprs", "Box", [CS.Vec3(d[1], d[2], d[3])])
    if kind == "sdfs_sphere":
        return it.call_static("Sdfs", "Sphere", [f(d[1])])
    if kind == "sdfs_box":
        return it.call_static("Sdfs", "Box", [f(d[1])])
    if kind == "sdfs_plane":
        return it.call_static("Sdfs", "Plane", [CS.Vec3(d[1], d[2], d[3]), f(d[4])])
    if kind == "sphere":
        return S("Sphere", [f(d[1])])
    if kind == "box":
        return S("Box", [CS.Vec3(d[1], d[2], d[3])])
    if kind == "with_color":
        return X(build(it, d[1]), "WithColor", f(d[2]), f(d[3]), f(d[4]))
    if kind == "translate":
        return X(build(it, d[1]), "Translate", f(d[2]), f(d[3]), f(d[4]))
    if kind == "union":
        return S("Union", [build(it, d[1]), build(it, d[2])])
    user = CS.Interp(CS.parse(GS.README_COLOUR), {"Vector3": GS.Vector3Host})
    colour = lambda i, p, q: user.call_static("UserCode", "Colour", [i, p, q])
    if kind == "repeat_xy_idx":
        return X(build(it, d[1]), "RepeatXY", f(d[2]), f(d[3]), colour)
    if kind == "repeat_xz_idx":
        return X(build(it, d[1]), "RepeatXZ", f(d[2]), f(d[3]), colour)
    raise KeyError(kind)


CASES = {
    # Tests/MarchingCubesTests.cs:11-28 (ColoredSpheres): 104 vertices
    "colored_spheres_32": (["union", ["translate", ["with_color", ["sphere", 0.4], 1, 0.2, 0.3], -1, 0, 0],
                            ["translate", ["with_color", ["sphere", 0.2], 0.1, 1, 0.3], 1, 0, 0]], [-3, -3, -3], [3, 3, 3], (32, 32, 32), False, 0.0, 1),
    # a sphere of radius 0.5 in 32^3 with clipToBounds (Tests/SdfTests.cs:29-52): 1248 vertices
    "sphere_32_clipped": (["sphere", 0.5], [-1, -1, -1], [1, 1, 1], (32, 32, 32), True, 0.0, 1),
    "readme_24_clipped": (["repeat_xy_idx", ["sphere", 0.5], 1.125, 1.125], [-2.8125] * 3, [2.8125] * 3, (24, 24, 24), True, 0.0, 1),
    "union_box_sphere_iso": (["union", ["box", 0.5, 0.5, 0.5], ["translate", ["sphere", 0.6], 0.4, 0.3, -0.2]], [-1.5, -1.25, -1.75], [1.5, 1.75, 1.25],
                             (20, 18, 22), False, 0.1, 1),
    "box_16_clipped_step2": (["box", 0.6, 0.6, 0.6], [-1, -1, -1], [1, 1, 1], (17, 16, 19), True, 0.0, 2),
    # the reference's other known answers on the path (Tests/MarchingCubesTests.cs:31-115, Tests/SdfTests.cs:29-39), batched Sdfs.*
    "nunit_sphere5": (["sdfs_sphere", 1.0], [-1.5] * 3, [1.5] * 3, (5, 5, 5), False, 0.0, 1),
    "nunit_sphere10": (["sdfs_sphere", 2.0], [-2.5] * 3, [2.5] * 3, (10, 10, 10), False, 0.0, 1),
    "nunit_unclipped_sphere10": (["sdfs_sphere", 2.0], [-1] * 3, [1] * 3, (10, 10, 10), False, 0.0, 1),
    "nunit_clipped_sphere10": (["sdfs_sphere", 2.0], [-1] * 3, [1] * 3, (10, 10, 10), True, 0.0, 1),
    "nunit_box10": (["sdfs_box", 2.0], [-2.5] * 3, [2.5] * 3, (10, 10, 10), False, 0.0, 1),
    "nunit_create_mesh_sphere": (["sdfs_sphere", 0.5], [-1] * 3, [1] * 3, (32, 32, 32), True, 0.0, 1),
    # Tests/MarchingCubesTests.cs:118-138 (Cylinder50): Sdfs.Cylinder(1, 3) = SdfExprs.Cylinder(1, 3).ToSdf(), 7456 vertices
    "nunit_cylinder50": (["exprs_cylinder", 1.0, 3.0], [-1.5, -3.5, -1.5], [1.5, 3.5, 1.5], (50, 50, 50), False, 0.0, 1),
    "exprs_union_coloured": (["union", ["translate", ["exprs_sphere", 0.45, [0.2, 0.4, 0.6]], 0.3, 0.0, -0.2], ["exprs_cylinder", 0.3, 0.5, [0.9, 0.1, 0.4]]],
                             [-1, -1, -1], [1, 1, 1], (18, 20, 16), True, 0.0, 1),
    # the expression-tree catalogue (SdfExpr.cs:53-212), its LINQ trees built and evaluated: the README scene as the README writes it,
    # nested unions with translated primitives (BASELINE config C4's construction, three of its eight), plain repeats, Color
    # Tests/SdfTests.cs:42-52 (SolidSphere): SdfExprs.Solid(p => p.Length() - 0.5f) in 32^3, clipped: 1248 vertices
    "nunit_solid_sphere": (["exprs_solid_sphere", 0.5], [-1] * 3, [1] * 3, (32, 32, 32), True, 0.0, 1),
    "exprs_readme_24_clipped": (["exprs_repeat_xy_idx", ["exprs_sphere", 0.5], 1.125, 1.125], [-2.8125] * 3, [2.8125] * 3, (24, 24, 24), True, 0.0, 1),
    "exprs_union3_translated": (["exprs_union", ["exprs_union", ["exprs_translate", ["exprs_sphere", 0.6], -1, -1, -1], ["exprs_translate", ["exprs_box", 0.5, 0.5, 0.5], 1, -1, -1]],
                                 ["exprs_translate", ["exprs_cylinder", 0.4, 0.6], -1, 1, -1]], [-2, -2, -2], [2, 2, 2], (22, 20, 12), True, 0.0, 1),
    "exprs_repeat_x_y_colour": (["exprs_repeat_y", ["exprs_repeat_x", ["exprs_color", ["exprs_sphere", 0.3], 0.2, 0.4, 0.6], 0.9], 1.1], [-2, -2, -1], [2, 2, 1],
                                (21, 19, 9), True, 0.0, 1),
    "exprs_repeat_xy_plain": (["exprs_repeat_xy", ["exprs_box", 0.25, 0.35, 0.3], 1.0, 1.25], [-2, -2, -1], [2, 2, 1], (20, 22, 10), False, 0.0, 1),
    "plane_tilted": (["sdfs_plane", 0.3, 0.0, 1.0, 0.05], [-1, -1, -1], [1, 1, 1], (14, 12, 10), False, 0.0, 1),
    "repeat_xz_box_clipped": (["repeat_xz_idx", ["box", 0.3, 0.3, 0.3], 1.5, 0.875], [-2.5, -1.0, -2.0], [2.5, 1.0, 2.0], (21, 9, 25), True, 0.0, 1),
}
EXPECT_VERTICES = {"colored_spheres_32": 104, "sphere_32_clipped": 1248, "nunit_sphere5": 54, "nunit_sphere10": 312, "nunit_unclipped_sphere10": 0,
                   "nunit_clipped_sphere10": 384, "nunit_box10": 384, "nunit_create_mesh_sphere": 1248, "nunit_cylinder50": 7456,
                   "nunit_solid_sphere": 1248}


def digest(a):
    """SHA-256 of the bytes of an array"""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# Tests/MarchingCubesTests.cs:141-171 (Sphere128Progress): 72 240 vertices.  2.1 M voxels through a tree-walking interpreter take
# minutes, and the arrays are too big for a fixture: `--big` runs it and stores the counts and SHA-256 digests of the arrays.
BIG = {"nunit_sphere128": (["sdfs_sphere", 3.0], [-3.1] * 3, [3.1] * 3, (128, 128, 128), False, 0.0, 1, 72240)}


def main_big():
    it, console = load()
    path = os.path.join(ROOT, "tests", "golden", "reference_path_big.json")
    out = {}
    for name, (descr, mn, mx, grid, clip, iso, step, expect) in BIG.items():
        values, colors, v, c, n, f, lines, final = run_case(it, console, descr, mn, mx, grid, clip, iso, step)
        print(f"{name}: {len(v)} vertices, {len(f) // 3} triangles")
        if len(v) != expect:
            raise SystemExit(f"{name}: the reference's own test asserts {expect} vertices")
        out[name] = {"scene": descr, "min": mn, "max": mx, "grid": list(grid), "clip": clip, "iso": iso, "step": step, "vertices": len(v), "indices": len(f),
                     "sha256": {"values": digest(values), "colors": digest(colors), "vertices": digest(v), "out_colors": digest(c),
                                "normals": digest(n), "faces": digest(f), "final_vertices": digest(final["vertices"]),
                                "final_normals": digest(final["normals"])},
                     "final_min": [float(q) for q in final["min"]], "final_max": [float(q) for q in final["max"]]}
    json.dump(out, open(path, "w"), indent=1)
    print(path)


def main():
    if "--big" in sys.argv:
        return main_big()
    it, console = load()
    blob, meta = {}, {}
    for name, (descr, mn, mx, grid, clip, iso, step) in CASES.items():
        values, colors, v, c, n, f, lines, final = run_case(it, console, descr, mn, mx, grid, clip, iso, step)
        print(f"{name}: grid {grid} clip {clip} iso {iso} step {step} -> {len(v)} vertices, {len(f) // 3} triangles, {lines} console lines")
        if name in EXPECT_VERTICES and len(v) != EXPECT_VERTICES[name]:
            raise SystemExit(f"{name}: the reference's own test asserts {EXPECT_VERTICES[name]} vertices")
        meta[name] = {"scene": descr, "min": mn, "max": mx, "grid": list(grid), "clip": clip, "iso": iso, "step": step, "console_lines": lines}
        blob[f"{name}/values"], blob[f"{name}/colors"] = values, colors
        blob[f"{name}/vertices"], blob[f"{name}/out_colors"], blob[f"{name}/normals"], blob[f"{name}/faces"] = v, c, n, f
        # ... and the mesh CreateMesh returns: after Mesh.Transform (Mesh.cs:47-64) with T(-(n-1)/2) S(size/(n-1)) T(center), Measure'd
        blob[f"{name}/final_vertices"], blob[f"{name}/final_normals"] = final["vertices"], final["normals"]
        blob[f"{name}/final_min"], blob[f"{name}/final_max"] = final["min"], final["max"]
    blob["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", "reference_path.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
