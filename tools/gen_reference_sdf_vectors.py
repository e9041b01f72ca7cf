#!/usr/bin/env python3
"""Golden vectors for the SDF catalogue, produced by EXECUTING the reference's own source (build container only).

The per-point catalogue `SdfFuncs` / `SdfFuncEx` (SdfKit/Sdf.cs:217-341: Sphere, Box, Union, Translate, WithColor, RepeatXY,
RepeatXZ through ModifyInputAndOutput), `VectorOps.Mod` / `VMax` (VectorData.cs:697-698, 860-861) and `SdfIndexedInput`
(SdfExpr.cs:71-75) are cut out of their files where they lie under /root/reference, parsed and run by tools/cs_subset.py on
seeded points.  The expression-tree catalogue `SdfExprs` builds the same arithmetic as LINQ expression trees (SdfExpr.cs:16-212),
which are outside the interpreter's subset; the batched `Sdfs.*` forms likewise (Memory<T> / Span<T>).

Output, committed: tests/golden/reference_sdf_points.npz -- per scene the points and the (r, g, b, w) the reference's lambdas
return.  tests/test_reference_vectors.py holds the oracle's orc_eval to them, bit for bit.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cs_subset as CS   # noqa: E402

REF = "/root/reference/SdfKit"
F32 = np.float32


def cut_braced(text, start_pattern):
    """the text of the declaration that starts at `start_pattern` up to its matching closing brace"""
    m = re.search(start_pattern, text)
    if not m:
        raise SystemExit(f"not found: {start_pattern}")
    i = text.index("{", m.start())
    depth, j = 0, i
    while True:
        depth += (text[j] == "{") - (text[j] == "}")
        if depth == 0:
            return text[m.start():j + 1]
        j += 1


def cut_expression_bodied(text, start_pattern):
    m = re.search(start_pattern, text)
    if not m:
        raise SystemExit(f"not found: {start_pattern}")
    return text[m.start():text.index(";", m.start()) + 1]


class Vector3Host:
    Zero = CS.Vec3()
    One = CS.Vec3(1, 1, 1)

    @staticmethod
    def Abs(v):
        return CS.Vec3(abs(v.X), abs(v.Y), abs(v.Z))

    @staticmethod
    def Max(a, b):   # BCL: (a.X > b.X) ? a.X : b.X per component
        return CS.Vec3(a.X if a.X > b.X else b.X, a.Y if a.Y > b.Y else b.Y, a.Z if a.Z > b.Z else b.Z)

    @staticmethod
    def Min(a, b):   # BCL: (a.X < b.X) ? a.X : b.X per component
        return CS.Vec3(a.X if a.X < b.X else b.X, a.Y if a.Y < b.Y else b.Y, a.Z if a.Z < b.Z else b.Z)


class MathHost:
    @staticmethod
    def Max(a, b):   # .NET Core 3.0+: IEEE 754:2019 maximum (NaN if either is NaN, +0 > -0)
        if isinstance(a, CS.F32) or isinstance(b, CS.F32):
            a, b = F32(a), F32(b)
            if np.isnan(a) or np.isnan(b):
                return F32(np.nan)
            if a == b:
                return b if np.signbit(a) else a
            return a if a > b else b
        return max(a, b)

    @staticmethod
    def Abs(x):
        return abs(x)


class MathFHost:
    @staticmethod
    def Floor(x):
        return F32(np.floor(F32(x)))

    @staticmethod
    def Sqrt(x):   # correctly rounded float32 root
        with np.errstate(all="ignore"):
            return F32(np.sqrt(F32(x)))

    @staticmethod
    def Abs(x):
        return F32(abs(F32(x)))

    @staticmethod
    def Max(a, b):
        return MathHost.Max(F32(a), F32(b))


def load():
    sdf_cs = open(os.path.join(REF, "Sdf.cs")).read()
    vec_cs = open(os.path.join(REF, "VectorData.cs")).read()
    expr_cs = open(os.path.join(REF, "SdfExpr.cs")).read()
    text = "\n".join([
        cut_braced(sdf_cs, r"public static class SdfFuncs\b"),
        cut_braced(sdf_cs, r"public static class SdfFuncEx\b"),
        cut_braced(expr_cs, r"public struct SdfIndexedInput\b"),
        "public static class VectorOps {",
        cut_expression_bodied(vec_cs, r"public static float Mod\(float a, float b\)"),
        cut_expression_bodied(vec_cs, r"public static float VMax\(Vector3 v\)"),
        "}",
    ])
    classes = CS.parse(text)
    it = CS.Interp(classes, {"Vector3": Vector3Host, "Math": MathHost, "MathF": MathFHost})
    it.static_imports = ["VectorOps"]   # (Sdf.cs:1: using static SdfKit.VectorOps)
    return it


def f(x):
    return F32(x)


# the colour lambda of the README scene (README.md:24-30; a caller's lambda, written here in the interpreter's input language)
README_COLOUR = "public static class UserCode { public static Vector3 Colour(Vector3 i, Vector3 p, Vector4 d) => 0.9f*Vector3.One - Vector3.Abs(i)/6f; }"


def scenes(it):
    """name -> (reference closure, description the test turns into the oracle's scene nodes)"""
    S = lambda *a: it.call_static("SdfFuncs", *a)
    X = lambda recv, name, *a: it.call_extension(recv, name, list(a))
    user = CS.Interp(CS.parse(README_COLOUR), {"Vector3": Vector3Host})
    colour = lambda i, p, d: user.call_static("UserCode", "Colour", [i, p, d])
    out = {}
    out["sphere"] = (S("Sphere", [f(0.75)]), ["sphere", 0.75])
    out["box3"] = (S("Box", [CS.Vec3(0.5, 0.75, 0.25)]), ["box", 0.5, 0.75, 0.25])
    out["box1"] = (S("Box", [f(0.6)]), ["box", 0.6, 0.6, 0.6])
    out["sphere_colour_translate"] = (X(X(S("Sphere", [f(0.4)]), "WithColor", f(1), f(0.2), f(0.3)), "Translate", f(-1), f(0), f(0.25)),
                                      ["translate", ["with_color", ["sphere", 0.4], 1, 0.2, 0.3], -1, 0, 0.25])
    a = X(X(S("Sphere", [f(0.4)]), "WithColor", f(1), f(0.2), f(0.3)), "Translate", f(-1), f(0), f(0))
    b = X(X(S("Sphere", [f(0.2)]), "WithColor", f(0.1), f(1), f(0.3)), "Translate", f(1), f(0), f(0))
    out["colored_spheres"] = (S("Union", [a, b]), ["union", ["translate", ["with_color", ["sphere", 0.4], 1, 0.2, 0.3], -1, 0, 0],
                                                    ["translate", ["with_color", ["sphere", 0.2], 0.1, 1, 0.3], 1, 0, 0]])
    out["union_box_sphere"] = (S("Union", [S("Box", [f(0.5)]), X(S("Sphere", [f(0.6)]), "Translate", CS.Vec3(0.4, 0.3, -0.2))]),
                               ["union", ["box", 0.5, 0.5, 0.5], ["translate", ["sphere", 0.6], 0.4, 0.3, -0.2]])
    out["readme_repeat_xy"] = (X(S("Sphere", [f(0.5)]), "RepeatXY", f(1.125), f(1.125), colour), ["repeat_xy_idx", ["sphere", 0.5], 1.125, 1.125])
    out["repeat_xz_box"] = (X(S("Box", [f(0.3)]), "RepeatXZ", f(1.5), f(0.875), colour), ["repeat_xz_idx", ["box", 0.3, 0.3, 0.3], 1.5, 0.875])
    return out


def points(seed, n=600):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-3.0, 3.0, size=(n, 3)).astype(np.float32)
    special = np.array([[0, 0, 0], [-0.0, 0.0, -0.0], [0.5625, 0.5625, 0], [-0.5625, 1.6875, 0.25], [0.75, -0.4375, 2.25], [1, 0, 0], [-1, 0, 0],
                        [0.5, 0.75, 0.25], [3, 3, 3], [-3, -3, -3], [1e-20, -1e-20, 1e-30]], dtype=np.float32)
    return np.concatenate([special, p])


def main():
    it = load()
    blob, names, descr = {}, [], {}
    for k, (name, (fn, d)) in enumerate(scenes(it).items()):
        pts = points(100 + k)
        out = np.zeros((len(pts), 4), dtype=np.float32)
        for i, p in enumerate(pts):
            v = fn(CS.Vec3(p[0], p[1], p[2]))
            out[i] = (v.X, v.Y, v.Z, v.W)
        names.append(name)
        blob[f"{name}/points"], blob[f"{name}/rgbw"] = pts, out
        descr[name] = d
        print(f"{name}: {len(pts)} points, w in [{out[:, 3].min():.4f}, {out[:, 3].max():.4f}]")
    import json
    blob["names"] = np.array(names)
    blob["scenes_json"] = np.array(json.dumps(descr))
    path = os.path.join(ROOT, "tests", "golden", "reference_sdf_points.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
