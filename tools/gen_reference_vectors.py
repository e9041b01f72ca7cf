#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's own source (build container only: reads /root/reference).

MarchingCubes.CreateMesh (SdfKit/MarchingCubes.cs:39-92, with TheBigSwitch / TestFace / TestInternal) and all of Cell
(SdfKit/Cell.cs) are parsed and run by tools/cs_subset.py -- a tree-walking interpreter of the C# subset those two files are
written in -- on small seeded random volumes; Luts.cs supplies the tables (parsed as data by tools/gen_luts.py).  What is
recorded is what `new Mesh(cell.Vertices, cell.Colors, cell.NegativeNormals, cell.Faces)` receives (MarchingCubes.cs:84),
i.e. everything up to Mesh.Transform (which is BCL Matrix4x4 arithmetic, not reference code).

Output, committed: tests/golden/reference_meshes.npz -- per case the inputs (values, colours, iso, step) and the outputs
(vertices in voxel-index units, colours, negative normals, faces, number of "Impossible case 13?" messages).
tests/test_reference_vectors.py holds the oracle to them, bit for bit.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cs_subset as CS   # noqa: E402
import gen_luts          # noqa: E402

REF = "/root/reference/SdfKit"
F32 = np.float32


class Grid3:
    def __init__(self, a, wrap):
        self.a, self.wrap = a, wrap

    def __getitem__(self, idx):
        return self.wrap(self.a[idx])


class Volume:
    def __init__(self, values, colors):
        self.NX, self.NY, self.NZ = (int(n) for n in values.shape)
        self.Values = Grid3(values, lambda v: F32(v))
        self.Colors = Grid3(colors, lambda c: CS.Vec3(c[0], c[1], c[2]))
        self.Size = CS.Opaque()
        self.Center = CS.Opaque()


class MeshCapture:
    last = None

    def __init__(self, vertices, colors, normals, faces):
        self.vertices, self.colors, self.normals, self.faces = vertices, colors, normals, faces
        MeshCapture.last = self

    def Transform(self, m):
        return None


class Vector3Host:
    Zero = CS.Vec3()

    @staticmethod
    def Normalize(v):   # BCL: value / value.Length(), Length = sqrt((x x + y y) + z z) in float32 (see cs_subset.py)
        with np.errstate(all="ignore"):
            ls = F32(F32(F32(v.X * v.X) + F32(v.Y * v.Y)) + F32(v.Z * v.Z))
            ln = F32(np.sqrt(ls))
            return CS.Vec3(F32(v.X / ln), F32(v.Y / ln), F32(v.Z / ln))


class MathHost:
    @staticmethod
    def Abs(x):
        return abs(x)


class ConsoleHost:
    def __init__(self):
        self.lines = []

    def WriteLine(self, *a):
        self.lines.append(a)


class OpaqueCallable(CS.Opaque):
    def __call__(self, *a):
        return self


def load():
    classes = {}
    for f in ("MarchingCubes.cs", "Cell.cs"):
        classes.update(CS.parse(open(os.path.join(REF, f)).read()))
    luts = type("LutsHost", (), {})()
    for name, shape, vals in gen_luts.parse_tables(open(os.path.join(REF, "Luts.cs")).read()):
        setattr(luts, name, np.array(vals, dtype=np.int8).reshape(shape))
    return classes, luts


def create_mesh(classes, luts, values, colors, iso, step):
    console = ConsoleHost()
    hosts = {"Luts": luts, "Math": MathHost, "Console": console, "Vector3": Vector3Host, "Matrix4x4": OpaqueCallable(), "Mesh": MeshCapture}
    it = CS.Interp(classes, hosts)
    MeshCapture.last = None
    it.call_static("MarchingCubes", "CreateMesh", [Volume(values, colors), F32(iso), int(step), None])
    m = MeshCapture.last
    v3 = lambda lst: np.array([[q.X, q.Y, q.Z] for q in lst], dtype=np.float32).reshape(-1, 3)
    return v3(m.vertices), v3(m.colors), v3(m.normals), np.array(m.faces, dtype=np.int32), len(console.lines)


def cases():
    """(name, values, colours, iso, step): uniform random fields reach every case, sub-tiling and the centre vertex within a few
    hundred cells; a smooth field gives shared vertices with many contributions per normal; step 2 and iso != 0 take the
    other parameter paths."""
    out = []
    for seed, shape, iso, step in ((0, (7, 6, 5), 0.0, 1), (1, (6, 7, 8), 0.0, 1), (2, (9, 5, 6), 0.25, 1), (3, (5, 5, 5), -0.125, 1),
                                   (4, (9, 8, 9), 0.0, 2), (5, (8, 9, 7), 0.0, 1), (6, (10, 10, 10), 0.0, 1)):
        rng = np.random.default_rng(seed)
        values = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
        colors = rng.uniform(0.0, 1.0, size=shape + (3,)).astype(np.float32)
        out.append((f"random{seed}", values, colors, iso, step))
    for seed in (10, 11):   # 17^3: every case, sub-tiling, centre vertex and interior test many times over
        rng = np.random.default_rng(seed)
        out.append((f"random17_{seed}", rng.uniform(-1.0, 1.0, size=(17, 17, 17)).astype(np.float32),
                    rng.uniform(0.0, 1.0, size=(17, 17, 17, 3)).astype(np.float32), 0.0, 1))
    # cells that resolve to NO tiling (case 13 with a face-test pattern the sub-configuration table maps to -1: the reference
    # prints a message and emits nothing, MarchingCubes.cs:362-366): at the volume corner, inside, on the far faces, both signs
    dead = np.array([1e-4, -2e-4, 1.5, -1.0, -2e-4, 1e-4, -1.0, 1.5], np.float32)
    corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    rng = np.random.default_rng(7)
    v = rng.uniform(-1, 1, (11, 10, 9)).astype(np.float32)
    c = rng.uniform(0, 1, v.shape + (3,)).astype(np.float32)
    for k, (x, y, z) in enumerate([(0, 0, 0), (5, 5, 5), (9, 8, 7), (0, 6, 3), (8, 0, 4), (3, 3, 0)]):
        for q, (dx, dy, dz) in enumerate(corner):
            v[x + dx, y + dy, z + dz] = (-1.0 if k in (1, 4) else 1.0) * dead[q]
    out.append(("dead13", v, c, 0.0, 1))
    n = 12
    g = (np.arange(n, dtype=np.float32) - F32(5.3)) * F32(0.31)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    sphere = (np.sqrt(x * x + y * y + z * z) - F32(1.37)).astype(np.float32)
    colors = np.stack([np.abs(np.sin(x)), np.abs(np.cos(y)), np.abs(np.sin(z + 1))], axis=-1).astype(np.float32)
    out.append(("sphere12", sphere, colors, 0.0, 1))
    out.append(("sphere12_step3", sphere, colors, 0.0, 3))
    return out


def main():
    classes, luts = load()
    blob, names = {}, []
    for name, values, colors, iso, step in cases():
        v, c, n, f, impossible = create_mesh(classes, luts, values, colors, iso, step)
        print(f"{name}: grid {values.shape} iso {iso} step {step} -> {len(v)} vertices, {len(f) // 3} triangles, {impossible} console lines")
        names.append(name)
        blob[f"{name}/values"], blob[f"{name}/colors"] = values, colors
        blob[f"{name}/iso_step"] = np.array([iso, step], dtype=np.float64)
        blob[f"{name}/vertices"], blob[f"{name}/out_colors"], blob[f"{name}/normals"], blob[f"{name}/faces"] = v, c, n, f
        blob[f"{name}/console_lines"] = np.array([impossible], dtype=np.int64)
    blob["names"] = np.array(names)
    out = os.path.join(ROOT, "tests", "golden", "reference_meshes.npz")
    np.savez_compressed(out, **blob)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
