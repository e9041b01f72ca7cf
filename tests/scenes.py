"""Scene catalogue used by the parity tests: each entry builds the SAME scene twice --
as an oracle scene graph (oracle/oracle.py, restating the reference's C#) and as a product
SDF program (sdfkit_amd.api, lowered to the GPU).  Inputs follow the reference's own tests
(Tests/MarchingCubesTests.cs, Tests/SdfTests.cs, README.md:24-30) and BASELINE.md's configs."""
import numpy as np

from oracle import oracle as O
from sdfkit_amd import SdfExprs, SdfFuncs, Sdfs, Vec3


def _readme_color(i, p, d):
    # (i, p, d) => 0.9f*Vector3.One - Vector3.Abs(i)/6f     README.md:24-30
    return 0.9 * Vec3.of(p.x.b, 1.0) - Vec3.Abs(i) / 6.0


def sphere_w(r):
    s = O.Scene(); s.sphere_w(r)
    return s, Sdfs.Sphere(r)


def box_w(b):
    s = O.Scene(); s.box_w(b)
    return s, Sdfs.Box(b)


def plane_w(n, d):
    s = O.Scene(); s.plane_w(n[0], n[1], n[2], d)
    return s, Sdfs.Plane(n, d)


def cylinder(r, h):
    s = O.Scene(); s.f_cylinder(r, h)
    return s, Sdfs.Cylinder(r, h)


def solid_sphere(r):
    s = O.Scene(); s.f_sphere(r)
    return s, SdfExprs.Solid(lambda p: p.Length() - np.float32(r)).ToSdf()


def colored_spheres():
    # MarchingCubesTests.cs:11-28
    s = O.Scene()
    a = s.f_translate(s.f_with_color(s.f_sphere(0.4), 1.0, 0.2, 0.3), -1, 0, 0)
    b = s.f_translate(s.f_with_color(s.f_sphere(0.2), 0.1, 1.0, 0.3), 1, 0, 0)
    s.f_union(a, b)
    p = SdfFuncs.Union(SdfFuncs.Sphere(0.4).WithColor(1.0, 0.2, 0.3).Translate(-1, 0, 0),
                       SdfFuncs.Sphere(0.2).WithColor(0.1, 1.0, 0.3).Translate(1, 0, 0)).ToSdf()
    return s, p


def readme_repeat_xy():
    # BASELINE C3: SdfExprs.Sphere(0.5f).RepeatXY(1.125f, 1.125f, (i,p,d) => ...)
    s = O.Scene()
    s.f_repeat_xy_idx(s.f_sphere(0.5), 1.125, 1.125, O.CF_README)
    p = SdfExprs.Sphere(0.5).RepeatXY(1.125, 1.125, _readme_color).ToSdf()
    return s, p


def repeat_xz_box():
    s = O.Scene()
    s.f_repeat_xz_idx(s.f_box(0.3), 1.0, 1.25, O.CF_README)
    p = SdfExprs.Box(0.3).RepeatXZ(1.0, 1.25, _readme_color).ToSdf()
    return s, p


def repeat_x_y_plain():
    s = O.Scene()
    s.f_repeat_y(s.f_repeat_x(s.f_sphere(0.3, (0.2, 0.4, 0.6)), 0.9), 1.1)
    p = SdfExprs.Sphere(0.3, (0.2, 0.4, 0.6)).RepeatX(0.9).RepeatY(1.1).ToSdf()
    return s, p


def repeat_xy_plain():
    s = O.Scene()
    s.f_repeat_xy(s.f_cylinder(0.25, 0.4, (0.5, 0.25, 1.0)), 1.0, 0.8)
    p = SdfExprs.Cylinder(0.25, 0.4, (0.5, 0.25, 1.0)).RepeatXY(1.0, 0.8).ToSdf()
    return s, p


def union8():
    # BASELINE C4: nested Union of 8 primitives at the octant centres of [-2,2]^3
    s = O.Scene()
    nodes, prods = [], []
    k = 0
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                kind = k % 3
                if kind == 0:
                    n = s.f_sphere(0.6); q = SdfExprs.Sphere(0.6)
                elif kind == 1:
                    n = s.f_box(0.5); q = SdfExprs.Box(0.5)
                else:
                    n = s.f_cylinder(0.4, 0.6); q = SdfExprs.Cylinder(0.4, 0.6)
                nodes.append(s.f_translate(n, sx, sy, sz))
                prods.append(q.Translate(sx, sy, sz))
                k += 1
    root, prod = nodes[0], prods[0]
    for n, q in zip(nodes[1:], prods[1:]):
        root = s.f_union(root, n)
        prod = SdfExprs.Union(prod, q)
    s.root = root
    return s, prod.ToSdf()


def sdf_with_color():
    s = O.Scene()
    s.sdf_with_color(s.box_w(0.7, 0.5, 0.3), 0.25, 0.5, 0.75)
    return s, Sdfs.Box((0.7, 0.5, 0.3)).WithColor(0.25, 0.5, 0.75)


CATALOGUE = {
    "sphere_w": lambda: sphere_w(1.0),
    "box_w": lambda: box_w(0.8),
    "plane_w": lambda: plane_w((0.3, 0.5, 0.8), 0.1),
    "cylinder": lambda: cylinder(0.6, 0.9),
    "solid_sphere": lambda: solid_sphere(0.5),
    "colored_spheres": colored_spheres,
    "readme_repeat_xy": readme_repeat_xy,
    "repeat_xz_box": repeat_xz_box,
    "repeat_x_y_plain": repeat_x_y_plain,
    "repeat_xy_plain": repeat_xy_plain,
    "union8": union8,
    "sdf_with_color": sdf_with_color,
}


def random_scene(seed, depth=3):
    """A random composition of the per-point catalogue (SdfExprs primitives, Translate, RepeatX /
    RepeatY / RepeatXY, RepeatXY with the README colour lambda, WithColor, Union), built twice --
    oracle graph and product program -- from the same draws.  Parameters are dyadic-ish float32
    values so that both sides see exactly the same constants."""
    rng = np.random.default_rng(seed)
    s = O.Scene()

    def f(lo, hi):
        return float(np.float32(rng.integers(int(lo * 16), int(hi * 16) + 1) / 16.0))

    def prim():
        k = int(rng.integers(0, 3))
        col = (f(0, 1), f(0, 1), f(0, 1))
        if k == 0:
            r = f(0.25, 0.75)
            return s.f_sphere(r, col), SdfExprs.Sphere(r, col)
        if k == 1:
            b = f(0.25, 0.625)
            return s.f_box(b), SdfExprs.Box(b)
        r, h = f(0.25, 0.5), f(0.25, 0.75)
        return s.f_cylinder(r, h, col), SdfExprs.Cylinder(r, h, col)

    def node(d):
        if d == 0:
            return prim()
        k = int(rng.integers(0, 7))
        if k == 0:
            a, pa = node(d - 1)
            b, pb = node(d - 1)
            return s.f_union(a, b), SdfExprs.Union(pa, pb)
        c, pc = node(d - 1)
        if k == 1:
            t = (f(-1.5, 1.5), f(-1.5, 1.5), f(-1.5, 1.5))
            return s.f_translate(c, *t), pc.Translate(*t)
        if k == 2:
            sx = f(0.75, 2.0)
            return s.f_repeat_x(c, sx), pc.RepeatX(sx)
        if k == 3:
            sy = f(0.75, 2.0)
            return s.f_repeat_y(c, sy), pc.RepeatY(sy)
        if k == 4:
            sx, sy = f(0.75, 2.0), f(0.75, 2.0)
            return s.f_repeat_xy(c, sx, sy), pc.RepeatXY(sx, sy)
        if k == 5:
            sx, sy = f(0.75, 2.0), f(0.75, 2.0)
            return s.f_repeat_xy_idx(c, sx, sy, O.CF_README), pc.RepeatXY(sx, sy, _readme_color)
        col = (f(0, 1), f(0, 1), f(0, 1))
        return s.f_with_color(c, *col), pc.WithColor(*col)

    root, prod = node(depth)
    s.root = root
    return s, prod.ToSdf()
