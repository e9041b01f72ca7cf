"""ctypes front-end of the CPU ORACLE (test infrastructure, not product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The product package `sdfkit_amd` never does.  See oracle/sdfk_oracle.h for what the
oracle restates and its pinning status (parity unpinned beyond the reference's own
known-answer tests).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# node kinds (sdfk_oracle.h)
SPHERE_W, BOX_W, PLANE_W, SDF_WITHCOLOR = 1, 2, 3, 4
F_SPHERE, F_BOX, F_CYLINDER, F_UNION, F_TRANSLATE, F_WITHCOLOR = 10, 11, 12, 13, 14, 15
F_REPEAT_X, F_REPEAT_Y, F_REPEAT_XY, F_REPEAT_XY_IDX, F_REPEAT_XZ_IDX, F_CONST = 16, 17, 18, 19, 20, 21
CF_CONST, CF_README = 0, 1


class _Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("f", C.c_float * 8)]


def build(force=False):
    so = os.path.join(_HERE, "libsdfk_oracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("sdfk_oracle.c", "sdfk_oracle_ray.c", "sdfk_oracle.h", "lewiner_luts.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libsdfk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        L.orc_eval.argtypes = [C.POINTER(_Node), C.c_int, fp, fp]
        L.orc_cell_size.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, fp]
        L.orc_sample.argtypes = [C.POINTER(_Node), C.c_int, fp, fp, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sample.restype = C.c_int
        L.orc_batch_sizes.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_batch_sizes.restype = C.c_int
        L.orc_sample_position.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int64, fp]
        L.orc_clip_to_bounds.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_march.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, fp, fp,
                                C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_march.restype = C.c_void_p
        L.orc_march_window.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_float]
        L.orc_march_window.restype = C.c_void_p
        L.orc_sample_window.argtypes = [C.POINTER(_Node), C.c_int, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p]
        L.orc_sample_window.restype = None
        L.orc_clip_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp]
        L.orc_clip_window.restype = None
        for name in ("vertex_count", "index_count", "cell_count", "impossible13"):
            f = getattr(L, "orc_mesh_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_int64
        for name in ("vertices", "colors", "normals", "grid_vertices", "grid_normals", "triangles", "cells"):
            f = getattr(L, "orc_mesh_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_void_p
        L.orc_mesh_bounds.argtypes = [C.c_void_p, fp, fp]
        L.orc_mesh_free.argtypes = [C.c_void_p]
        L.orc_resolve_tiling.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_resolve_tiling.restype = C.c_int
        L.orc_hardware_threads.restype = C.c_int
        L.orc_mat_look_at.argtypes = [fp, fp, fp, fp]
        L.orc_mat_perspective_fov.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, fp]
        L.orc_mat_mul.argtypes = [fp, fp, fp]
        L.orc_mat_invert.argtypes = [fp, fp]
        L.orc_mat_invert.restype = C.c_int
        L.orc_ray_camera.argtypes = [fp, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp]
        L.orc_raymarch.argtypes = [C.POINTER(_Node), C.c_int, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


class Scene:
    """A flat node array; the last node added is the root unless `root` is set."""

    def __init__(self):
        self.nodes = []
        self.root = -1

    def _add(self, kind, f=(), a=-1, b=-1):
        n = _Node()
        n.kind, n.a, n.b = kind, a, b
        for i, v in enumerate(f):
            n.f[i] = float(np.float32(v))
        self.nodes.append(n)
        self.root = len(self.nodes) - 1
        return self.root

    # batched Sdfs.* (write W only)
    def sphere_w(self, r): return self._add(SPHERE_W, [r])
    def box_w(self, bx, by=None, bz=None):
        by = bx if by is None else by
        bz = bx if bz is None else bz
        return self._add(BOX_W, [bx, by, bz])
    def plane_w(self, nx, ny, nz, d): return self._add(PLANE_W, [nx, ny, nz, d])
    def sdf_with_color(self, child, r, g, b): return self._add(SDF_WITHCOLOR, [r, g, b], a=child)
    # SdfFunc / SdfExpr forms
    def f_sphere(self, r, rgb=(1, 1, 1)): return self._add(F_SPHERE, [r, *rgb])
    def f_box(self, bx, by=None, bz=None):
        by = bx if by is None else by
        bz = bx if bz is None else bz
        return self._add(F_BOX, [bx, by, bz])
    def f_cylinder(self, r, h, rgb=(1, 1, 1)): return self._add(F_CYLINDER, [r, h, *rgb])
    def f_union(self, a, b): return self._add(F_UNION, a=a, b=b)
    def f_translate(self, child, x, y, z): return self._add(F_TRANSLATE, [x, y, z], a=child)
    def f_with_color(self, child, r, g, b): return self._add(F_WITHCOLOR, [r, g, b], a=child)
    def f_repeat_x(self, child, sx): return self._add(F_REPEAT_X, [sx], a=child)
    def f_repeat_y(self, child, sy): return self._add(F_REPEAT_Y, [sy], a=child)
    def f_repeat_xy(self, child, sx, sy): return self._add(F_REPEAT_XY, [sx, sy], a=child)
    def f_repeat_xy_idx(self, child, sx, sy, colorfn=CF_README, prm=(0, 0, 0)):
        return self._add(F_REPEAT_XY_IDX, [sx, sy, *prm], a=child, b=colorfn)
    def f_repeat_xz_idx(self, child, sx, sz, colorfn=CF_README, prm=(0, 0, 0)):
        return self._add(F_REPEAT_XZ_IDX, [sx, sz, *prm], a=child, b=colorfn)
    def f_const(self, r, g, b, w): return self._add(F_CONST, [r, g, b, w])

    def carray(self):
        return (_Node * len(self.nodes))(*self.nodes)


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def eval_point(scene, p):
    out = (C.c_float * 4)(0, 0, 0, 0)
    lib().orc_eval(scene.carray(), scene.root, _f3(p), out)
    return np.array(out[:], dtype=np.float32)


def cell_size(mn, mx, nx, ny, nz):
    d = (C.c_float * 3)()
    lib().orc_cell_size(_f3(mn), _f3(mx), nx, ny, nz, d)
    return np.array(d[:], dtype=np.float32)


def sample_position(mn, mx, nx, ny, nz, i):
    p = (C.c_float * 3)()
    lib().orc_sample_position(_f3(mn), _f3(mx), nx, ny, nz, i, p)
    return np.array(p[:], dtype=np.float32)


def batch_sizes(ntotal, batch):
    nb = (ntotal + batch - 1) // batch
    arr = (C.c_int * nb)()
    lib().orc_batch_sizes(ntotal, batch, arr, nb)
    return list(arr)


def sample(scene, mn, mx, nx, ny, nz, batch=2048, threads=0, with_colors=True):
    """Voxels.SampleSdf: returns (values[nx,ny,nz], colors[nx,ny,nz,3] or None)."""
    values = np.zeros((nx, ny, nz), dtype=np.float32)
    colors = np.zeros((nx, ny, nz, 3), dtype=np.float32) if with_colors else None
    lib().orc_sample(scene.carray(), scene.root, _f3(mn), _f3(mx), nx, ny, nz, batch, threads,
                     values.ctypes.data, colors.ctypes.data if with_colors else None)
    return values, colors


def clip_to_bounds(values, mn, mx):
    assert values.dtype == np.float32 and values.flags.c_contiguous
    nx, ny, nz = values.shape
    lib().orc_clip_to_bounds(values.ctypes.data, nx, ny, nz, _f3(mn), _f3(mx))
    return values


class OracleMesh:
    def __init__(self, h):
        L = lib()
        nv = L.orc_mesh_vertex_count(h)
        ni = L.orc_mesh_index_count(h)
        nc = L.orc_mesh_cell_count(h)

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros((0,), dtype=dt)
            ct = C.c_float if dt == np.float32 else C.c_int32
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).astype(dt, copy=True)

        self.vertices = arr(L.orc_mesh_vertices(h), nv * 3, np.float32).reshape(-1, 3)
        self.colors = arr(L.orc_mesh_colors(h), nv * 3, np.float32).reshape(-1, 3)
        self.normals = arr(L.orc_mesh_normals(h), nv * 3, np.float32).reshape(-1, 3)
        self.grid_vertices = arr(L.orc_mesh_grid_vertices(h), nv * 3, np.float32).reshape(-1, 3)
        self.grid_normals = arr(L.orc_mesh_grid_normals(h), nv * 3, np.float32).reshape(-1, 3)   # NegativeNormals before Transform
        self.triangles = arr(L.orc_mesh_triangles(h), ni, np.int32)
        self.cells = arr(L.orc_mesh_cells(h), nc * 4, np.int32).reshape(-1, 4)
        self.impossible13 = int(L.orc_mesh_impossible13(h))
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        L.orc_mesh_bounds(h, mn, mx)
        self.min = np.array(mn[:], dtype=np.float32)
        self.max = np.array(mx[:], dtype=np.float32)
        L.orc_mesh_free(h)

    @property
    def center(self):
        return (self.min + self.max) * np.float32(0.5)

    @property
    def size(self):
        return self.max - self.min


_PROGRESS_T = C.CFUNCTYPE(None, C.c_float, C.c_void_p)


def march(values, colors, mn, mx, iso=0.0, step=1, progress=None):
    """MarchingCubes.CreateMesh on [nx,ny,nz] float32 values (+ optional [nx,ny,nz,3] colours)."""
    values = np.ascontiguousarray(values, dtype=np.float32)
    nx, ny, nz = values.shape
    cp = None
    if colors is not None:
        colors = np.ascontiguousarray(colors, dtype=np.float32)
        cp = colors.ctypes.data
    cb = _PROGRESS_T(lambda v, u: progress(v)) if progress else None
    h = lib().orc_march(values.ctypes.data, cp, nx, ny, nz, _f3(mn), _f3(mx),
                        C.c_float(iso), step, C.cast(cb, C.c_void_p) if cb else None, None)
    return OracleMesh(h)


def sample_window(scene, mn, mx, nx, ny, nz, z0, nzw, threads=0, with_colors=True, clip=False):
    """Voxels.SampleSdf (+ ClipToBounds) on the planes [z0, z0 + nzw) of the nx*ny*nz grid: ([nx, ny, nzw], colours)."""
    values = np.zeros((nx, ny, nzw), dtype=np.float32)
    colors = np.zeros((nx, ny, nzw, 3), dtype=np.float32) if with_colors else None
    lib().orc_sample_window(scene.carray(), scene.root, _f3(mn), _f3(mx), nx, ny, nz, z0, nzw, threads,
                            values.ctypes.data, colors.ctypes.data if with_colors else None)
    if clip:
        lib().orc_clip_window(values.ctypes.data, nx, ny, nz, z0, nzw, _f3(mn), _f3(mx))
    return values, colors


def march_window(values, colors, z0, nz_global, mn, mx, iso=0.0):
    """The serial sweep on a window of planes [z0, z0 + values.shape[2]) of a grid of nz_global planes (global z
    coordinates, the whole grid's transform).  cells[:, 0] holds window-local linear cell indices."""
    values = np.ascontiguousarray(values, dtype=np.float32)
    nx, ny, nzw = values.shape
    cp = None
    if colors is not None:
        colors = np.ascontiguousarray(colors, dtype=np.float32)
        cp = colors.ctypes.data
    return OracleMesh(lib().orc_march_window(values.ctypes.data, cp, nx, ny, nzw, z0, nz_global, _f3(mn), _f3(mx), C.c_float(iso)))


def window_part(wm, nx, ny, z0, lb, le):
    """The part of a window mesh `wm` (march_window of planes starting at z0 <= lb - 2, or z0 = 0) that belongs to the cell
    layers [lb, le): (vertices, colors, normals, triangles with ids local to the part -- seam references are negative).
    This is what sdfk_march_slab(.., lb, le, vertex_base = 0) returns for the same layers."""
    assert z0 == 0 or z0 <= lb - 2
    cz = wm.cells[:, 0] // (nx * ny) + z0           # global layer of every active cell, sweep order
    nt = wm.cells[:, 3]
    tri_start = np.concatenate([[0], np.cumsum(nt * 3)])
    i0, i1 = np.searchsorted(cz, lb, "left"), np.searchsorted(cz, le, "left")
    t0, t1 = int(tri_start[i0]), int(tri_start[i1])
    first = np.full(len(wm.vertices), len(wm.triangles), np.int64)   # vertices are numbered in order of first reference
    np.minimum.at(first, wm.triangles, np.arange(len(wm.triangles)))
    v0, v1 = int(np.count_nonzero(first < t0)), int(np.count_nonzero(first < t1))
    tri = (wm.triangles[t0:t1].astype(np.int64) - v0).astype(np.int32)
    return wm.vertices[v0:v1], wm.colors[v0:v1], wm.normals[v0:v1], tri


def transform(vertices, normals, matrix):
    """Mesh.Transform (Mesh.cs:47-64) on copies of [n,3] float32 arrays: (vertices, normals, min, max)."""
    v = np.array(vertices, dtype=np.float32, order="C")
    q = np.array(normals, dtype=np.float32, order="C")
    m = (C.c_float * 16)(*[float(np.float32(x)) for x in np.asarray(matrix, np.float32).ravel()])
    lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
    L = lib()
    L.orc_transform_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_transform_arrays(v.ctypes.data, q.ctypes.data, len(v), m, lo, hi)
    return v, q, np.array(lo[:], np.float32), np.array(hi[:], np.float32)


def resolve_tiling(v8):
    v = (C.c_double * 8)(*[float(x) for x in v8])
    off, nt = C.c_int(), C.c_int()
    idx = lib().orc_resolve_tiling(v, C.byref(off), C.byref(nt))
    return idx, off.value, nt.value


def hardware_threads():
    return lib().orc_hardware_threads()


# ---- RayMarcher (RayMarcher.cs:45-211) -------------------------------------------------
def _f16(m):
    return (C.c_float * 16)(*[float(np.float32(x)) for x in np.asarray(m, np.float32).ravel()])


def look_at(pos, target, up):
    """Matrix4x4.CreateLookAt, row-major 4x4 float32."""
    out = (C.c_float * 16)()
    lib().orc_mat_look_at(_f3(pos), _f3(target), _f3(up), out)
    return np.array(out[:], np.float32).reshape(4, 4)


def ray_camera(view, fov_degrees, width, height, near, far):
    """Host part of RayMarcher.GetCameraRays: (camera position, inverse view-projection)."""
    cam, vpi = (C.c_float * 3)(), (C.c_float * 16)()
    lib().orc_ray_camera(_f16(view), fov_degrees, width, height, near, far, cam, vpi)
    return np.array(cam[:], np.float32), np.array(vpi[:], np.float32).reshape(4, 4)


def raymarch(scene, width, height, view=None, fov_degrees=60.0, near=1.0, far=100.0, iterations=40,
             want_depth=True, want_rgb=True, threads=0):
    """RayMarcher.RenderDepth / Render.  Returns (depth [h, w] or None, rgb [h, w, 3] or None)."""
    if view is None:
        view = look_at((0, 0, 5), (0, 0, 0), (0, 1, 0))   # RayMarcher.cs:22-23
    cam, vpi = ray_camera(view, fov_degrees, width, height, near, far)
    depth = np.empty((height, width), np.float32) if want_depth else None
    rgb = np.empty((height, width, 3), np.float32) if want_rgb else None
    lib().orc_raymarch(scene.carray(), scene.root, width, height, _f3(cam), _f16(vpi), near, far, iterations,
                       depth.ctypes.data if want_depth else None, rgb.ctypes.data if want_rgb else None, threads)
    return depth, rgb
