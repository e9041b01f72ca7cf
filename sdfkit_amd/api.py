"""Host-side mirror of SdfKit's public API for the hot path, over the C ABI.

Same names, argument meaning and error behaviour as the reference (file:line cited per
member) so that tests read like the reference's own: `Sdfs`, `SdfFuncs`, `SdfExprs`
(+ the `SdfEx`/`SdfFuncEx`/`SdfExprEx` extension methods as ordinary methods),
`Voxels`, `MarchingCubes`, `Mesh`.  All compute happens in libsdfkit_hip.so on the GPU;
this module only builds SDF programs and moves arrays.  The C# shim a maintainer would
write is the same thing with `[DllImport]`.
"""
import ctypes as C
import math

import numpy as np

from . import _native as N
from .expr import MathF, Mod, VMax, Vec3, Vec4, select_lt, trace

DefaultBatchSize = 2 * 1024  # SdfConfig.DefaultBatchSize (Sdf.cs:13); no meaning on the GPU


def _v3(v):
    if np.isscalar(v):
        v = (v, v, v)
    return np.array([np.float32(v[0]), np.float32(v[1]), np.float32(v[2])], dtype=np.float32)


# ---------------------------------------------------------------------------
# Sdf: the reference's `Sdf` delegate (Sdf.cs:8), restricted to GPU-lowerable SDFs
# ---------------------------------------------------------------------------
class Sdf:
    """A batched SDF.  Wraps a per-point symbolic function `fn(Vec3) -> Vec4` and whether
    it writes colour (delegates that only assign `.W` leave colour (0,0,0), Voxels.cs:88-92)."""

    def __init__(self, fn, writes_color=True):
        self.fn = fn
        self.writes_color = bool(writes_color)
        self._prog = None
        self._ir = None

    # -- lowering -----------------------------------------------------------
    def ir(self):
        if self._ir is None:
            ops, out = trace(self.fn, self.writes_color)
            arr = (N.Op * len(ops))()
            for i, (op, a, b, c, d, imm) in enumerate(ops):
                arr[i].opcode, arr[i].a, arr[i].b, arr[i].c, arr[i].d, arr[i].imm = op, a, b, c, d, imm
            self._ir = (arr, len(ops), (C.c_int32 * 4)(*out))
        return self._ir

    def check(self):
        """Generate + compile the sampling kernel for gfx950 (no device needed)."""
        arr, n, out = self.ir()
        N.check(N.lib().sdfk_program_check(arr, n, out, int(self.writes_color)))

    def program(self):
        if self._prog is None:
            N.init()
            arr, n, out = self.ir()
            h = C.c_void_p()
            N.check(N.lib().sdfk_program_create(arr, n, out, int(self.writes_color), C.byref(h)))
            self._prog = h
        return self._prog

    def source(self):
        return N.lib().sdfk_program_source(self.program()).decode()

    def __del__(self):
        try:
            if self._prog is not None and N._lib is not None and N._inited_device is not None:
                N._lib.sdfk_program_destroy(self._prog)
        except Exception:
            pass

    # -- SdfEx extension methods (Sdf.cs:20-116) ------------------------------
    def WithColor(self, *color):
        """SdfEx.WithColor (Sdf.cs:101-115)."""
        col = color[0] if len(color) == 1 else color
        inner = self.fn
        return Sdf(lambda p: Vec4.of(col, inner(p).w), True)

    def ToVoxels(self, min, max, nx, ny, nz, batchSize=DefaultBatchSize, maxDegreeOfParallelism=-1,
                 clipToBounds=True):
        """SdfEx.ToVoxels (Sdf.cs:49-57); ClipToBounds is fused into the sampling kernel."""
        v = Voxels(min, max, nx, ny, nz)
        v._sample(self, clip=clipToBounds)
        return v

    def Sample(self, points, colorsAndDistances=None, batchSize=DefaultBatchSize, maxDegreeOfParallelism=-1):
        """SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points.  points [n, 3] float32; returns (and, when given, fills)
        colorsAndDistances [n, 4] = (r, g, b, distance) per point.  Like the reference's delegates, an SDF that only assigns .W
        leaves X, Y, Z of the caller's elements untouched (zeros in a fresh array).  batchSize / maxDegreeOfParallelism are
        accepted and ignored (one lane per point on the GPU)."""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        out = colorsAndDistances
        if out is None:
            out = np.zeros((len(pts), 4), np.float32)
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.shape != (len(pts), 4):
            raise ValueError("colorsAndDistances must be a C-contiguous float32 array of shape [n, 4]")
        N.check(N.lib().sdfk_eval_points(self.program(), pts.ctypes.data, len(pts), out.ctypes.data))
        return out

    def ToImage(self, width, height, *camera, **kw):
        """SdfEx.ToImage (Sdf.cs:65-99): camera = viewTransform, or position, target, up."""
        from .raymarch import to_image
        return to_image(self, width, height, *camera, **kw)

    def ToMesh(self, min, max, nx, ny, nz, batchSize=DefaultBatchSize, maxDegreeOfParallelism=-1,
               clipToBounds=True, isoValue=0.0, step=1, progress=None):
        """SdfEx.ToMesh (Sdf.cs:59-63): device-resident sample -> mesh.  In a process that has joined a sharding context
        (sdfkit_amd.dist.init: one process per GPU) the SAME call cuts the grid into Z slabs over the GPUs of the node and
        every rank gets the whole mesh -- collective: every rank makes the call (sdfk_dist_to_mesh)."""
        h = C.c_void_p()
        if step == 1:
            w, r, b = C.c_int32(), C.c_int32(), C.c_int32()
            N.check(N.lib().sdfk_dist_info(C.byref(w), C.byref(r), C.byref(b)))
            if b.value and w.value > 1:
                N.check(N.lib().sdfk_dist_to_mesh(self.program(), N.f3(min), N.f3(max), nx, ny, nz, 1 if clipToBounds else 0,
                                                  C.c_float(isoValue), C.byref(h)))
                _report_progress(progress, nz, step)
                return Mesh._from_handle(h)
        N.check(N.lib().sdfk_sample_march(self.program(), N.f3(min), N.f3(max), nx, ny, nz,
                                          1 if clipToBounds else 0, C.c_float(isoValue), step, C.byref(h)))
        _report_progress(progress, nz, step)
        return Mesh._from_handle(h)


def _report_progress(progress, nz, step):
    """IProgress<float> contract of MarchingCubes.cs:53-81: one report per z layer,
    (float)z / (nz - 2*step)."""
    if progress is None:
        return
    zb = nz - 2 * step
    z = -step
    while z < zb:
        z += step
        progress(float(np.float32(z) / np.float32(zb)) if zb != 0 else float("nan"))


# ---------------------------------------------------------------------------
# per-point SDF functions: SdfFuncs / SdfExprs share semantics (Sdf.cs:217-341, SdfExpr.cs)
# ---------------------------------------------------------------------------
class SdfFunc:
    """Per-point SDF (the reference's SdfFunc delegate / SdfExpr tree)."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, p):
        return self.fn(p)

    # SdfFuncEx / SdfExprEx members
    def ToSdf(self):
        """SdfFuncEx.ToSdf (Sdf.cs:301-313) / SdfExprEx.ToSdf (SdfExpr.cs:208-211)."""
        return Sdf(self.fn, True)

    def Translate(self, *offset):
        """SdfFuncEx.Translate (Sdf.cs:315-326): sdf(p - offset)."""
        off = offset[0] if len(offset) == 1 else offset
        f = self.fn
        return SdfFunc(lambda p: f(p - Vec3.of(p.x.b, off)))

    def WithColor(self, *color):
        """SdfFuncEx.WithColor (Sdf.cs:328-340)."""
        col = color[0] if len(color) == 1 else color
        f = self.fn
        return SdfFunc(lambda p: Vec4.of(col, f(p).w))

    def Color(self, *color):
        """SdfExprEx.Color (SdfExpr.cs:143-147)."""
        return self.WithColor(*color)

    def ModifyInput(self, change_position):
        """SdfExprEx.ModifyInput (SdfExpr.cs:79-89)."""
        f = self.fn
        return SdfFunc(lambda p: f(change_position(p)))

    def ModifyOutput(self, mod):
        """SdfExprEx.ModifyOutput (SdfExpr.cs:91-111): colour = mod(p, d), distance kept."""
        f = self.fn

        def g(p):
            d = f(p)
            return Vec4.of(mod(p, d), d.w)
        return SdfFunc(g)

    def ModifyInputAndOutput(self, mod_input, mod_output):
        """SdfExprEx.ModifyInputAndOutput (SdfExpr.cs:113-141; Sdf.cs:253-265):
        mod_input(p) -> (position, index); colour = mod_output(index, position, d)."""
        f = self.fn

        def g(p):
            mp, index = mod_input(p)
            d = f(mp)
            return Vec4.of(mod_output(index, mp, d), d.w)
        return SdfFunc(g)

    def RepeatX(self, sizeX):
        """SdfExprEx.RepeatX (SdfExpr.cs:149-153)."""
        sx = np.float32(sizeX)
        return self.ModifyInput(lambda p: Vec3(SdfFunc._rep2(p.x, sx), p.y, p.z))

    def RepeatY(self, sizeY):
        """SdfExprEx.RepeatY (SdfExpr.cs:197-201)."""
        sy = np.float32(sizeY)
        return self.ModifyInput(lambda p: Vec3(p.x, SdfFunc._rep2(p.y, sy), p.z))

    @staticmethod
    def _rep2(c, s):
        # Mod((p.X + sizeX * 0.5f), sizeX) - sizeX * 0.5f, every float op recorded
        b = c.b
        sv = b.const(s)
        half = sv * b.const(0.5)
        return Mod(c + half, sv) - sv * b.const(0.5)

    @staticmethod
    def _idx(c, s):
        # MathF.Floor((p.X + sizeX * 0.5f) / sizeX)
        b = c.b
        sv = b.const(s)
        return MathF.Floor((c + sv * b.const(0.5)) / sv)

    def RepeatXY(self, sizeX, sizeY, mod=None):
        """SdfExprEx.RepeatXY (SdfExpr.cs:155-178) / SdfFuncEx.RepeatXY (Sdf.cs:267-282)."""
        sx, sy = np.float32(sizeX), np.float32(sizeY)
        if mod is None:
            return self.ModifyInput(lambda p: Vec3(SdfFunc._rep2(p.x, sx), SdfFunc._rep2(p.y, sy), p.z))
        return self.ModifyInputAndOutput(
            lambda p: (Vec3(SdfFunc._rep2(p.x, sx), SdfFunc._rep2(p.y, sy), p.z),
                       Vec3(SdfFunc._idx(p.x, sx), SdfFunc._idx(p.y, sy), p.x.b.const(0.0))),
            mod)

    def RepeatXZ(self, sizeX, sizeZ, mod):
        """SdfExprEx.RepeatXZ (SdfExpr.cs:180-195) / SdfFuncEx.RepeatXZ (Sdf.cs:284-299)."""
        sx, sz = np.float32(sizeX), np.float32(sizeZ)
        return self.ModifyInputAndOutput(
            lambda p: (Vec3(SdfFunc._rep2(p.x, sx), p.y, SdfFunc._rep2(p.z, sz)),
                       Vec3(SdfFunc._idx(p.x, sx), p.x.b.const(0.0), SdfFunc._idx(p.z, sz))),
            mod)


def _box_distance(p, bounds):
    # Vector3.Max(wd, Zero).Length() + VMax(Vector3.Min(wd, Zero))   (Sdf.cs:134-136)
    wd = Vec3.Abs(p) - Vec3.of(p.x.b, bounds)
    return Vec3.Max(wd, 0.0).Length() + VMax(Vec3.Min(wd, 0.0))


class SdfFuncs:
    """Sdf.cs:217-249"""

    @staticmethod
    def Box(bounds):
        return SdfFunc(lambda p: Vec4.of(1.0, _box_distance(p, bounds)))

    @staticmethod
    def Sphere(radius):
        return SdfFunc(lambda p: Vec4.of(1.0, p.Length() - np.float32(radius)))

    @staticmethod
    def Union(a, b):
        def g(p):
            da, db = a(p), b(p)
            pick = lambda u, v: select_lt(da.w, db.w, u, v)  # da.W < db.W ? da : db
            return Vec4(pick(da.x, db.x), pick(da.y, db.y), pick(da.z, db.z), pick(da.w, db.w))
        return SdfFunc(g)


class SdfExprs:
    """SdfExpr.cs:16-69"""

    @staticmethod
    def Box(bounds):
        return SdfFuncs.Box(bounds)

    @staticmethod
    def Cylinder(r, h, color=(1.0, 1.0, 1.0)):
        r, h = np.float32(r), np.float32(h)
        return SdfFunc(lambda p: Vec4.of(
            color, MathF.Max(MathF.Sqrt(p.x * p.x + p.z * p.z) - r, MathF.Abs(p.y) - h)))

    @staticmethod
    def Solid(dist, color=(1.0, 1.0, 1.0)):
        """SdfExprs.Solid(SdfDistExpr[, color]) (SdfExpr.cs:35-45): `dist(p)` symbolic."""
        return SdfFunc(lambda p: Vec4.of(color, dist(p)))

    @staticmethod
    def Sphere(r, color=(1.0, 1.0, 1.0)):
        return SdfFunc(lambda p: Vec4.of(color, p.Length() - np.float32(r)))

    @staticmethod
    def Union(a, b):
        return SdfFuncs.Union(a, b)


class Sdfs:
    """Batched catalogue, Sdf.cs:118-215."""

    @staticmethod
    def Box(bounds):
        return Sdf(lambda p: Vec4(None, None, None, _box_distance(p, bounds)), writes_color=False)

    @staticmethod
    def Cylinder(radius, height):
        return SdfExprs.Cylinder(radius, height).ToSdf()

    @staticmethod
    def Plane(normal, distanceFromOrigin):
        return Sdf(lambda p: Vec4(None, None, None, Vec3.Dot(p, normal) + np.float32(distanceFromOrigin)),
                   writes_color=False)

    @staticmethod
    def PlaneXY(z=0.0):
        return Sdfs.Plane((0.0, 0.0, 1.0), z)

    @staticmethod
    def PlaneXZ(y=0.0):
        return Sdfs.Plane((0.0, 1.0, 0.0), y)

    @staticmethod
    def Solid(sdf, color=None):
        """Sdfs.Solid(SdfFunc) / Sdfs.Solid(SdfDistFunc[, color]) (Sdf.cs:172-200)."""
        if isinstance(sdf, SdfFunc):
            return Sdf(sdf.fn, True)
        col = (1.0, 1.0, 1.0) if color is None else color
        return Sdf(lambda p: Vec4.of(col, sdf(p)), True)

    @staticmethod
    def Sphere(radius):
        return Sdf(lambda p: Vec4(None, None, None, p.Length() - np.float32(radius)), writes_color=False)


# ---------------------------------------------------------------------------
# Voxels (Voxels.cs)
# ---------------------------------------------------------------------------
class _InstanceOrStatic:
    """`obj.f(...)` -> instance form, `Class.f(...)` -> static form (C# overload pair)."""

    def __init__(self, inst, static):
        self.inst = inst
        self.static = static.__func__ if isinstance(static, staticmethod) else static

    def __get__(self, obj, owner):
        if obj is None:
            return self.static
        return lambda *a, **k: self.inst(obj, *a, **k)


class Voxels:
    """A regular 3-D grid of distance values, resident on the GPU; `Values` / `Colors`
    materialise host copies laid out like the reference's `float[nx,ny,nz]` /
    `Vector3[nx,ny,nz]` (Voxels.cs:8-9)."""

    def __init__(self, *args):
        # Voxels(min, max, nx, ny, nz)  (Voxels.cs:37-40)  or  Voxels(values, colors, min, max) (:23-35)
        if len(args) == 5:
            mn, mx, nx, ny, nz = args
            values = colors = None
        elif len(args) == 4:
            values, colors, mn, mx = args
            values = np.ascontiguousarray(values, dtype=np.float32)
            nx, ny, nz = values.shape
            if colors is not None:
                colors = np.ascontiguousarray(colors, dtype=np.float32).reshape(nx, ny, nz, 3)
        else:
            raise TypeError("Voxels(min, max, nx, ny, nz) or Voxels(values, colors, min, max)")
        self.NX, self.NY, self.NZ = int(nx), int(ny), int(nz)
        self.Min, self.Max = _v3(mn), _v3(mx)
        ext = self.Max - self.Min
        self.DX = np.float32(ext[0] / np.float32(nx)) if nx >= 1 else np.float32(0)
        self.DY = np.float32(ext[1] / np.float32(ny)) if ny >= 1 else np.float32(0)
        self.DZ = np.float32(ext[2] / np.float32(nz)) if nz >= 1 else np.float32(0)
        self._h = None            # sdfk_volume*
        self._has_colors = False
        self._host_values = values
        self._host_colors = colors
        self._host_newer = values is not None

    # IBoundedVolume (IBoundedVolume.cs:6-13; Voxels.cs:17-21)
    @property
    def Center(self): return (self.Min + self.Max) * np.float32(0.5)
    @property
    def Size(self): return self.Max - self.Min
    @property
    def Radius(self):
        s = self.Size
        return np.float32(np.sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) * np.float32(0.5))

    # -- device handle --------------------------------------------------------
    def _ensure_device(self, with_colors):
        N.init()
        if self._h is not None and with_colors and not self._has_colors:
            self._free()
        if self._h is None:
            h = C.c_void_p()
            N.check(N.lib().sdfk_volume_create(self.NX, self.NY, self.NZ, N.f3(self.Min), N.f3(self.Max),
                                               1 if with_colors else 0, C.byref(h)))
            self._h, self._has_colors = h, bool(with_colors)
        return self._h

    def _sync_to_device(self):
        """Host arrays the caller may have edited win (Voxels.Values is a public array)."""
        if self._host_values is None:
            if self._h is None:  # never sampled: the reference's arrays are zero-filled
                self._host_values = np.zeros((self.NX, self.NY, self.NZ), dtype=np.float32)
            else:
                return self._h
        has_c = self._host_colors is not None and bool(np.any(self._host_colors))
        h = self._ensure_device(has_c or self._has_colors)
        cp = None
        if self._has_colors:
            if self._host_colors is None:
                self._host_colors = np.zeros((self.NX, self.NY, self.NZ, 3), dtype=np.float32)
            cp = self._host_colors.ctypes.data
        N.check(N.lib().sdfk_volume_upload(h, self._host_values.ctypes.data, cp))
        return h

    def _free(self):
        if self._h is not None and N._lib is not None and N._inited_device is not None:
            N._lib.sdfk_volume_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    # -- Values / Colors / indexers (Voxels.cs:8-9,42-65) -----------------------
    def _download(self):
        if self._host_values is None:
            self._host_values = np.zeros((self.NX, self.NY, self.NZ), dtype=np.float32)
            self._host_colors = np.zeros((self.NX, self.NY, self.NZ, 3), dtype=np.float32)
            if self._h is not None:
                N.check(N.lib().sdfk_volume_download(self._h, self._host_values.ctypes.data,
                                                     self._host_colors.ctypes.data))
        elif self._host_colors is None:
            self._host_colors = np.zeros((self.NX, self.NY, self.NZ, 3), dtype=np.float32)

    @property
    def Values(self):
        self._download()
        return self._host_values

    @property
    def Colors(self):
        self._download()
        return self._host_colors

    def _index_of(self, p):
        p = _v3(p)
        return (int((p[0] - self.Min[0]) / self.DX), int((p[1] - self.Min[1]) / self.DY),
                int((p[2] - self.Min[2]) / self.DZ))

    def __getitem__(self, key):
        if len(key) != 3 or not all(isinstance(k, (int, np.integer)) for k in key):
            key = self._index_of(key[0] if len(key) == 1 else key)
        return self.Values[key]

    def __setitem__(self, key, value):
        if len(key) != 3 or not all(isinstance(k, (int, np.integer)) for k in key):
            key = self._index_of(key[0] if len(key) == 1 else key)
        self.Values[key] = value

    # -- sampling ---------------------------------------------------------------
    def _sample(self, sdf, clip=False):
        if not isinstance(sdf, Sdf):
            raise TypeError("only SDFs built from Sdfs/SdfFuncs/SdfExprs can be lowered to the GPU; "
                            "an opaque delegate has no GPU form and this library has no CPU path")
        prog = sdf.program()
        h = self._ensure_device(sdf.writes_color)
        N.check(N.lib().sdfk_sample(prog, h, 1 if clip else 0))
        self._host_values = self._host_colors = None  # device copy is now the truth

    def _sample_instance(self, sdf, batchSize=DefaultBatchSize, maxDegreeOfParallelism=-1):
        """Voxels.SampleSdf(Sdf, batchSize, maxDegreeOfParallelism) (Voxels.cs:72-125).
        batchSize / maxDegreeOfParallelism are accepted and ignored on the GPU."""
        self._sample(sdf, clip=False)

    @staticmethod
    def _sample_static(sdf, min, max, nx, ny, nz, batchSize=DefaultBatchSize, maxDegreeOfParallelism=-1):
        """static Voxels.SampleSdf(Sdf, min, max, nx, ny, nz, ...) (Voxels.cs:169-174)."""
        if isinstance(sdf, SdfFunc):  # the Func<Vector3,Vector4> overload (Voxels.cs:176-189)
            sdf = sdf.ToSdf()
        v = Voxels(min, max, nx, ny, nz)
        v._sample(sdf, clip=False)
        return v

    # C# overloads one name for the instance and the static form
    SampleSdf = _InstanceOrStatic(_sample_instance, _sample_static)

    def ClipToBounds(self):
        """Voxels.ClipToBounds (Voxels.cs:133-167)."""
        h = self._sync_to_device()
        N.check(N.lib().sdfk_volume_clip_to_bounds(h))
        self._host_values = self._host_colors = None

    def ToMesh(self, isoValue=0.0, step=1, progress=None):
        """Voxels.ToMesh (Voxels.cs:67-70)."""
        return MarchingCubes.CreateMesh(self, isoValue, step, progress)


# ---------------------------------------------------------------------------
# MarchingCubes / Mesh
# ---------------------------------------------------------------------------
class MarchingCubes:
    @staticmethod
    def CreateMesh(volume, isoValue=0.0, step=1, progress=None):
        """MarchingCubes.CreateMesh(Voxels, isoValue, step, progress) (MarchingCubes.cs:39-92)."""
        h = volume._sync_to_device()
        m = C.c_void_p()
        N.check(N.lib().sdfk_march(h, C.c_float(isoValue), int(step), C.byref(m)))
        _report_progress(progress, volume.NZ, int(step))
        return Mesh._from_handle(m)


class MeshArrayPool:
    """Exact-length arrays of meshes that have been handed back (Mesh.Recycle), keyed by (dtype, shape).

    The reference's Mesh owns four managed arrays of exactly Vertices.Length / Triangles.Length elements (Mesh.cs:10-13), so
    the hand-off of a device mesh is four NEW arrays -- and on a growing managed heap a new array is memory nobody has touched:
    26 MB of page faults, 1.0 of the 1.6 ms one call takes at 512^3, against 0.75 ms into arrays whose pages are resident.  A
    host that meshes the same grid shape again and again (an editor, an animation) gets the same counts again and again:
    this pool hands the arrays of a recycled mesh to the next mesh of the same size.  A miss falls back to `alloc` (default: the
    library's pinned arena; the C# shim: GC.AllocateUninitializedArray, shim/SdfKit.Hip/Voxels.Hip.cs `MeshArrayPool`).
    Like System.Buffers.ArrayPool: whoever recycles a mesh must not touch its arrays afterwards."""

    def __init__(self, alloc=None, per_key=4, max_bytes=1 << 30):
        self.alloc = alloc or N.pinned_empty
        self.per_key, self.max_bytes = per_key, max_bytes
        self._free = {}          # (dtype str, shape) -> [arrays]
        self._order = []         # keys, least recently returned first
        self.bytes = 0
        self.hits = self.misses = 0

    def rent(self, shape, dtype):
        key = (np.dtype(dtype).str, tuple(shape))
        lst = self._free.get(key)
        if lst:
            a = lst.pop()
            self.bytes -= a.nbytes
            self.hits += 1
            return a
        self.misses += 1
        return self.alloc(shape, dtype)

    def give_back(self, a):
        if a is None or a.nbytes == 0:
            return
        key = (a.dtype.str, tuple(a.shape))
        lst = self._free.setdefault(key, [])
        if len(lst) >= self.per_key:
            return               # (dropped: the allocator behind it takes it back)
        lst.append(a)
        self.bytes += a.nbytes
        if key in self._order:
            self._order.remove(key)
        self._order.append(key)
        while self.bytes > self.max_bytes and self._order:      # the size class nobody has returned to for longest goes first
            old = self._order.pop(0)
            for b in self._free.pop(old, []):
                self.bytes -= b.nbytes

    def clear(self):
        self._free.clear()
        self._order.clear()
        self.bytes = 0


class Mesh:
    """Mesh.cs:8-64: Vertices/Colors/Normals [n,3] float32, Triangles int32[]."""

    Pool = None   # the default MeshArrayPool of _from_handle (created on first use)

    def __init__(self, vertices, colors, normals, triangles, mn=None, mx=None):
        self.Vertices, self.Colors, self.Normals, self.Triangles = vertices, colors, normals, triangles
        self.Min = np.zeros(3, np.float32) if mn is None else mn
        self.Max = np.zeros(3, np.float32) if mx is None else mx
        self.ActiveCells = 0
        self.ImpossibleCase13Cells = 0
        self._pool = None

    @staticmethod
    def _from_handle(h, pool=None):
        L = N.lib()
        nv, ni = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(h, C.byref(nv), C.byref(ni)))
        # The four exact-length arrays come from a pool of arrays that earlier meshes of the same size handed back
        # (Mesh.Recycle): resident pages, no first-touch faults.  A miss allocates -- by default in the library's pinned host
        # arena, where the copy below is a plain DMA transfer.
        if pool is None:
            if Mesh.Pool is None:
                Mesh.Pool = MeshArrayPool()
            pool = Mesh.Pool
        v = pool.rent((nv.value, 3), np.float32)
        c = pool.rent((nv.value, 3), np.float32)
        n = pool.rent((nv.value, 3), np.float32)
        t = pool.rent((ni.value,), np.int32)
        N.check(L.sdfk_mesh_copy(h, v.ctypes.data, c.ctypes.data, n.ctypes.data, t.ctypes.data))
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        N.check(L.sdfk_mesh_bounds(h, mn, mx))
        na, n13 = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_stats(h, C.byref(na), C.byref(n13)))
        L.sdfk_mesh_free(h)
        m = Mesh(v, c, n, t, np.array(mn[:], np.float32), np.array(mx[:], np.float32))
        m.ActiveCells, m.ImpossibleCase13Cells = na.value, n13.value
        m._pool = pool
        return m

    def Recycle(self):
        """Hands the four arrays back to the pool they came from (the next mesh of the same size gets them) and empties this
        mesh.  Not in the reference (a managed Mesh is simply collected); the shim offers the same opt-in
        (`Mesh.Recycle()`, shim/SdfKit.Hip/Voxels.Hip.cs).  The arrays must not be used afterwards."""
        pool = self._pool
        if pool is not None:
            for a in (self.Vertices, self.Colors, self.Normals, self.Triangles):
                pool.give_back(a)
        self.Vertices = self.Colors = self.Normals = np.zeros((0, 3), np.float32)
        self.Triangles = np.zeros((0,), np.int32)
        self._pool = None

    def Transform(self, transform):
        """Mesh.Transform(Matrix4x4) (Mesh.cs:47-64) on the host arrays: Vector3.Transform for the positions,
        Vector3.TransformNormal with Transpose(Invert(transform without its translation row)) + Vector3.Normalize for the
        normals, then Measure.  float32 throughout, products summed left to right (System.Numerics' row-vector form).
        (The device-resident form is sdfk_mesh_transform, for hosts that keep the handle.)"""
        from .raymarch import Matrix4x4
        f32 = np.float32
        M = np.array(transform, f32).reshape(4, 4)
        nm = M.copy()
        nm[3] = [0, 0, 0, 1]
        _, inv = Matrix4x4.Invert(nm)
        NT = inv.T.copy()
        V, Q = self.Vertices, self.Normals
        x, y, z = V[:, 0].copy(), V[:, 1].copy(), V[:, 2].copy()
        a, b, c = Q[:, 0].copy(), Q[:, 1].copy(), Q[:, 2].copy()
        with np.errstate(all="ignore"):
            for j in range(3):
                V[:, j] = ((x * M[0, j] + y * M[1, j]) + z * M[2, j]) + M[3, j]
            t = [(a * NT[0, j] + b * NT[1, j]) + c * NT[2, j] for j in range(3)]
            ln = np.sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2])
            for j in range(3):
                Q[:, j] = t[j] / ln
        if len(V):     # Measure (Mesh.cs:30-45): Vector3.Min / Max componentwise, in order
            self.Min, self.Max = V.min(axis=0), V.max(axis=0)

    @staticmethod
    def normal_matrix(transform):
        """The matrix Mesh.Transform applies to normals (Mesh.cs:49-55), for sdfk_mesh_transform."""
        from .raymarch import Matrix4x4
        nm = np.array(transform, np.float32).reshape(4, 4).copy()
        nm[3] = [0, 0, 0, 1]
        return Matrix4x4.Invert(nm)[1].T.copy()

    @property
    def Center(self): return (self.Min + self.Max) * np.float32(0.5)
    @property
    def Size(self): return self.Max - self.Min
    @property
    def Radius(self):
        s = self.Size
        return np.float32(np.sqrt((s[0] * s[0] + s[1] * s[1]) + s[2] * s[2]) * np.float32(0.5))

    def WriteObj(self, path_or_file):
        """Mesh.WriteObj (Mesh.cs:66-97): `v`, then `vn`, then `f a//a b//b c//c`, 1-based."""
        own = isinstance(path_or_file, str)
        w = open(path_or_file, "w") if own else path_or_file
        try:
            for v in self.Vertices:
                w.write("v %s %s %s\n" % tuple(_fmt_single(x) for x in v))
            for v in self.Normals:
                w.write("vn %s %s %s\n" % tuple(_fmt_single(x) for x in v))
            t = self.Triangles
            for i in range(0, len(t), 3):
                a, b, c = int(t[i]) + 1, int(t[i + 1]) + 1, int(t[i + 2]) + 1
                w.write(f"f {a}//{a} {b}//{b} {c}//{c}\n")
        finally:
            if own:
                w.close()


def _fmt_single(x):
    """Invariant-culture System.Single.ToString() of .NET Core 3.0+ (the runtime the reference's global.json pins): the
    shortest round-trip digits through format 'G'; scientific (d.dddE+XX) when the decimal-point position (decimal exponent
    + 1) exceeds max(number of digits, 7) or is below -3 -- Number.Formatting.cs: nMaxDigits = Math.Max(number.DigitsCount,
    SinglePrecision), FormatGeneral: digPos > nMaxDigits || digPos < -3.  12345678f -> "12345678", 1e7f -> "1E+07"."""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "Infinity" if x > 0 else "-Infinity"
    if x == 0:
        return "-0" if np.signbit(x) else "0"
    s = np.format_float_scientific(x, unique=True, trim="-")  # d.ddde±XX
    mant, exp = s.split("e")
    e = int(exp)
    neg = mant.startswith("-")
    digits = mant.lstrip("-").replace(".", "")
    if -5 < e < max(len(digits), 7):
        if e >= 0:
            ip, fp = digits[:e + 1].ljust(e + 1, "0"), digits[e + 1:]
        else:
            ip, fp = "0", "0" * (-e - 1) + digits
        out = ip + ("." + fp if fp else "")
    else:
        out = digits[0] + ("." + digits[1:] if len(digits) > 1 else "") + "E" + ("+" if e >= 0 else "-") + "%02d" % abs(e)
    return ("-" if neg else "") + out
