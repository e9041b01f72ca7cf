"""Host-side mirror of SdfKit's RayMarcher (RayMarcher.cs) and of the image types it returns,
over the C ABI entry sdfk_raymarch.

The camera set-up of RayMarcher.GetCameraRays (RayMarcher.cs:97-112) is host work in the
reference too (System.Numerics Matrix4x4): the C# shim computes it with the BCL; here the
same float32 arithmetic is restated with numpy scalars (`Matrix4x4.CreateLookAt`,
`CreatePerspectiveFieldOfView`, `operator *`, `Invert` -- software cofactor path).  Per-pixel
work runs in the JIT kernel `sdfk_raymarch` (csrc/sample_codegen.h).
"""
import ctypes as C
import ctypes.util

import numpy as np

from . import _native as N

f32 = np.float32
_libm = None


def _tanf(x):
    """MathF.Tan: the C runtime's tanf (numpy's own float32 tan may round differently)."""
    global _libm
    if _libm is None:
        _libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm.tanf.restype = C.c_float
        _libm.tanf.argtypes = [C.c_float]
    return f32(_libm.tanf(C.c_float(float(x))))


def _v3(v):
    return [f32(v[0]), f32(v[1]), f32(v[2])]


def _length(v):
    return f32(np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))


def _normalize(v):          # Vector3.Normalize = v / Length(v)
    l = _length(v)
    with np.errstate(all="ignore"):
        return [v[0] / l, v[1] / l, v[2] / l]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


class Matrix4x4:
    """The handful of System.Numerics.Matrix4x4 members the RayMarcher uses; row-major 4x4 float32."""

    @staticmethod
    def CreateLookAt(cameraPosition, cameraTarget, cameraUpVector):
        pos, tgt, up = _v3(cameraPosition), _v3(cameraTarget), _v3(cameraUpVector)
        z = _normalize([pos[0] - tgt[0], pos[1] - tgt[1], pos[2] - tgt[2]])
        x = _normalize(_cross(up, z))
        y = _cross(z, x)
        m = np.zeros((4, 4), f32)
        m[0, :3] = [x[0], y[0], z[0]]
        m[1, :3] = [x[1], y[1], z[1]]
        m[2, :3] = [x[2], y[2], z[2]]
        m[3] = [-_dot(x, pos), -_dot(y, pos), -_dot(z, pos), f32(1)]
        return m

    @staticmethod
    def CreatePerspectiveFieldOfView(fieldOfView, aspectRatio, nearPlaneDistance, farPlaneDistance):
        fov, aspect, near, far = f32(fieldOfView), f32(aspectRatio), f32(nearPlaneDistance), f32(farPlaneDistance)
        yscale = f32(1) / _tanf(fov * f32(0.5))
        xscale = yscale / aspect
        nfr = f32(-1) if np.isposinf(far) else far / (near - far)
        m = np.zeros((4, 4), f32)
        m[0, 0], m[1, 1], m[2, 2], m[2, 3], m[3, 2] = xscale, yscale, nfr, f32(-1), near * nfr
        return m

    @staticmethod
    def Multiply(a, b):
        a, b = np.asarray(a, f32), np.asarray(b, f32)
        r = np.zeros((4, 4), f32)
        for i in range(4):
            for j in range(4):
                r[i, j] = ((a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]) + a[i, 3] * b[3, j]
        return r

    @staticmethod
    def Invert(M):
        (a, b, c, d), (e, f, g, h), (i, j, k, l), (m, n, o, p) = [[f32(x) for x in row] for row in np.asarray(M, f32)]
        kp_lo, jp_ln, jo_kn = k * p - l * o, j * p - l * n, j * o - k * n
        ip_lm, io_km, in_jm = i * p - l * m, i * o - k * m, i * n - j * m
        a11 = +(f * kp_lo - g * jp_ln + h * jo_kn)
        a12 = -(e * kp_lo - g * ip_lm + h * io_km)
        a13 = +(e * jp_ln - f * ip_lm + h * in_jm)
        a14 = -(e * jo_kn - f * io_km + g * in_jm)
        det = a * a11 + b * a12 + c * a13 + d * a14
        if abs(det) < f32(1.1920929e-07):
            return False, np.full((4, 4), np.nan, f32)
        inv = f32(1) / det
        R = np.zeros((4, 4), f32)
        R[0, 0], R[1, 0], R[2, 0], R[3, 0] = a11 * inv, a12 * inv, a13 * inv, a14 * inv
        R[0, 1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv
        R[1, 1] = +(a * kp_lo - c * ip_lm + d * io_km) * inv
        R[2, 1] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv
        R[3, 1] = +(a * jo_kn - b * io_km + c * in_jm) * inv
        gp_ho, fp_hn, fo_gn = g * p - h * o, f * p - h * n, f * o - g * n
        ep_hm, eo_gm, en_fm = e * p - h * m, e * o - g * m, e * n - f * m
        R[0, 2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv
        R[1, 2] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv
        R[2, 2] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv
        R[3, 2] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv
        gl_hk, fl_hj, fk_gj = g * l - h * k, f * l - h * j, f * k - g * j
        el_hi, ek_gi, ej_fi = e * l - h * i, e * k - g * i, e * j - f * i
        R[0, 3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv
        R[1, 3] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv
        R[2, 3] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv
        R[3, 3] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv
        return True, R


class FloatData:
    """FloatData (VectorData.cs:137-280): Values is row-major [Height, Width]; indexer is [x, y]."""

    def __init__(self, values):
        self.Values = values
        self.Height, self.Width = values.shape

    def __getitem__(self, xy):
        x, y = xy
        return self.Values[y, x]

    def SaveDepthTga(self, path, near, far):
        """VectorData.cs:244-279: 8-bit greyscale, top-down, 255 at `near`, 0 at `far`."""
        near, far = f32(near), f32(far)
        v = self.Values
        with np.errstate(all="ignore"):
            g = (f32(255.0) * (far - v) / (far - near)).astype(np.uint8)
        g = np.where(v >= far, np.uint8(0), np.where(v <= near, np.uint8(255), g)).astype(np.uint8)
        with open(path, "wb") as f:
            f.write(_tga_header(3, self.Width, self.Height, 8))
            f.write(g.tobytes())


class Vec3Data:
    """Vec3Data (VectorData.cs:343-620): Values is row-major [Height, Width, 3]."""

    def __init__(self, values):
        self.Values = values
        self.Height, self.Width = values.shape[:2]

    def __getitem__(self, xy):
        x, y = xy
        return self.Values[y, x]

    def SaveTga(self, path):
        """VectorData.cs:570-619: 24-bit BGR, top-down, channel * 255 truncated and clamped."""
        with np.errstate(all="ignore"):
            v = self.Values[:, :, ::-1] * f32(255.0)
            b = np.where(v <= 0, 0, np.where(v >= 255, 255, np.nan_to_num(v, nan=0.0))).astype(np.uint8)
        with open(path, "wb") as f:
            f.write(_tga_header(2, self.Width, self.Height, 24))
            f.write(b.tobytes())


def _tga_header(image_type, width, height, bpp):
    import struct
    return struct.pack("<BBBHHBHHHHBB", 0, 0, image_type, 0, 0, 0, 0, 0, width, height, bpp, 0b00100000)


class RayMarcher:
    """RayMarcher (RayMarcher.cs:7-43): same constructor, properties and defaults."""
    DefaultNearPlaneDistance = 1.0
    DefaultFarPlaneDistance = 100.0
    DefaultVerticalFieldOfViewDegrees = 60.0
    DefaultDepthIterations = 40

    def __init__(self, width, height, sdf, batchSize=2048, maxDegreeOfParallelism=-1):
        self.width, self.height, self.sdf = int(width), int(height), sdf
        self.ViewTransform = Matrix4x4.CreateLookAt((0, 0, 5), (0, 0, 0), (0, 1, 0))   # RayMarcher.cs:22-23
        self.NearPlaneDistance = self.DefaultNearPlaneDistance
        self.FarPlaneDistance = self.DefaultFarPlaneDistance
        self.VerticalFieldOfViewDegrees = self.DefaultVerticalFieldOfViewDegrees
        self.DepthIterations = self.DefaultDepthIterations

    def camera(self):
        """Host part of GetCameraRays (RayMarcher.cs:97-112): (camera position, inverse view-projection)."""
        _, cam = Matrix4x4.Invert(self.ViewTransform)
        zero = f32(0)
        pos = [((zero * cam[0, q] + zero * cam[1, q]) + zero * cam[2, q]) + cam[3, q] for q in range(3)]
        proj = Matrix4x4.CreatePerspectiveFieldOfView(
            f32(self.VerticalFieldOfViewDegrees) * f32(np.pi) / f32(180.0),
            f32(self.width) / f32(self.height), self.NearPlaneDistance, self.FarPlaneDistance)
        _, vpi = Matrix4x4.Invert(Matrix4x4.Multiply(self.ViewTransform, proj))
        return np.array(pos, f32), vpi

    def _run(self, want_depth, want_rgb):
        from .api import Sdf
        if not isinstance(self.sdf, Sdf):
            raise TypeError("only SDFs built from Sdfs/SdfFuncs/SdfExprs can be lowered to the GPU")
        pos, vpi = self.camera()
        depth = np.empty((self.height, self.width), f32) if want_depth else None
        rgb = np.empty((self.height, self.width, 3), f32) if want_rgb else None
        N.check(N.lib().sdfk_raymarch(self.sdf.program(), self.width, self.height, N.f3(pos),
                                      (C.c_float * 16)(*[float(x) for x in vpi.ravel()]),
                                      C.c_float(self.NearPlaneDistance), C.c_float(self.FarPlaneDistance),
                                      int(self.DepthIterations),
                                      depth.ctypes.data if want_depth else None, rgb.ctypes.data if want_rgb else None))
        return depth, rgb

    def Render(self):
        """RayMarcher.Render (RayMarcher.cs:45-66): RGB image."""
        return Vec3Data(self._run(False, True)[1])

    def RenderDepth(self):
        """RayMarcher.RenderDepth (RayMarcher.cs:71-78): depth along every ray."""
        return FloatData(self._run(True, False)[0])


def to_image(sdf, width, height, *camera, verticalFieldOfViewDegrees=RayMarcher.DefaultVerticalFieldOfViewDegrees,
             nearPlaneDistance=RayMarcher.DefaultNearPlaneDistance, farPlaneDistance=RayMarcher.DefaultFarPlaneDistance,
             depthIterations=RayMarcher.DefaultDepthIterations, batchSize=2048, maxDegreeOfParallelism=-1):
    """SdfEx.ToImage (Sdf.cs:65-99): camera = (viewTransform,) or (position, target, up)."""
    view = camera[0] if len(camera) == 1 else Matrix4x4.CreateLookAt(*camera)
    rm = RayMarcher(width, height, sdf, batchSize, maxDegreeOfParallelism)
    rm.ViewTransform = np.asarray(view, f32)
    rm.VerticalFieldOfViewDegrees = verticalFieldOfViewDegrees
    rm.NearPlaneDistance, rm.FarPlaneDistance = nearPlaneDistance, farPlaneDistance
    rm.DepthIterations = depthIterations
    return rm.Render()
