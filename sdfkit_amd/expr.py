"""Symbolic float32 values that record an SDF program (include/sdfkit_hip.h `sdfk_op`).

This plays the role LINQ expression trees play in the reference: `SdfExpr =
Expression<SdfFunc>` (GlobalUsings.cs:16) keeps the scene inspectable so it can be
compiled (SdfExpr.cs:225-273).  Here Python operator overloading records every float32
operation, in the reference's evaluation order, as one SSA instruction; the native
library JIT-compiles the list for the GPU.  Nothing is evaluated on the host.

Arithmetic conventions restated from the .NET BCL (System.Numerics, not in the
reference tree): Vector3.Length = sqrt((x*x + y*y) + z*z); Vector3.Dot likewise;
Vector3.Min/Max = compare-select; MathF.Max/Min and Math.Max/Min = IEEE-754:2019
maximum/minimum; Vector3 / float = component-wise division; float * Vector3 =
component-wise multiply.
"""
import numpy as np

# opcodes (enum sdfk_opcode)
OP_CONST, OP_X, OP_Y, OP_Z = 0, 1, 2, 3
OP_ADD, OP_SUB, OP_MUL, OP_DIV = 4, 5, 6, 7
OP_NEG, OP_ABS, OP_SQRT, OP_FLOOR = 8, 9, 10, 11
OP_MIN_SEL, OP_MAX_SEL, OP_MIN_IEEE, OP_MAX_IEEE, OP_SEL_LT = 12, 13, 14, 15, 16


class Builder:
    """Accumulates the flat SSA instruction list."""

    def __init__(self):
        self.ops = []           # (opcode, a, b, c, d, imm)
        self._inputs = {}

    def emit(self, opcode, a=-1, b=-1, c=-1, d=-1, imm=0.0):
        self.ops.append((opcode, a, b, c, d, float(imm)))
        return Val(self, len(self.ops) - 1)

    def const(self, x):
        # One CONST per mention, never merged by VALUE: the constants of a program are kernel arguments and the compiled
        # kernels are shared by every program of the same structure (sdfk_program_create) -- a structure must not change
        # because two of a scene's parameters happen to be equal in one frame of an animation.
        return self.emit(OP_CONST, imm=float(np.float32(x)))

    def input(self, axis):
        if axis not in self._inputs:
            self._inputs[axis] = self.emit((OP_X, OP_Y, OP_Z)[axis])
        return self._inputs[axis]

    def lift(self, x):
        if isinstance(x, Val):
            if x.b is not self:
                raise ValueError("value belongs to another program")
            return x
        return self.const(x)


class Val:
    """One float32 SSA value."""
    __slots__ = ("b", "id")

    def __init__(self, b, id_):
        self.b, self.id = b, id_

    def _bin(self, op, other, swap=False):
        o = self.b.lift(other)
        l, r = (o, self) if swap else (self, o)
        return self.b.emit(op, l.id, r.id)

    def __add__(self, o): return self._bin(OP_ADD, o)
    def __radd__(self, o): return self._bin(OP_ADD, o, True)
    def __sub__(self, o): return self._bin(OP_SUB, o)
    def __rsub__(self, o): return self._bin(OP_SUB, o, True)
    def __mul__(self, o): return self._bin(OP_MUL, o)
    def __rmul__(self, o): return self._bin(OP_MUL, o, True)
    def __truediv__(self, o): return self._bin(OP_DIV, o)
    def __rtruediv__(self, o): return self._bin(OP_DIV, o, True)
    def __neg__(self): return self.b.emit(OP_NEG, self.id)
    def __abs__(self): return self.b.emit(OP_ABS, self.id)


def _builder_of(*xs):
    for x in xs:
        if isinstance(x, Val):
            return x.b
        if isinstance(x, Vec3):
            return x.x.b
    raise ValueError("no symbolic operand")


class MathF:
    """System.MathF members the SDF catalogue uses."""

    @staticmethod
    def Sqrt(a): return a.b.emit(OP_SQRT, a.id)
    @staticmethod
    def Abs(a): return a.b.emit(OP_ABS, a.id)
    @staticmethod
    def Floor(a): return a.b.emit(OP_FLOOR, a.id)
    @staticmethod
    def Max(a, b_):
        b = _builder_of(a, b_)
        return b.emit(OP_MAX_IEEE, b.lift(a).id, b.lift(b_).id)
    @staticmethod
    def Min(a, b_):
        b = _builder_of(a, b_)
        return b.emit(OP_MIN_IEEE, b.lift(a).id, b.lift(b_).id)


def select_lt(a, b_, c, d):
    """(a < b) ? c : d"""
    b = _builder_of(a, b_, c, d)
    return b.emit(OP_SEL_LT, b.lift(a).id, b.lift(b_).id, b.lift(c).id, b.lift(d).id)


class Vec3:
    """System.Numerics.Vector3 over symbolic components."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = x, y, z

    X = property(lambda s: s.x)
    Y = property(lambda s: s.y)
    Z = property(lambda s: s.z)

    @staticmethod
    def of(b, v):
        if isinstance(v, Vec3):
            return v
        if np.isscalar(v):
            c = b.const(v)
            return Vec3(c, c, c)
        return Vec3(b.lift(v[0]), b.lift(v[1]), b.lift(v[2]))

    def _map2(self, o, f):
        b = _builder_of(self)
        if isinstance(o, (Val, int, float, np.floating)):
            s = b.lift(o)
            return Vec3(f(self.x, s), f(self.y, s), f(self.z, s))
        o = Vec3.of(b, o)
        return Vec3(f(self.x, o.x), f(self.y, o.y), f(self.z, o.z))

    def __add__(self, o): return self._map2(o, lambda a, c: a + c)
    def __sub__(self, o): return self._map2(o, lambda a, c: a - c)
    def __mul__(self, o): return self._map2(o, lambda a, c: a * c)
    def __truediv__(self, o): return self._map2(o, lambda a, c: a / c)
    def __rmul__(self, o): return self._map2(o, lambda a, c: c * a)
    def __rsub__(self, o): return Vec3.of(_builder_of(self), o) - self
    def __radd__(self, o): return Vec3.of(_builder_of(self), o) + self
    def __neg__(self): return Vec3(-self.x, -self.y, -self.z)

    def Length(self):
        return MathF.Sqrt((self.x * self.x + self.y * self.y) + self.z * self.z)

    @staticmethod
    def Dot(a, c):
        c = Vec3.of(_builder_of(a), c)
        return (a.x * c.x + a.y * c.y) + a.z * c.z

    @staticmethod
    def Abs(a): return Vec3(abs(a.x), abs(a.y), abs(a.z))

    @staticmethod
    def Max(a, c):
        b = _builder_of(a)
        c = Vec3.of(b, c)
        return Vec3(*(b.emit(OP_MAX_SEL, p.id, q.id) for p, q in ((a.x, c.x), (a.y, c.y), (a.z, c.z))))

    @staticmethod
    def Min(a, c):
        b = _builder_of(a)
        c = Vec3.of(b, c)
        return Vec3(*(b.emit(OP_MIN_SEL, p.id, q.id) for p, q in ((a.x, c.x), (a.y, c.y), (a.z, c.z))))


class Vec4:
    """SdfOutput = Vector4(colour.XYZ, distance W) (GlobalUsings.cs:10)."""
    __slots__ = ("x", "y", "z", "w")

    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = x, y, z, w

    W = property(lambda s: s.w)

    @staticmethod
    def of(color, w):
        b = _builder_of(w)
        c = Vec3.of(b, color)
        return Vec4(c.x, c.y, c.z, w)


def Mod(a, b_):
    """VectorOps.Mod: a - b*floor(a/b) (VectorData.cs:697-698)."""
    b = _builder_of(a, b_)
    a, b_ = b.lift(a), b.lift(b_)
    return a - b_ * MathF.Floor(a / b_)


def VMax(v):
    """VectorOps.VMax: Math.Max(Math.Max(x, y), z) (VectorData.cs:860-861)."""
    return MathF.Max(MathF.Max(v.x, v.y), v.z)


def trace(fn, writes_color=True):
    """Run the per-point function `fn(Vec3) -> Vec4` symbolically.
    Returns (ops, out_rgbw) ready for sdfk_program_create."""
    b = Builder()
    p = Vec3(b.input(0), b.input(1), b.input(2))
    out = fn(p)
    if not isinstance(out, Vec4):
        raise TypeError("an SDF must return Vec4(colour, distance)")
    w = b.lift(out.w)
    if writes_color:
        rgb = [b.lift(out.x).id, b.lift(out.y).id, b.lift(out.z).id]
    else:
        rgb = [-1, -1, -1]
    return b.ops, rgb + [w.id]
