"""The op lists the C# shim's lowering (shim/SdfKit.Hip/Lowering.cs, GpuProgram.cs) emits, replayed through the
C ABI.  .NET is not available here, so the visitor cannot run; what CAN be pinned is its output for known trees:
`Emitter` below is GpuProgram's builder (ops 0-2 = X, Y, Z; constants pooled by bit pattern at first use; no CSE)
and each scene function performs, call for call, what Lowering.Visitor does on the expression tree the reference
builds (SdfExpr.cs) -- operands left to right, then the operation.

CPU: the lists compile (sdfk_program_check) and evaluate bit-identically to the Python mirror's own programs under
the numpy IR interpreter.  GPU: they sample and mesh bit-identically to the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import ir_interp
from oracle import oracle as O
from sdfkit_amd import _native as N
from sdfkit_amd import expr as E
from tests import scenes as S

CONST, X, Y, Z, ADD, SUB, MUL, DIV, NEG, ABS, SQRT, FLOOR, MIN_SEL, MAX_SEL, MIN_IEEE, MAX_IEEE, SEL_LT = range(17)


class Emitter:
    """GpuProgram (shim/SdfKit.Hip/GpuProgram.cs)"""

    def __init__(self):
        self.ops, self.consts = [], {}
        self.X, self.Y, self.Z = self.emit(X), self.emit(Y), self.emit(Z)

    def emit(self, op, a=-1, b=-1, c=-1, d=-1, imm=0.0):
        self.ops.append((op, a, b, c, d, float(imm)))
        return len(self.ops) - 1

    def const(self, x):   # GpuProgram.Const: one CONST per mention (constants are kernel arguments: never pooled by value)
        return self.emit(CONST, imm=float(np.float32(x)))

    # Lowering.Length / Visitor.Call("Mod")
    def length(self, x, y, z):
        return self.emit(SQRT, self.emit(ADD, self.emit(ADD, self.emit(MUL, x, x), self.emit(MUL, y, y)), self.emit(MUL, z, z)))

    def mod(self, x, y):
        return self.emit(SUB, x, self.emit(MUL, y, self.emit(FLOOR, self.emit(DIV, x, y))))


def _repeat_coord(g, c, size):
    # Subtract(Call Mod(Add(p.C, Multiply(size, 0.5f)), size), Multiply(size, 0.5f))      SdfExpr.cs:151,159-160
    a = g.emit(ADD, c, g.emit(MUL, g.const(size), g.const(0.5)))
    m = g.mod(a, g.const(size))
    return g.emit(SUB, m, g.emit(MUL, g.const(size), g.const(0.5)))


def _repeat_index(g, c, size):
    # Call MathF.Floor(Divide(Add(p.C, Multiply(size, 0.5f)), size))                      SdfExpr.cs:172-173
    return g.emit(FLOOR, g.emit(DIV, g.emit(ADD, c, g.emit(MUL, g.const(size), g.const(0.5))), g.const(size)))


def readme_scene():
    """SdfExprs.Sphere(0.5f).RepeatXY(1.125f, 1.125f, (i,p,d) => 0.9f*Vector3.One - Vector3.Abs(i)/6f)   README.md:24-30
    = ModifyInputAndOutput block (SdfExpr.cs:113-141): i = modInput(p); mp = i.Position; d = sdf(mp);
    mo = modOutput(i.Index, mp, d); new Vector4(mo, d.W)."""
    g = Emitter()
    pos = (_repeat_coord(g, g.X, 1.125), _repeat_coord(g, g.Y, 1.125), g.Z)       # MemberInit: Position first ...
    idx = (_repeat_index(g, g.X, 1.125), _repeat_index(g, g.Y, 1.125), g.const(0.0))   # ... then Index
    one = g.const(1.0)                                   # new Vector4(1, 1, 1, p.Length() - r)   SdfExpr.cs:50-51
    w = g.emit(SUB, g.length(*pos), g.const(0.5))
    k = g.const(0.9)                                     # Multiply(0.9f, Vector3.One): left operand first
    one3 = (g.const(1.0),) * 3
    lhs = [g.emit(MUL, k, o) for o in one3]
    ab = [g.emit(ABS, i) for i in idx]                   # Vector3.Abs(i)
    six = g.const(6.0)
    rhs = [g.emit(DIV, a, six) for a in ab]              # Vector3 / float: scalar broadcast
    mo = [g.emit(SUB, l, r) for l, r in zip(lhs, rhs)]
    assert one != one3[0]            # (one CONST per mention: GpuProgram.Const does not pool by value)
    return g.ops, mo + [w], True


def union_scene():
    """SdfExprs.Union(SdfExprs.Sphere(0.6f).ModifyInput(p => p - c1), SdfExprs.Box(0.5f).ModifyInput(p => p - c2)):
    Block { da = a(p); db = b(p); (da.W < db.W) ? da : db }   SdfExpr.cs:53-68."""
    g = Emitter()

    def translate(c):            # op_Subtraction(p, constant Vector3): component-wise
        cs = [g.const(v) for v in c]
        return [g.emit(SUB, p, q) for p, q in zip((g.X, g.Y, g.Z), cs)]

    p1 = translate((-1.0, 0.25, 0.5))
    one = g.const(1.0)
    da = [one, one, one, g.emit(SUB, g.length(*p1), g.const(0.6))]
    p2 = translate((1.0, -0.25, 0.0))
    # Box (SdfExpr.cs:18-24): new Vector4(Vector3.One, Max(Abs(p) - bounds, Zero).Length() + VMax(Min(Abs(p) - bounds, Zero)))
    one_b = [g.const(1.0)] * 3
    b = [g.const(0.5)] * 3

    def wd():
        ab = [g.emit(ABS, q) for q in p2]
        return [g.emit(SUB, a, c) for a, c in zip(ab, b)]

    w1 = wd()
    zero = [g.const(0.0)] * 3
    hi = [g.emit(MAX_SEL, a, z) for a, z in zip(w1, zero)]
    ln = g.length(*hi)
    w2 = wd()                    # the expression-tree Box evaluates Abs(p) - bounds twice (no local variable)
    lo = [g.emit(MIN_SEL, a, z) for a, z in zip(w2, zero)]
    vmax = g.emit(MAX_IEEE, g.emit(MAX_IEEE, lo[0], lo[1]), lo[2])
    db = one_b + [g.emit(ADD, ln, vmax)]
    out = [g.emit(SEL_LT, da[3], db[3], t, f) for t, f in zip(da, db)]
    return g.ops, out, True


def sphere_tag():
    """GpuSdf.Sphere(1) -- the [GpuProgram] tag of Sdfs.Sphere (Sdf.cs:202-215): W only."""
    g = Emitter()
    w = g.emit(SUB, g.length(g.X, g.Y, g.Z), g.const(1.0))
    return g.ops, [-1, -1, -1, w], False


def _c_ops(ops, out):
    arr = (N.Op * len(ops))()
    for i, (op, a, b, c, d, imm) in enumerate(ops):
        arr[i].opcode, arr[i].a, arr[i].b, arr[i].c, arr[i].d, arr[i].imm = op, a, b, c, d, imm
    return arr, len(ops), (C.c_int32 * 4)(*out)


def _union_mirror():
    from sdfkit_amd import SdfExprs, Vec3
    a = SdfExprs.Sphere(0.6).ModifyInput(lambda p: p - Vec3.of(p.x.b, (-1.0, 0.25, 0.5)))
    b = SdfExprs.Box(0.5).ModifyInput(lambda p: p - Vec3.of(p.x.b, (1.0, -0.25, 0.0)))
    return SdfExprs.Union(a, b).ToSdf()


def _union_oracle():
    s = O.Scene()
    a = s.f_translate(s.f_sphere(0.6), -1.0, 0.25, 0.5)
    b = s.f_translate(s.f_box(0.5), 1.0, -0.25, 0.0)
    s.f_union(a, b)
    return s


CASES = {
    "readme": (readme_scene, lambda: S.readme_repeat_xy()[1], lambda: S.readme_repeat_xy()[0]),
    "union": (union_scene, _union_mirror, _union_oracle),
    "sphere_tag": (sphere_tag, lambda: S.sphere_w(1.0)[1], lambda: S.sphere_w(1.0)[0]),
}


def test_readme_list_is_the_documented_one():
    ops, out, _ = readme_scene()
    assert len(ops) == 68 and out == [65, 66, 67, 52]      # (52 ops while GpuProgram.Const pooled constants by value: rounds 2-3)
    assert [o[0] for o in ops[:16]] == [X, Y, Z, CONST, CONST, MUL, ADD, CONST, DIV, FLOOR, MUL, SUB, CONST, CONST, MUL, SUB]
    assert ops[3][5] == 1.125 and ops[4][5] == 0.5 and ops[7][5] == 1.125 and ops[15][1:3] == (11, 14)


@pytest.mark.parametrize("name", sorted(CASES))
def test_shim_lists_compile_and_match_the_mirror(name):
    build, mirror, _ = CASES[name]
    ops, out, wc = build()
    arr, n, o = _c_ops(ops, out)
    N.check(N.lib().sdfk_program_check(arr, n, o, int(wc)))          # hiprtc for gfx950, no device needed
    # same function as the Python mirror's program, bit for bit, on awkward points
    m_arr, m_n, m_out = mirror().ir()
    m_ops = [(m_arr[i].opcode, m_arr[i].a, m_arr[i].b, m_arr[i].c, m_arr[i].d, m_arr[i].imm) for i in range(m_n)]
    rng = np.random.default_rng(5)
    pts = (rng.standard_normal((4096, 3)) * 2.5).astype(np.float32)
    pts[:64] = np.float32(0.5625) * rng.integers(-6, 7, (64, 3)).astype(np.float32)   # on the repeat seams
    mine = ir_interp.run(ops, out, pts)
    theirs = ir_interp.run(m_ops, list(m_out), pts)
    for k in range(4):
        if out[k] >= 0:
            assert np.array_equal(mine[k].view(np.uint32), theirs[k].view(np.uint32)), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_shim_lists_through_the_abi(gpu, name):
    """sdfk_program_create on the shim's list -> sdfk_sample_march -> the oracle's mesh, array by array."""
    from sdfkit_amd import Mesh
    from tests.test_gpu_parity import assert_mesh_equal
    build, _, oracle_scene = CASES[name]
    ops, out, wc = build()
    arr, n, o = _c_ops(ops, out)
    L = N.lib()
    prog = C.c_void_p()
    N.check(L.sdfk_program_create(arr, n, o, int(wc), C.byref(prog)))
    mn, mx, dims = [-2.8125] * 3, [2.8125] * 3, (64, 56, 48)
    try:
        h = C.c_void_p()
        N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), *dims, 1, C.c_float(0.0), 1, C.byref(h)))
        m = Mesh._from_handle(h)
    finally:
        L.sdfk_program_destroy(prog)
    ov, oc = O.sample(oracle_scene(), mn, mx, *dims)
    O.clip_to_bounds(ov, mn, mx)
    assert_mesh_equal(m, O.march(ov, oc, mn, mx))
    assert len(m.Vertices) > 500
