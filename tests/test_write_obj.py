"""Mesh.WriteObj (Mesh.cs:66-97) and the invariant-culture Single.ToString() behind it, without a GPU.

Two writers exist -- sdfkit_amd.api.Mesh.WriteObj (Python mirror) and SdfKit::Mesh::WriteObj (include/SdfKit.hpp) -- and the
reference's tests never compare the text they produce.  What pins it here:
  * the formatting RULE of .NET Core 3.0+ (the runtime global.json pins; Number.Formatting.cs: shortest round-trip digits
    through format 'G', scientific iff the decimal-point position > max(digit count, 7) or < -3) on literals a .NET host
    prints: 12345678f -> "12345678", 16777216f -> "16777216", 1e7f -> "1E+07", 1e-5f -> "1E-05";
  * the two writers against EACH OTHER: every float of a sweep, and the OBJ text of the reference's Sphere5 mesh
    (MarchingCubesTests.cs:31-45, 54 vertices) line by line."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sdfkit_amd.api import Mesh, _fmt_single

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fmt") / "libformat_host.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "format_host.cpp"), "-o", so])
    L = C.CDLL(so)
    L.fmt_single.argtypes = [C.c_float, C.c_char_p, C.c_int]
    L.write_obj.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_char_p]
    return L


def _cpp(L, x):
    buf = C.create_string_buffer(64)
    assert L.fmt_single(C.c_float(float(x)), buf, 64) > 0
    return buf.value.decode()


# what `((float)x).ToString(CultureInfo.InvariantCulture)` prints on .NET Core 3.0+ (Number.Formatting.cs)
DOTNET = [
    (1.5, "1.5"), (-0.1, "-0.1"), (100.0, "100"), (0.33333334, "0.33333334"), (1234567.0, "1234567"),
    (12345678.0, "12345678"),            # 8 digits, decimal point at 8 = max(8, 7): fixed
    (16777216.0, "16777216"), (-12345678.0, "-12345678"),
    (1e7, "1E+07"), (1.5e7, "1.5E+07"),  # 1-2 digits, decimal point at 8 > 7
    (123456792.0, "1.2345679E+08"),      # 8 digits, decimal point at 9
    (105242136.0, "105242136"),          # 9 digits, decimal point at 9
    (1052421360.0, "1.0524214E+09"), (3.4028235e38, "3.4028235E+38"),
    (0.0001, "0.0001"), (0.00012345678, "0.00012345678"),   # decimal point at -3: fixed
    (1e-5, "1E-05"), (0.000012345678, "1.2345678E-05"), (1e-45, "1E-45"),
    (0.0, "0"), (-0.0, "-0"), (float("nan"), "NaN"), (float("inf"), "Infinity"), (float("-inf"), "-Infinity"),
]


@pytest.mark.parametrize("x,text", DOTNET)
def test_single_to_string_rule(host, x, text):
    assert _fmt_single(np.float32(x)) == text
    assert _cpp(host, np.float32(x)) == text


def test_python_and_cpp_formatters_agree_on_a_sweep(host):
    rng = np.random.default_rng(5)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 40000, dtype=np.uint64).astype(np.uint32),
                           # every decade boundary and its neighbours, where the notation switches
                           np.array([int(np.float32(10.0 ** k).view(np.uint32)) + d for k in range(-44, 39) for d in (-1, 0, 1)], np.int64).astype(np.uint32),
                           # integers around 2^24 .. 2^30: 8- and 9-digit values
                           rng.integers(9_000_000, 1_100_000_000, 20000).astype(np.float32).view(np.uint32)])
    xs = bits.view(np.float32)
    bad = [(float(x), _fmt_single(x), _cpp(host, x)) for x in xs if _fmt_single(x) != _cpp(host, x)]
    assert not bad, bad[:5]
    for x in xs[::97]:   # and the text reads back as the same float (round trip), whatever the notation
        if np.isfinite(x):
            assert np.float32(float(_fmt_single(x))) == x


def test_write_obj_sphere5_line_by_line(host, tmp_path):
    """The reference's Sphere5 scene (MarchingCubesTests.cs:31-45): the mesh from the oracle (no GPU here), written by both
    writers; `v`, then `vn`, then `f a//a b//b c//c`, 1-based (Mesh.cs:72-97)."""
    from oracle import oracle as O
    s = O.Scene()
    s.sphere_w(1.0)
    mn, mx = [-1.5] * 3, [1.5] * 3
    v, c = O.sample(s, mn, mx, 5, 5, 5)
    ref = O.march(v, c, mn, mx)
    assert len(ref.vertices) == 54
    V = np.ascontiguousarray(ref.vertices, np.float32)
    Nn = np.ascontiguousarray(ref.normals, np.float32)
    T = np.ascontiguousarray(ref.triangles, np.int32)
    m = Mesh(V, np.zeros_like(V), Nn, T)
    py_path, cc_path = str(tmp_path / "py.obj"), str(tmp_path / "cc.obj")
    m.WriteObj(py_path)
    assert host.write_obj(V.ctypes.data, Nn.ctypes.data, len(V), T.ctypes.data, len(T), cc_path.encode()) == 0
    py, cc = open(py_path).read().split("\n"), open(cc_path).read().split("\n")
    assert len(py) == len(cc) == 2 * len(V) + len(T) // 3 + 1
    for i, (a, b) in enumerate(zip(py, cc)):
        assert a == b, (i, a, b)
    assert py[0].startswith("v ") and py[len(V)].startswith("vn ") and py[2 * len(V)].startswith("f ")
    a, b, c3 = (int(t) + 1 for t in T[:3])
    assert py[2 * len(V)] == f"f {a}//{a} {b}//{b} {c3}//{c3}"
    # the text carries the float32 values exactly (shortest round-trip digits)
    back = np.array([[np.float32(float(t)) for t in ln.split()[1:]] for ln in py[:len(V)]], np.float32)
    assert np.array_equal(back, V)
    # and a file object works like a path (Mesh.cs:66-70: WriteObj(string) wraps WriteObj(TextWriter))
    import io
    w = io.StringIO()
    m.WriteObj(w)
    assert w.getvalue() == open(py_path).read()
