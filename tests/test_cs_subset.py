"""tools/cs_subset.py -- the interpreter that executes the reference's C# for the golden vectors -- held to the C# language rules it
relies on (ECMA-334: numeric promotions, truncating integer division, implicit conversions, struct copy semantics, closures,
overload resolution, extension methods).  Snippets written for this test; nothing here comes from the reference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cs_subset as CS  # noqa: E402

F32 = np.float32


class V3Host:
    Zero = CS.Vec3()
    One = CS.Vec3(1, 1, 1)


def run(body, ret="double", hosts=None, extra=""):
    src = "public static class T { " + extra + " public static " + ret + " F() { " + body + " } }"
    it = CS.Interp(CS.parse(src), dict({"Vector3": V3Host}, **(hosts or {})))
    return it.call_static("T", "F", [])


def test_integer_arithmetic_truncates_toward_zero():
    assert run("return -7 / 2;", "int") == -3 and run("return 7 / -2;", "int") == -3 and run("return -7 % 3;", "int") == -1
    assert run("int i = 17; return (i / 5) % 3;", "int") == 0


def test_binary_numeric_promotion():
    v = run("float a = 0.1f; double b = 0.2; return a + b;")
    assert isinstance(v, float) and v == float(F32(0.1)) + 0.2                       # float + double -> double
    v = run("float a = 0.1f; float b = 0.2f; return a * b;", "float")
    assert isinstance(v, F32) and v == F32(F32(0.1) * F32(0.2))                       # float * float -> float, rounded once
    v = run("int i = 3; float d = 0.3f; return i * d;", "float")
    assert isinstance(v, F32) and v == F32(F32(3) * F32(0.3))                         # int * float -> float
    v = run("int i = 3; return i / 2f;", "float")
    assert v == F32(1.5)
    assert run("int n = 7; return (float)3 / n;", "float") == F32(F32(3) / F32(7))    # cast binds tighter than '/'


def test_implicit_conversions_on_assignment_and_return():
    assert run("double d = 3; return d / 2;") == 1.5                                  # int -> double on declaration
    assert run("double t; t = 0; t += 1; return t / 4;") == 0.25                      # ... and on assignment
    v = run("float f = 1; return f;", "double")
    assert isinstance(v, float) and v == 1.0                                          # float -> double on return
    with pytest.raises(TypeError):
        run("double d = 0.5; float f = d; return f;")                                 # double -> float needs a cast


def test_casts_round_to_nearest_even():
    assert run("double d = 0.1; return (float)d;", "float") == F32(0.1)
    assert run("double d = 16777217.0; return (float)d;", "float") == F32(16777216.0)  # tie -> even
    assert run("return (int)2.9;", "int") == 2 and run("return (int)-2.9;", "int") == -2


def test_vector3_is_a_struct():
    # assignment copies; mutating the copy leaves the original alone
    assert run("Vector3 a = new Vector3(1f, 2f, 3f); Vector3 b = a; b.X = 9f; return a.X + b.X;", "float") == F32(10)
    # array elements are variables: element.field = ... mutates in place
    assert run("var arr = new Vector3[2]; arr[1].Y = 5f; var c = arr[1]; c.Y = 7f; return arr[1].Y;", "float") == F32(5)
    assert run("var v = new Vector3(2f); return v.X + v.Y + v.Z;", "float") == F32(6)


def test_vector3_arithmetic_is_componentwise_float():
    v = run("var a = new Vector3(0.1f, 0.2f, 0.3f); var b = a * 3f + Vector3.One; return b.Y;", "float")
    assert v == F32(F32(F32(0.2) * F32(3)) + F32(1))
    with pytest.raises(TypeError):
        run("var a = new Vector3(1f, 1f, 1f); var b = a * 0.5; return b.X;", "float")   # Vector3 * double does not exist


def test_closures_capture_variables_not_values():
    body = "int k = 1; System.Func<int, int> f = x => x + k; k = 10; return f(1);"
    # (declaring the delegate type is outside the subset: the variable is `var` in the files this tool reads)
    assert run(body.replace("System.Func<int, int> f", "var f"), "int") == 11


def test_overloads_and_extension_methods():
    extra = ("public static int G(int a) { return 1; } public static int G(int a, int b) { return 2; } "
             "public static int G(Vector3 v) { return 3; } public static int Twice(this Vector3 v, int k) { return 2 * k; }")
    assert run("return G(1) * 100 + G(1, 2) * 10 + G(new Vector3(1f, 1f, 1f));", "int", extra=extra) == 123
    assert run("var v = new Vector3(1f, 1f, 1f); return v.Twice(21);", "int", extra=extra) == 42


def test_ternary_tuples_loops_and_compound_assignment():
    assert run("int a = 0, b = 0; (a, b) = (3, 4); return a < b ? a : b;", "int") == 3
    assert run("int s = 0; for (var i = 0; i < 5; ++i) { s += i; } int j = 0; while (j < 3) { j++; s -= 1; } return s;", "int") == 7
    assert run("double x = 8; x /= 2; x *= 3; x -= 1; return x;") == 11.0


def test_properties_and_fields_of_an_instance():
    src = ("public class P { int n; public float Scale { get; private set; } public int Twice => 2 * n; "
           "public P(int n) { this.n = n; Scale = 0.5f; } public float Apply(float x) { return x * Scale; } } "
           "public static class T { public static float F() { var p = new P(21); return p.Twice + p.Apply(4f); } }")
    it = CS.Interp(CS.parse(src), {})
    assert it.call_static("T", "F", []) == F32(44)


def test_ieee_special_values_flow_through():
    assert np.isinf(run("double z = 0.0; return 1.0 / z;"))
    assert np.isnan(run("double z = 0.0; return z / z;"))
    assert run("double z = 0.0; double n = z / z; return n > 0.0 ? 1.0 : 2.0;") == 2.0   # comparisons with NaN are false
