"""Constants of an SDF program are kernel arguments: one hiprtc compile per program STRUCTURE.

In the reference `Sdfs.Sphere(radius)` is a closure (Sdf.cs:202-214): a new radius costs nothing.  Here the generated source
-- hence the module and the on-disk cache entry -- is keyed by opcodes + operand ids + outputs; the constants travel with
every launch (csrc/sample_codegen.h).  Results stay bit-identical to the oracle for every constant."""
import ctypes as C
import time

import numpy as np
import pytest

import sdfkit_amd as K
from sdfkit_amd import SdfExprs, Sdfs, Voxels
from sdfkit_amd import _native as N
from oracle import oracle as O
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    N.init(0)
    return True


def _stats():
    a, b, c = C.c_int64(), C.c_int64(), C.c_double()
    N.check(N.lib().sdfk_jit_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def test_hundred_radii_one_compile(gpu, tmp_path):
    """100 spheres of different radii: hiprtc runs once; every mesh is the oracle's, bit for bit."""
    N.check(N.lib().sdfk_set_cache_dir(str(tmp_path / "jit").encode()))
    try:
        from sdfkit_amd.api import Sdf
        from sdfkit_amd.expr import MathF, Vec4
        mn, mx, n = [-1.5] * 3, [1.5] * 3, 40

        def sphere(r):
            # a sphere whose program no other test builds (`+ (x - x)` adds +0: the same values, another structure), so that
            # the counters below do not depend on what ran before in this process
            return Sdf(lambda p: Vec4.of((0, 0, 0), (MathF.Sqrt((p.x * p.x + p.y * p.y) + p.z * p.z) - r) + (p.x - p.x)), False)

        c0, h0, _ = _stats()
        worst = 0.0
        for i in range(100):
            r = 0.3 + 0.0107 * i          # (1.0 is not among them: it would not be baked either -- `len - r` has nothing to fold)
            t0 = time.perf_counter()
            mesh = sphere(r).ToMesh(mn, mx, n, n, n)
            nv = len(mesh.Vertices)
            dt = time.perf_counter() - t0
            if i > 0:
                worst = max(worst, dt)
            s = O.Scene(); s.sphere_w(r)
            ov, oc = O.sample(s, mn, mx, n, n, n)
            O.clip_to_bounds(ov, mn, mx)
            assert_mesh_equal(mesh, O.march(ov, oc, mn, mx))
            assert nv > 0
        c1, h1, _ = _stats()
        # ONE module for the hundred programs (a launch-bound grid always stores its volume, captured job or not: one kernel set;
        # with SDFK_GRAPHS=0 in the environment every grid takes the elided path: one set as well)
        assert (c1 - c0) + (h1 - h0) == 1, "one kernel set for the hundred programs"
        assert c1 - c0 == 1
        assert worst < 0.05, f"a new radius cost {worst * 1e3:.1f} ms"   # (incl. the Python tracer and a synchronous mesh read-back)
    finally:
        N.check(N.lib().sdfk_set_cache_dir(None))


def test_structure_is_the_key_not_the_constants(gpu):
    a, b = Sdfs.Sphere(0.7), Sdfs.Sphere(0.9)
    assert a.source() == b.source()
    assert "K.k[" in a.source()
    c = Sdfs.Box((0.5, 0.4, 0.3))
    assert c.source() != a.source()


def test_constants_that_fold_are_part_of_the_structure(gpu):
    """x / 2^k, x * +-1, x - 0 ... fold exactly when the constant is a literal: those stay literals (the structure
    changes with them), results are the oracle's either way."""
    mn, mx, dims = [-2.0] * 3, [2.0] * 3, (36, 40, 44)
    for period in (2.0, 1.5, 1.0, 0.75):
        sdf = SdfExprs.Sphere(0.4375, (0.25, 0.5, 0.75)).RepeatX(period).ToSdf()
        scene = O.Scene()
        scene.f_repeat_x(scene.f_sphere(0.4375, (0.25, 0.5, 0.75)), period)
        ov, oc = O.sample(scene, mn, mx, *dims)
        vol = Voxels.SampleSdf(sdf, mn, mx, *dims)
        assert np.array_equal(vol.Values, ov) and np.array_equal(vol.Colors, oc), period
    s2 = SdfExprs.Sphere(0.4375, (0.25, 0.5, 0.75)).RepeatX(2.0).ToSdf().source()
    s15 = SdfExprs.Sphere(0.4375, (0.25, 0.5, 0.75)).RepeatX(1.5).ToSdf().source()
    s075 = SdfExprs.Sphere(0.4375, (0.25, 0.5, 0.75)).RepeatX(0.75).ToSdf().source()
    assert s15 == s075            # ordinary constants: arguments
    assert s2 != s15              # a power-of-two divisor: a literal (the division becomes a multiplication)


def test_special_values_as_arguments(gpu):
    """NaN, infinities, signed zeros and denormals arrive through the argument block unchanged."""
    from sdfkit_amd.api import Sdf
    from sdfkit_amd.expr import Vec4
    mn, mx, dims = [-1.0] * 3, [1.0] * 3, (8, 8, 8)
    for c in (float("nan"), float("inf"), -float("inf"), -0.0, 0.0, 1e-42, -1e-42, 3.4028234663852886e38):
        sdf = Sdf(lambda p, c=c: Vec4.of((0, 0, 0), p.x * 0.0 + (p.y - p.y) + np.float32(c)), False)
        vol = Voxels.SampleSdf(sdf, mn, mx, *dims)
        v = vol.Values.ravel()
        want = np.float32(0.0) + np.float32(c)
        if np.isnan(want):
            assert np.isnan(v).all()
        else:
            # (x * 0 + 0 is +0 or -0 depending on the sign of x: compare through the same float32 arithmetic)
            px = (np.float32(-1.0) + np.float32(0.5) * np.float32(0.25)) + np.arange(8, dtype=np.float32) * np.float32(0.25)
            ref = (px * np.float32(0.0) + np.float32(0.0)) + np.float32(c)
            got = vol.Values[:, 0, 0]
            assert np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32)), c


def test_programs_with_many_constants_keep_literals(gpu):
    """More than 28 constants: all literals (scalar-register budget; csrc/sample_codegen.h) -- the 8-primitive union of
    BASELINE config C4 is such a program; a 3-primitive union is not."""
    import bench
    big = bench.scene_for("union8")[0]
    assert "K.k[0]" not in big.source() and "struct SdfkK { float k[1]; }" in big.source()
    a = (SdfExprs.Union(SdfExprs.Sphere(0.6).Translate(-1, 0, 0), SdfExprs.Union(SdfExprs.Box(0.5).Translate(1, 0, 0), SdfExprs.Cylinder(0.4, 0.6)))).ToSdf()
    b = (SdfExprs.Union(SdfExprs.Sphere(0.55).Translate(-1.25, 0, 0), SdfExprs.Union(SdfExprs.Box(0.45).Translate(1.5, 0, 0), SdfExprs.Cylinder(0.35, 0.65)))).ToSdf()
    assert a.source() == b.source() and "K.k[" in a.source()
