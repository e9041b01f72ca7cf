"""SDFK_OPT_ELIDE_VOLUME: sdfk_sample_march (SdfEx.ToMesh, Sdf.cs:59-63 -- the Voxels is a temporary) without storing the volume.

The sampler leaves the sign bits only (sdfk_sample_signs), the corners of the active cells and the vertex colours are
re-evaluated by the program; every mesh must be the oracle's, bit for bit, exactly as with the stored volume."""
import ctypes as C

import numpy as np
import pytest

from sdfkit_amd import Sdfs
from sdfkit_amd import _native as N
from sdfkit_amd.api import Sdf
from sdfkit_amd.expr import MathF, Vec4
from oracle import oracle as O
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    N.init(0)
    return True


def _kernels_launched(fn):
    L = N.lib()
    N.check(L.sdfk_profile_reset())
    N.check(L.sdfk_profile_enable(1))
    try:
        out = fn()
    finally:
        N.check(L.sdfk_profile_enable(0))
    return out, {k for k, v in N.profile_snapshot().items() if v[1]}


# grids above the captured-graph limit (2^24 voxels) take the elided path; every row shape of the sampler (z tiles, plane chunks)
@pytest.mark.parametrize("name", sorted(S.CATALOGUE))
@pytest.mark.parametrize("dims", [(264, 260, 256), (300, 236, 250)])
def test_elided_mesh_is_the_oracles(gpu, name, dims):
    scene, sdf = S.CATALOGUE[name]()
    mn, mx = [-2.8125] * 3, [2.8125] * 3
    ov, oc = O.sample(scene, mn, mx, *dims)
    O.clip_to_bounds(ov, mn, mx)
    om = O.march(ov, oc, mn, mx)
    with N.option(N.OPT_ELIDE_VOLUME, 1):
        for rep in range(3):      # exact two-phase path first, then the speculative one
            mesh, ran = _kernels_launched(lambda: sdf.ToMesh(mn, mx, *dims))
            assert_mesh_equal(mesh, om)
            assert any(k.startswith("sdfk_sample_signs") for k in ran) and not any(k.startswith("sdfk_sample_bits") for k in ran), ran
            assert "k_gather_corners" not in ran and "k_signbits" not in ran
    with N.option(N.OPT_ELIDE_VOLUME, 0):                                 # the default: the volume is stored
        mesh, ran = _kernels_launched(lambda: sdf.ToMesh(mn, mx, *dims))
        assert_mesh_equal(mesh, om)
        assert any(k.startswith("sdfk_sample_bits") for k in ran)


def test_no_clip_and_other_iso(gpu):
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (272, 264, 256)
    s = O.Scene(); s.sphere_w(1.0)
    ov, oc = O.sample(s, mn, mx, *dims)
    for iso in (0.0, 0.125):
        om = O.march(ov, oc, mn, mx, iso=iso)
        with N.option(N.OPT_ELIDE_VOLUME, 1):
            for rep in range(2):
                assert_mesh_equal(Sdfs.Sphere(1.0).ToMesh(mn, mx, *dims, clipToBounds=False, isoValue=iso), om)


def test_case_13_sign_words_fall_back_to_a_stored_volume(gpu):
    """A sphere with a small 3-D checkerboard block inside: the cells of the block have the sign words 0xA5 / 0x5A, for which the
    dead-cell test of k_resolve must read voxels.  The elided job notices (n_case13 in the mirrored counters), the volume gets
    its storage and is sampled again with stores; later volumes of the program are stored from the start."""
    from sdfkit_amd.expr import select_lt, trace
    from oracle import ir_interp as I
    n = 264

    def field(p):
        f = lambda t: MathF.Floor((t + 1.0) * (n / 2.0))          # the voxel index along one axis (sample points sit at index + 0.5)
        s = f(p.x) + f(p.y) + f(p.z)
        odd = s - MathF.Floor(s * 0.5) * 2.0                       # 0 or 1
        block = MathF.Max(MathF.Max(MathF.Abs(p.x), MathF.Abs(p.y)), MathF.Abs(p.z))
        sphere = MathF.Sqrt((p.x * p.x + p.y * p.y) + p.z * p.z) - 0.7
        return Vec4.of((0, 0, 0), select_lt(block, 0.05, (odd - 0.5) * (0.25 + p.x), sphere))

    sdf = Sdf(field, False)
    mn, mx, dims = [-1.0] * 3, [1.0] * 3, (n, n, n)
    ops, out = trace(field, False)
    ov, oc = I.sample(ops, out, False, mn, mx, *dims)                 # the same float32 program, op for op
    om = O.march(ov, oc, mn, mx)
    assert len(om.vertices) > 100000
    with N.option(N.OPT_ELIDE_VOLUME, 1):
        for rep in range(3):
            mesh, ran = _kernels_launched(lambda: sdf.ToMesh(mn, mx, *dims, clipToBounds=False))
            assert_mesh_equal(mesh, om)
            if rep == 0:
                assert any(k.startswith("sdfk_sample_signs") for k in ran) and any(k.startswith("sdfk_sample_bits") for k in ran), ran
            else:
                assert not any(k.startswith("sdfk_sample_signs") for k in ran), ran      # stored from the start now
