"""SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points -- sdfk_eval_points.

Pinned by the EXECUTED reference: tests/golden/reference_sdf_points.npz holds what the reference's own SdfFuncs / SdfFuncEx lambdas and
expression trees return at 611 points per scene (period seams, signed zeros, denormals; tools/gen_reference_sdf_vectors.py); the GPU
must return the same (r, g, b, w), bit for bit.  And by the oracle (orc_eval) on the test catalogue at random points."""
import ctypes as C
import json

import numpy as np
import pytest

from sdfkit_amd import _native as N
from oracle import oracle as O
from tests import scenes as S
from tests.test_reference_vectors import SDF, SDF_NAMES, mirror_sdf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    N.init(0)
    return True


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", SDF_NAMES)
def test_sample_equals_the_executed_reference(gpu, name):
    descr = json.loads(str(SDF["scenes_json"]))[name]
    sdf = mirror_sdf(descr).ToSdf()
    pts, want = SDF[f"{name}/points"], SDF[f"{name}/rgbw"]
    sentinel = np.full((len(pts), 4), 7.25, np.float32)
    got = sdf.Sample(pts, sentinel)
    assert got is sentinel
    assert np.array_equal(_bits(got[:, 3]), _bits(want[:, 3])), name
    if sdf.writes_color:
        assert np.array_equal(_bits(got[:, :3]), _bits(want[:, :3])), name
    else:      # a delegate that only assigns .W leaves X, Y, Z of the caller's elements alone (Sdf.cs:211)
        assert (got[:, :3] == 7.25).all()
    # a fresh output array: zeros where nothing is written
    fresh = sdf.Sample(pts)
    assert np.array_equal(_bits(fresh[:, 3]), _bits(want[:, 3]))
    if not sdf.writes_color:
        assert not fresh[:, :3].any()


@pytest.mark.parametrize("name", sorted(S.CATALOGUE))
def test_sample_equals_the_oracle_at_random_points(gpu, name):
    scene, sdf = S.CATALOGUE[name]()
    rng = np.random.default_rng(11)
    n = 20011                                   # (not a multiple of the workgroup size)
    pts = rng.uniform(-3.0, 3.0, (n, 3)).astype(np.float32)
    pts[:7] = [[0, 0, 0], [-0.0, 0.0, -0.0], [0.5625, 0.5625, 0], [-0.5625, 0, 0.5625], [1e-39, -1e-39, 1e-41], [2.8125, -2.8125, 2.8125], [1.125, 1.125, 1.125]]
    got = sdf.Sample(pts)
    want = np.stack([O.eval_point(scene, p) for p in pts[:3000]])
    assert np.array_equal(_bits(got[:3000, 3]), _bits(want[:, 3])), name
    if sdf.writes_color:
        assert np.array_equal(_bits(got[:3000, :3]), _bits(want[:, :3])), name
    # the same points through the grid sampler's arithmetic elsewhere: the device form, caller-owned buffers
    import torch
    dp = torch.from_numpy(pts).cuda()
    do = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    N.check(N.lib().sdfk_eval_points_device(sdf.program(), C.c_void_p(dp.data_ptr()), n, C.c_void_p(do.data_ptr())))
    N.check(N.lib().sdfk_synchronize())
    assert np.array_equal(_bits(do.cpu().numpy()), _bits(got if sdf.writes_color else np.concatenate([np.zeros((n, 3), np.float32), got[:, 3:]], axis=1)))


def test_sample_argument_checks(gpu):
    L = N.lib()
    _, sdf = S.sphere_w(1.0)
    p = sdf.program()
    out = np.zeros((4, 4), np.float32)
    pts = np.zeros((4, 3), np.float32)
    assert L.sdfk_eval_points(p, None, 4, out.ctypes.data) == N.ERR_INVALID
    assert L.sdfk_eval_points(p, pts.ctypes.data, 4, None) == N.ERR_INVALID
    assert L.sdfk_eval_points(p, pts.ctypes.data, -1, out.ctypes.data) == N.ERR_INVALID
    assert L.sdfk_eval_points(None, pts.ctypes.data, 4, out.ctypes.data) == N.ERR_INVALID
    assert L.sdfk_eval_points(p, None, 0, None) == 0          # nothing to do
    with pytest.raises(ValueError):
        sdf.Sample(pts, np.zeros((4, 3), np.float32))
