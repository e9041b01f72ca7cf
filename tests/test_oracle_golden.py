"""Pins the CPU oracle against every known-answer test the reference holds for the path
(SURVEY.md section 8c).  Expected numbers are the literals asserted by the reference's
NUnit tests; file:line given per case."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import scenes as S


def _mesh(scene, mn, mx, n, clip=False, iso=0.0, step=1, progress=None):
    if isinstance(n, int):
        n = (n, n, n)
    v, c = O.sample(scene, mn, mx, *n)
    if clip:
        O.clip_to_bounds(v, mn, mx)
    return O.march(v, c, mn, mx, iso, step, progress)


def _len(v):
    return float(np.sqrt(np.sum(np.asarray(v, np.float64) ** 2)))


def test_colored_spheres():  # Tests/MarchingCubesTests.cs:11-28
    m = _mesh(S.colored_spheres()[0], [-3] * 3, [3] * 3, 32)
    assert len(m.vertices) == 104 and len(m.colors) == 104
    assert m.colors[0][0] > 0.5


def test_sphere5():  # MarchingCubesTests.cs:31-45
    m = _mesh(S.sphere_w(1.0)[0], [-1.5] * 3, [1.5] * 3, 5)
    assert len(m.vertices) == 54
    assert _len(m.center) <= 1e-6
    assert abs(m.size[0] / 2 - 1.0) <= 0.3


def test_sphere10():  # MarchingCubesTests.cs:48-62
    m = _mesh(S.sphere_w(2.0)[0], [-2.5] * 3, [2.5] * 3, 10)
    assert len(m.vertices) == 312
    assert _len(m.center) <= 1e-6
    assert abs(m.size[0] / 2 - 2.0) <= 0.2


def test_unclipped_sphere10():  # MarchingCubesTests.cs:65-79
    m = _mesh(S.sphere_w(2.0)[0], [-1] * 3, [1] * 3, 10)
    assert len(m.vertices) == 0 and len(m.triangles) == 0


def test_clipped_sphere10():  # MarchingCubesTests.cs:82-98
    m = _mesh(S.sphere_w(2.0)[0], [-1] * 3, [1] * 3, 10, clip=True)
    assert len(m.vertices) == 384
    assert _len(m.center) <= 1e-6
    assert abs(m.size[0] - 2.0) <= 1e-1


def test_box10():  # MarchingCubesTests.cs:101-115
    m = _mesh(S.box_w(2.0)[0], [-2.5] * 3, [2.5] * 3, 10)
    assert len(m.vertices) == 384
    assert _len(m.center) <= 1e-6
    assert abs(m.size[0] / 2 - 2.0) <= 0.3


def test_cylinder50():  # MarchingCubesTests.cs:118-138
    m = _mesh(S.cylinder(1, 3)[0], [-1.5, -3.5, -1.5], [1.5, 3.5, 1.5], 50)
    assert len(m.vertices) == 7456
    assert np.all(np.abs(m.center) <= 1e-6)
    assert abs(m.size[0] / 2 - 1.0) <= 0.1


def test_sphere128_progress():  # MarchingCubesTests.cs:141-171
    got = []
    m = _mesh(S.sphere_w(3.0)[0], [-3.1] * 3, [3.1] * 3, 128, progress=got.append)
    assert len(m.vertices) == 72240
    assert all(0.0 <= f <= 1.0 for f in got)
    assert any(f < 1e-6 for f in got) and any(1.0 - f < 1e-6 for f in got)
    assert _len(m.center) <= 1e-6
    assert abs(m.size[0] / 2 - 3.0) <= 0.1


def test_create_mesh_sphere():  # Tests/SdfTests.cs:29-39 (ToMesh, clipToBounds default true)
    m = _mesh(S.sphere_w(0.5)[0], [-1] * 3, [1] * 3, 32, clip=True)
    assert len(m.vertices) == 1248


def test_solid_sphere():  # SdfTests.cs:42-52: expression path must agree with the delegate path
    m = _mesh(S.solid_sphere(0.5)[0], [-1] * 3, [1] * 3, 32, clip=True)
    assert len(m.vertices) == 1248


def test_volume_dims_and_size():  # Tests/VolumeTests.cs:11-38
    d = O.cell_size([-1] * 3, [1] * 3, 5, 7, 11)
    assert np.allclose(d * np.array([5, 7, 11], np.float32), 2.0)


def test_one_is_centered():  # VolumeTests.cs:41-58
    p = O.sample_position([-1] * 3, [1] * 3, 1, 1, 1, 0)
    assert _len(p) < 1e-3
    s = O.Scene(); s.f_const(0, 0, 0, 1)
    v, _ = O.sample(s, [-1] * 3, [1] * 3, 1, 1, 1)
    assert v[0, 0, 0] == 1


def test_three_has_center():  # VolumeTests.cs:61-80
    assert any(_len(O.sample_position([-1] * 3, [1] * 3, 3, 3, 3, i)) < 1e-3 for i in range(27))


def test_sphere_center_values():  # VolumeTests.cs:83-106, SdfTests.cs:12-26
    v, _ = O.sample(S.sphere_w(0.5)[0], [-1] * 3, [1] * 3, 5, 5, 5)
    assert abs(v[2, 2, 2] + 0.5) <= 1e-3
    v, _ = O.sample(S.solid_sphere(0.5)[0], [-1] * 3, [1] * 3, 128, 128, 128)
    O.clip_to_bounds(v, [-1] * 3, [1] * 3)
    assert abs(v[63, 63, 63] + 0.5) <= 2e-2


def test_batch_slicing():  # VolumeTests.cs:109-135: n == 70 everywhere except the tail n == 22
    sizes = O.batch_sizes(128 ** 3, 70)
    assert set(sizes[:-1]) == {70} and sizes[-1] == 22


def test_threads_do_not_change_result():
    s = S.readme_repeat_xy()[0]
    a, ca = O.sample(s, [-2.8125] * 3, [2.8125] * 3, 24, 20, 28, threads=1)
    b, cb = O.sample(s, [-2.8125] * 3, [2.8125] * 3, 24, 20, 28, threads=4, batch=70)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)


def test_survey_config_counts():
    # the survey measured V = 8616 as the number of sign-changing grid edges
    m = _mesh(S.sphere_w(1.0)[0], [-1.5] * 3, [1.5] * 3, 64)
    assert len(m.vertices) == 8616
    v, _ = O.sample(S.sphere_w(1.0)[0], [-1.5] * 3, [1.5] * 3, 64, 64, 64)
    s = v > 0
    edges = (s[1:] != s[:-1]).sum() + (s[:, 1:] != s[:, :-1]).sum() + (s[:, :, 1:] != s[:, :, :-1]).sum()
    assert edges == 8616


@pytest.mark.parametrize("seed", range(4))
def test_mesh_is_consistent_on_random_volumes(seed):
    """Structural invariants the serial algorithm guarantees (oracle-only, unpinned by the
    reference): every index valid, every vertex referenced, first references ascending."""
    rng = np.random.default_rng(seed)
    v = rng.uniform(-1, 1, (13, 11, 9)).astype(np.float32)
    c = rng.uniform(0, 1, (13, 11, 9, 3)).astype(np.float32)
    m = O.march(v, c, [-1] * 3, [1] * 3)
    t = m.triangles
    assert t.min() >= 0 and t.max() == len(m.vertices) - 1
    first = np.full(len(m.vertices), -1)
    seen = 0
    for i in t:
        if first[i] < 0:
            assert i == seen  # vertices are numbered in order of first reference
            first[i] = 1
            seen += 1
    assert seen == len(m.vertices)
