"""RayMarcher.

CPU part: the oracle against the known answers of the reference's Tests/RayMarcherTests.cs, and
the host-side camera arithmetic of the product mirror against the oracle's (bit for bit).
GPU part (-m gpu): the JIT sphere-tracing kernel, through the C ABI, against the oracle --
depth and colour images bit-exact (float32 arithmetic restated op for op; the reference's own
tests pin only single pixels with tolerances: finer parity is oracle <-> HIP only)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import scenes as S


def _depth(scene, w=50, h=30, **kw):
    return O.raymarch(scene, w, h, want_rgb=False, **kw)[0]


# ---- the reference's known answers (FloatData indexer is [x, y]) --------------------------
def test_sphere_depth():  # RayMarcherTests.cs:10-24
    s = O.Scene(); s.sphere_w(1.0)
    d = _depth(s)
    assert d.shape == (30, 50)
    assert abs(d[15, 25] - 4.0) <= 1.0e-2
    assert d[0, 0] > 9.0


def test_box_depth():  # RayMarcherTests.cs:27-41
    s = O.Scene(); s.box_w(1.0)
    d = _depth(s)
    assert abs(d[15, 25] - 4.0) <= 1.0e-2
    assert d[0, 0] > 9.0


def test_cylinder_depth():  # RayMarcherTests.cs:44-62: Cylinder(r, 2r).RepeatX(4r), pixel [w/2, h/2-2]
    r = 0.25
    s = O.Scene(); s.root = s.f_repeat_x(s.f_cylinder(r, 2 * r), 4 * r)
    d = _depth(s)
    assert abs(d[13, 25] - (5 - r)) <= 1.0e-1
    assert d[0, 0] > 9.0


def test_plane_depth():  # RayMarcherTests.cs:65-78: Sdfs.PlaneXY()
    s = O.Scene(); s.plane_w(0, 0, 1, 0.0)
    d = _depth(s)
    assert abs(d[15, 25] - 5.0) <= 1.0e-2
    assert d[0, 0] < 9.0


def test_render_is_sky_where_nothing_is_hit_and_lit_where_something_is():
    scene, _ = S.readme_repeat_xy()
    view = O.look_at((-2, 2, 4), (0, 0, 0), (0, 1, 0))      # TimeRender, RayMarcherTests.cs:131-134
    depth, rgb = O.raymarch(scene, 96, 54, view=view)
    sky = depth > 100.0
    assert sky.any() and (~sky).any()
    assert np.array_equal(rgb[sky], np.broadcast_to(np.float32([0.5, 0.75, 1.0]), rgb[sky].shape))
    lit = rgb[~sky]
    assert np.isfinite(lit).all() and lit.max() <= 1.0 + 1e-6      # 0.1 ambient + diffuse * colour (<= 0.9)
    assert (lit[:, 0] > 0.1).any()                                  # something is actually lit


# ---- host-side camera arithmetic of the product mirror == the oracle's --------------------
@pytest.mark.parametrize("cam", [((0, 0, 5), (0, 0, 0), (0, 1, 0), 50, 30, 60.0, 1.0, 100.0),
                                 ((-2, 2, 4), (0, 0, 0), (0, 1, 0), 192, 108, 60.0, 1.0, 100.0),
                                 ((3, -1, 2.5), (0.2, 0.1, -0.3), (0, 0, 1), 1920, 1080, 45.0, 0.5, 250.0)])
def test_camera_math_matches_oracle(cam):
    from sdfkit_amd import Matrix4x4, RayMarcher
    pos, tgt, up, w, h, fov, near, far = cam
    view = Matrix4x4.CreateLookAt(pos, tgt, up)
    assert np.array_equal(view, O.look_at(pos, tgt, up))
    rm = RayMarcher(w, h, None)
    rm.ViewTransform, rm.VerticalFieldOfViewDegrees, rm.NearPlaneDistance, rm.FarPlaneDistance = view, fov, near, far
    c, vpi = rm.camera()
    oc, ovpi = O.ray_camera(view, fov, w, h, near, far)
    assert np.array_equal(c, oc) and np.array_equal(vpi, ovpi)


def test_tga_writers(tmp_path):
    from sdfkit_amd import FloatData, Vec3Data
    d = FloatData(np.float32([[2.0, 3.0, 6.5], [10.0, 11.0, 4.0]]))
    d.SaveDepthTga(str(tmp_path / "d.tga"), 3, 10)          # VectorData.cs:244-279
    raw = (tmp_path / "d.tga").read_bytes()
    assert raw[:18] == bytes([0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 2, 0, 8, 0x20])
    assert list(raw[18:]) == [255, 255, int(255.0 * 3.5 / 7.0), 0, 0, int(np.float32(255.0) * np.float32(6.0) / np.float32(7.0))]
    c = Vec3Data(np.float32([[[0.5, 0.75, 1.0], [-1.0, 2.0, 0.1]]]))
    c.SaveTga(str(tmp_path / "c.tga"))                      # VectorData.cs:570-619: B, G, R
    raw = (tmp_path / "c.tga").read_bytes()
    assert raw[:18] == bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 1, 0, 24, 0x20])
    assert list(raw[18:]) == [255, 191, 127, 25, 255, 0]
    assert d[2, 0] == np.float32(6.5) and d.Width == 3 and d.Height == 2      # indexer is [x, y]


# ---- GPU parity ------------------------------------------------------------------------------
GPU_SCENES = ["sphere_w", "box_w", "plane_w", "cylinder", "readme_repeat_xy", "union8", "sdf_with_color", "repeat_xz_box"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_SCENES)
def test_gpu_render_and_depth_equal_oracle(gpu, name):
    from sdfkit_amd import Matrix4x4, RayMarcher
    scene, sdf = S.CATALOGUE[name]()
    for (w, h, cam, iters) in [(50, 30, None, 40), (97, 61, ((-2, 2, 4), (0, 0, 0), (0, 1, 0)), 25)]:
        rm = RayMarcher(w, h, sdf)
        rm.DepthIterations = iters
        view = None
        if cam:
            rm.ViewTransform = Matrix4x4.CreateLookAt(*cam)
            view = O.look_at(*cam)
        od, orgb = O.raymarch(scene, w, h, view=view, iterations=iters)
        d = rm.RenderDepth()
        assert (d.Width, d.Height) == (w, h)
        assert np.array_equal(d.Values, od, equal_nan=True), f"{name}: depth differs (max {np.nanmax(np.abs(d.Values - od))})"
        img = rm.Render()
        assert np.array_equal(img.Values, orgb, equal_nan=True), f"{name}: colour differs (max {np.nanmax(np.abs(img.Values - orgb))})"


@pytest.mark.gpu
def test_gpu_reference_known_answers(gpu):
    """Tests/RayMarcherTests.cs through the product mirror."""
    from sdfkit_amd import RayMarcher, SdfExprs, Sdfs
    w, h = 50, 30
    img = RayMarcher(w, h, Sdfs.Sphere(1.0)).RenderDepth()
    assert (img.Width, img.Height) == (w, h)
    assert abs(img[w // 2, h // 2] - 4.0) <= 1.0e-2 and img[0, 0] > 9.0
    img = RayMarcher(w, h, Sdfs.Box(1.0)).RenderDepth()
    assert abs(img[w // 2, h // 2] - 4.0) <= 1.0e-2 and img[0, 0] > 9.0
    r = 0.25
    img = RayMarcher(w, h, SdfExprs.Cylinder(r, r * 2).RepeatX(4 * r).ToSdf()).RenderDepth()
    assert abs(img[w // 2, h // 2 - 2] - (5 - r)) <= 1.0e-1 and img[0, 0] > 9.0
    img = RayMarcher(w, h, Sdfs.PlaneXY()).RenderDepth()
    assert abs(img[w // 2, h // 2] - 5.0) <= 1.0e-2 and img[0, 0] < 9.0
    # SphereRepeat (RayMarcherTests.cs:96-108) via SdfEx.ToImage
    from sdfkit_amd import Vec3
    sdf = SdfExprs.Sphere(0.5).RepeatXY(1.125, 1.125, lambda i, p, d: 0.9 * Vec3.of(p.x.b, 1.0) - Vec3.Abs(i) / 6.0).ToSdf()
    im = sdf.ToImage(192, 108, (-2, 2, 4), (0, 0, 0), (0, 1, 0))
    assert (im.Width, im.Height) == (192, 108) and np.isfinite(im.Values).all()


@pytest.mark.gpu
def test_gpu_config_c5_full_hd_properties(gpu):
    """BASELINE config C5: 1920x1080, 256 steps, RepeatXY scene, camera (-2,2,4) -> origin.  Rows of
    the full-size frame against the oracle (all host threads: a few seconds on the GPU box)."""
    from sdfkit_amd import Matrix4x4, RayMarcher
    scene, sdf = S.readme_repeat_xy()
    rm = RayMarcher(1920, 1080, sdf)
    rm.DepthIterations = 256
    rm.ViewTransform = Matrix4x4.CreateLookAt((-2, 2, 4), (0, 0, 0), (0, 1, 0))
    img = rm.Render().Values
    dep = rm.RenderDepth().Values
    sky = dep > 100.0
    assert (~sky).mean() > 0.5      # (sky pixels whose depth ran off to infinity are NaN: 0 * inf, as in the reference)
    od, orgb = O.raymarch(scene, 1920, 1080, view=O.look_at((-2, 2, 4), (0, 0, 0), (0, 1, 0)), iterations=256)
    assert np.array_equal(dep, od, equal_nan=True) and np.array_equal(img, orgb, equal_nan=True)
