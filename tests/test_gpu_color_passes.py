"""SDFK_OPT_COLOR_PASSES: a colour volume (Voxels.SampleSdf of a program that assigns r, g, b: Voxels.cs:112-120, 16 B per voxel into two
arrays) sampled in ONE fused pass or in TWO -- values + sign bytes by the fused kernel without its colour half (sdfk_sample_bits_nc*),
then the colour array as one linear stream (sdfk_sample_colors, csrc/sample_codegen.h).  Whatever the option says, Values and Colors are
the oracle's bit for bit, and so is the mesh built from the first pass's sign bits -- for every shape of the sampler (z tiles of one
row, chunks of the (y, z) plane, rows and planes shorter than a workgroup's 256 voxels) and with ClipToBounds."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import _native as N
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COLOUR_SCENES = ["readme_repeat_xy", "colored_spheres", "union8", "cylinder", "repeat_xz_box"]
SHAPES = [((32, 32, 32), False), ((17, 23, 29), True), ((10, 12, 260), True), ((9, 7, 516), False), ((16, 5, 256), True), ((8, 6, 768), False),
          ((9, 4, 7), True), ((5, 3, 257), True), ((3, 2, 5), False), ((40, 12, 36), True), ((7, 70, 4), True)]


@pytest.mark.parametrize("name", COLOUR_SCENES)
def test_two_passes_give_the_same_volume_and_mesh(gpu, name):
    scene, sdf = S.CATALOGUE[name]()
    assert sdf.writes_color
    mn, mx = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]
    for dims, clip in SHAPES:
        ov, oc = O.sample(scene, mn, mx, *dims)
        if clip:
            O.clip_to_bounds(ov, mn, mx)
        om = O.march(ov, oc, mn, mx)
        for passes in (2, 1):
            with N.option(N.OPT_COLOR_PASSES, passes):
                v = sdf.ToVoxels(mn, mx, *dims, clipToBounds=clip)
                assert np.array_equal(v.Values, ov), (name, dims, passes, "values")
                assert np.array_equal(v.Colors, oc), (name, dims, passes, "colours", int(np.count_nonzero(v.Colors != oc)))
                assert_mesh_equal(v.ToMesh(), om)
                with N.option(N.OPT_VCOLOR_EVAL, 0):      # vertex colours GATHERED from the colour volume instead of re-evaluated
                    assert_mesh_equal(v.ToMesh(), om)
                with N.option(N.OPT_ELIDE_VOLUME, 0):     # SdfEx.ToMesh with its temporary volume STORED: the same two passes inside the fused call
                    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims, clipToBounds=clip), om)


def test_the_default_picks_two_passes_for_a_tiny_program_on_a_large_grid(gpu):
    """Option 0: by the program's size and the grid's.  One primitive with a constant colour (Sdfs.Cylinder: 16 operations) on 128 x 128 x 256
    (2^22 voxels) takes the two-pass route -- sdfk_sample_colors shows up among the profiled kernels --; the README scene (64 operations:
    its second pass would cost more than the store pattern gains, profiles/r06_ab_color_passes.txt) and a small grid do not."""
    L = N.lib()
    mn, mx = [-2.8125] * 3, [2.8125] * 3

    def kernels(sdf, dims):
        N.check(L.sdfk_profile_reset())
        N.check(L.sdfk_profile_enable(1))
        v = sdf.ToVoxels(mn, mx, *dims, clipToBounds=True)
        N.check(L.sdfk_synchronize())
        N.check(L.sdfk_profile_enable(0))
        names = {k for k, c in N.profile_snapshot().items() if c[1]}
        N.check(L.sdfk_profile_reset())
        return names, v

    assert N.get_option(N.OPT_COLOR_PASSES) == 0 or "SDFK_COLOR_PASSES" in os.environ
    with N.option(N.OPT_COLOR_PASSES, 0):
        scene, sdf = S.CATALOGUE["cylinder"]()
        assert sdf.writes_color and sdf.ir()[1] <= 24
        names, v = kernels(sdf, (128, 128, 256))
        assert "sdfk_sample_colors" in names and "sdfk_sample_bits_nc_clip" in names and "sdfk_sample_bits_clip" not in names, names
        ov, oc = O.sample(scene, mn, mx, 128, 128, 256)
        O.clip_to_bounds(ov, mn, mx)
        assert np.array_equal(v.Values, ov) and np.array_equal(v.Colors, oc)
        names, _ = kernels(sdf, (64, 64, 64))
        assert "sdfk_sample_colors" not in names, names
        _, readme = S.CATALOGUE["readme_repeat_xy"]()
        names, _ = kernels(readme, (128, 128, 256))
        assert "sdfk_sample_colors" not in names and "sdfk_sample_bits_clip" in names, names
    with N.option(N.OPT_COLOR_PASSES, 1):
        names, _ = kernels(sdf, (128, 128, 256))
        assert "sdfk_sample_colors" not in names, names
    with N.option(N.OPT_COLOR_PASSES, 2):
        names, _ = kernels(readme, (32, 32, 32))
        assert "sdfk_sample_colors" in names, names


def test_sharded_slab_steps_with_two_passes(gpu):
    """The Z-slab step samples into slab volumes (z0 > 0, context planes): two ranks sharing the GPU, the README scene, two passes forced."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join("tests", "multirank_worker.py"), "readme_repeat_xy", "40", "36", "44"]
    # (SDFK_NO_VCOLOR_EVAL: the meshes' vertex colours are gathered from the slab volumes' colour arrays -- what the second pass wrote)
    for extra in ({}, {"SDFK_NO_VCOLOR_EVAL": "1"}):
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, SDFK_COLOR_PASSES="2", **extra))
        assert out.returncode == 0 and out.stdout.count("identical") == 2, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.parametrize("seed", range(8))
def test_two_passes_on_random_compositions(gpu, seed):
    """Random compositions of the per-point catalogue (tests/scenes.py random_scene: primitives, Translate, RepeatX / Y / XY with and without the
    README colour lambda, WithColor, Union): the volume of two passes == the volume of one pass == the oracle's, on a row-tiled and a
    plane-chunk shape."""
    scene, sdf = S.random_scene(seed)
    if not sdf.writes_color:
        pytest.skip("this composition assigns no colour")
    mn, mx = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]
    for dims, clip in (((12, 10, 256), True), ((11, 13, 37), False)):
        ov, oc = O.sample(scene, mn, mx, *dims)
        if clip:
            O.clip_to_bounds(ov, mn, mx)
        vols = []
        for passes in (1, 2):
            with N.option(N.OPT_COLOR_PASSES, passes):
                v = sdf.ToVoxels(mn, mx, *dims, clipToBounds=clip)
                vols.append((v.Values.copy(), v.Colors.copy()))
                assert np.array_equal(v.Values, ov) and np.array_equal(v.Colors, oc), (seed, dims, passes)
        assert np.array_equal(vols[0][0], vols[1][0]) and np.array_equal(vols[0][1], vols[1][1])
