"""The oracle's window forms (orc_sample_window / orc_clip_window / orc_march_window: test infrastructure for grids whose
whole-volume serial sweep does not fit a test, BASELINE config C4 at 1024^3) pinned against the whole-volume oracle --
which is itself pinned on the reference's known answers (tests/test_oracle_golden.py): on grids small enough to do both,
a window's voxels are the whole volume's voxels, and the part of a window's mesh that belongs to cell layers [lb, le) is,
bit for bit, that part of the whole mesh (positions use global z, the transform is the whole grid's)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import scenes as S

MN, MX = [-2.8125, -2.5, -2.25], [2.8125, 2.75, 2.5]


@pytest.mark.parametrize("name,dims", [("readme_repeat_xy", (22, 19, 37)), ("union8", (18, 21, 33)), ("sphere_w", (16, 16, 24))])
def test_window_equals_whole(name, dims):
    scene, _ = S.CATALOGUE[name]()
    nx, ny, nz = dims
    v, c = O.sample(scene, MN, MX, *dims)
    vc = v.copy()
    O.clip_to_bounds(vc, MN, MX)
    whole = O.march(vc, c, MN, MX)
    assert len(whole.vertices) > 200
    for lb, le in ((0, 5), (3, 9), (7, 8), (10, 24), (nz - 9, nz - 1), (0, nz - 1)):
        z0, z1 = max(lb - 2, 0), min(le + 2, nz)
        wv, wc = O.sample_window(scene, MN, MX, nx, ny, nz, z0, z1 - z0)
        assert np.array_equal(wv, v[:, :, z0:z1]) and np.array_equal(wc, c[:, :, z0:z1])
        wv, wc = O.sample_window(scene, MN, MX, nx, ny, nz, z0, z1 - z0, clip=True)
        assert np.array_equal(wv, vc[:, :, z0:z1])
        wm = O.march_window(wv, wc, z0, nz, MN, MX)
        got = O.window_part(wm, nx, ny, z0, lb, le)
        want = O.window_part(whole, nx, ny, 0, lb, le)
        for a, b in zip(got, want):
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
    # the whole grid as one window is the whole-volume sweep
    wm = O.march_window(vc, c, 0, nz, MN, MX)
    for f in ("vertices", "colors", "normals", "triangles", "cells"):
        assert np.array_equal(getattr(wm, f), getattr(whole, f), equal_nan=True)


def test_window_with_impossible_case_13_cells_at_the_seam():
    """Cells that emit nothing ("Impossible case 13?") neither create nor reference vertices: creation passes on.  Random
    volumes with planted case-13 cells around the window's first layers."""
    rng = np.random.default_rng(21)
    nx, ny, nz = 12, 11, 20
    v = rng.standard_normal((nx, ny, nz)).astype(np.float32)
    c = rng.random((nx, ny, nz, 3), dtype=np.float32)
    whole = O.march(v, c, MN, MX)
    for lb, le in ((4, 9), (5, 6), (9, 19)):
        z0, z1 = max(lb - 2, 0), min(le + 2, nz)
        wm = O.march_window(v[:, :, z0:z1], c[:, :, z0:z1], z0, nz, MN, MX)
        for a, b in zip(O.window_part(wm, nx, ny, z0, lb, le), O.window_part(whole, nx, ny, 0, lb, le)):
            assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True)
