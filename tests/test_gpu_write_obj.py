"""Mesh.WriteObj (Mesh.cs:66-97) of GPU-produced meshes: the reference's Sphere5, ColoredSpheres and Cylinder50 scenes
(MarchingCubesTests.cs:31-45, 11-28, 118-138) sampled + meshed by the HIP path, written by BOTH host writers -- the Python mirror
(sdfkit_amd.api.Mesh.WriteObj) and the C++ host layer over the C ABI (include/SdfKit.hpp, tests/cpp/obj_host.cpp) -- must be
byte-identical to the oracle's mesh of the same scene written by the same writer, and have the reference's line structure:
every `v x y z`, then every `vn x y z`, then `f a//a b//b c//c` with 1-based indices.

No reference-held .obj exists; the NUMBER FORMAT is pinned to the .NET Core 3.0+ rule only (tests/test_write_obj.py,
tests/golden/README.md)."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from sdfkit_amd import MarchingCubes, Voxels
from sdfkit_amd.api import Mesh
from tests import scenes as S

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name -> (scene builder, min, max, grid, vertices the reference's test asserts)
SCENES = {
    "Sphere5": (lambda: S.sphere_w(1.0), [-1.5] * 3, [1.5] * 3, 5, 54),
    "ColoredSpheres": (S.colored_spheres, [-3.0] * 3, [3.0] * 3, 32, 104),
    "Cylinder50": (lambda: S.cylinder(1.0, 3.0), [-1.5, -3.5, -1.5], [1.5, 3.5, 1.5], 50, 7456),
}


@pytest.fixture(scope="module")
def obj_host(tmp_path_factory):
    from sdfkit_amd import _native as N
    N.lib()
    exe = str(tmp_path_factory.mktemp("obj") / "obj_host")
    libdir = os.path.join(ROOT, "sdfkit_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "obj_host.cpp"), "-o", exe,
                           "-L", libdir, "-lsdfkit_hip", f"-Wl,-rpath,{libdir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


V_LINE = re.compile(r"^v \S+ \S+ \S+$")
VN_LINE = re.compile(r"^vn \S+ \S+ \S+$")
F_LINE = re.compile(r"^f (\d+)//\1 (\d+)//\2 (\d+)//\3$")


@pytest.mark.parametrize("name", sorted(SCENES))
def test_gpu_mesh_obj_is_the_oracle_mesh_obj(gpu, obj_host, tmp_path, name):
    build, mn, mx, n, expect_v = SCENES[name]
    scene, sdf = build()
    # the oracle's mesh of the scene, written by the Python writer: the expected text
    ov, oc = O.sample(scene, mn, mx, n, n, n)
    ref = O.march(ov, oc, mn, mx)
    assert len(ref.vertices) == expect_v
    want_path = str(tmp_path / "oracle.obj")
    Mesh(np.ascontiguousarray(ref.vertices, np.float32), np.ascontiguousarray(ref.colors, np.float32),
         np.ascontiguousarray(ref.normals, np.float32), np.ascontiguousarray(ref.triangles, np.int32)).WriteObj(want_path)
    want = open(want_path, "rb").read()

    # GPU: Voxels.SampleSdf -> MarchingCubes.CreateMesh as the reference's test does, Python mirror's writer
    volume = Voxels.SampleSdf(sdf, mn, mx, n, n, n)
    mesh = MarchingCubes.CreateMesh(volume, 0.0, 1)
    assert len(mesh.Vertices) == expect_v
    py_path = str(tmp_path / "gpu_py.obj")
    mesh.WriteObj(py_path)
    got_py = open(py_path, "rb").read()
    assert got_py == want, f"{name}: the Python writer's text of the GPU mesh differs from the oracle mesh's"

    # GPU through the C++ host layer (SdfKit.hpp over the C ABI) and ITS writer
    cc_path = str(tmp_path / "gpu_cc.obj")
    p = subprocess.run([obj_host, name, cc_path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert f"{expect_v} vertices" in p.stdout
    got_cc = open(cc_path, "rb").read()
    assert got_cc == want, f"{name}: the C++ writer's text of the GPU mesh differs from the oracle mesh's"

    # line structure (Mesh.cs:72-97): V `v` lines, V `vn` lines, T `f` lines, 1-based, nothing else
    lines = want.decode("ascii").split("\n")
    assert lines[-1] == ""
    lines = lines[:-1]
    nv, nt = len(ref.vertices), len(ref.triangles) // 3
    assert len(lines) == 2 * nv + nt
    assert all(V_LINE.match(ln) for ln in lines[:nv])
    assert all(VN_LINE.match(ln) for ln in lines[nv:2 * nv])
    faces = [F_LINE.match(ln) for ln in lines[2 * nv:]]
    assert all(faces)
    idx = np.array([[int(g) for g in m.groups()] for m in faces], np.int64).ravel()
    assert idx.min() >= 1 and idx.max() == nv          # 1-based: index 0 of Triangles is written as 1
    assert np.array_equal(idx - 1, ref.triangles)
    # the `v` lines carry the float32 positions exactly (shortest round-trip digits)
    back = np.array([[np.float32(float(t)) for t in ln.split()[1:]] for ln in lines[:nv]], np.float32)
    assert np.array_equal(back, ref.vertices)
