// obj_host.cpp -- Mesh.WriteObj (Mesh.cs:66-97) of a GPU-produced mesh through the C++ host layer (include/SdfKit.hpp over the
// C ABI): `obj_host <scene> <path>` samples + meshes one of the reference's NUnit scenes on the GPU exactly as the test does
// (MarchingCubesTests.cs:11-28, 31-45, 118-138) and writes the OBJ file.  tests/test_gpu_write_obj.py compares the bytes with
// the oracle's mesh written by the same writers.
#include <cstdio>
#include <cstring>
#include <string>

#include "SdfKit.hpp"

using namespace SdfKit;

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: obj_host <ColoredSpheres|Sphere5|Cylinder50> <out.obj>\n"); return 2; }
    const std::string scene = argv[1];
    Mesh mesh;
    if (scene == "ColoredSpheres") {
        auto sdf = SdfFuncs::Union(SdfFuncs::Sphere(0.4f).WithColor(1.0f, 0.2f, 0.3f).Translate(-1, 0, 0),
                                   SdfFuncs::Sphere(0.2f).WithColor(0.1f, 1.0f, 0.3f).Translate(1, 0, 0));
        auto volume = Voxels::SampleSdf(sdf.ToSdf(), -3.0f * Vector3::One(), 3.0f * Vector3::One(), 32, 32, 32);
        mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    } else if (scene == "Sphere5") {
        auto volume = Voxels::SampleSdf(Sdfs::Sphere(1.0f), -1.5f * Vector3::One(), 1.5f * Vector3::One(), 5, 5, 5);
        mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    } else if (scene == "Cylinder50") {
        auto volume = Voxels::SampleSdf(Sdfs::Cylinder(1, 3), Vector3(-1.5f, -3.5f, -1.5f), Vector3(1.5f, 3.5f, 1.5f), 50, 50, 50);
        mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    } else {
        fprintf(stderr, "unknown scene %s\n", argv[1]);
        return 2;
    }
    mesh.WriteObj(std::string(argv[2]));
    printf("%zu vertices, %zu indices\n", mesh.Vertices.size(), mesh.Triangles.size());
    sdfk_shutdown();
    return 0;
}
