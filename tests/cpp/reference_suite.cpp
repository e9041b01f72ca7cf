// reference_suite.cpp -- the reference's NUnit tests for the hot path, restated against the
// C++ host layer (include/SdfKit.hpp) so that they read like the originals:
//   Tests/MarchingCubesTests.cs (8 tests), Tests/SdfTests.cs (3), Tests/VolumeTests.cs (those
//   that do not need an opaque CPU delegate).  Expected values are the literals asserted by the
// reference.  Runs on the GPU through libsdfkit_hip.so (tests/test_gpu_cpp_host.py builds it).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>

#include "SdfKit.hpp"

using namespace SdfKit;

static int g_fail = 0, g_run = 0;
#define ARE_EQUAL(expected, actual)                                                                     \
    do { if (!((expected) == (actual))) { printf("  FAIL %s:%d: expected %s == %s (%g vs %g)\n", __FILE__, __LINE__, #expected, #actual, (double)(expected), (double)(actual)); g_fail++; } } while (0)
#define ARE_EQUAL_TOL(expected, actual, tol)                                                            \
    do { if (!(std::fabs((double)(expected) - (double)(actual)) <= (tol))) { printf("  FAIL %s:%d: |%s - %s| = %g > %g\n", __FILE__, __LINE__, #expected, #actual, std::fabs((double)(expected) - (double)(actual)), (double)(tol)); g_fail++; } } while (0)
#define IS_TRUE(c) do { if (!(c)) { printf("  FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)
#define TEST(name) static void name(); static void run_##name() { g_run++; printf("%s\n", #name); name(); } static void name()

TEST(ColoredSpheres)   // MarchingCubesTests.cs:11-28
{
    float r = 1.0f;
    auto sdf = SdfFuncs::Union(SdfFuncs::Sphere(r * 0.4f).WithColor(1.0f, 0.2f, 0.3f).Translate(-1, 0, 0),
                               SdfFuncs::Sphere(r * 0.2f).WithColor(0.1f, 1.0f, 0.3f).Translate(1, 0, 0));
    auto volume = Voxels::SampleSdf(sdf.ToSdf(), -3.0f * Vector3::One(), 3.0f * Vector3::One(), 32, 32, 32);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(104, (int)mesh.Vertices.size());
    ARE_EQUAL(104, (int)mesh.Colors.size());
    IS_TRUE(mesh.Colors[0].X > 0.5f);
}

TEST(Sphere5)   // MarchingCubesTests.cs:31-45
{
    float r = 1.0f;
    auto volume = Voxels::SampleSdf(Sdfs::Sphere(r), -1.5f * Vector3::One(), 1.5f * Vector3::One(), 5, 5, 5);
    ARE_EQUAL(5, volume.NX);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(54, (int)mesh.Vertices.size());
    ARE_EQUAL_TOL(mesh.Center().Length(), 0.0f, 1e-6f);
    ARE_EQUAL_TOL(r, mesh.Size().X / 2.0f, 0.3f);
}

TEST(Sphere10)   // MarchingCubesTests.cs:48-62
{
    float r = 2.0f;
    auto volume = Voxels::SampleSdf(Sdfs::Sphere(r), -2.5f * Vector3::One(), 2.5f * Vector3::One(), 10, 10, 10);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(312, (int)mesh.Vertices.size());
    ARE_EQUAL_TOL(mesh.Center().Length(), 0.0f, 1e-6f);
    ARE_EQUAL_TOL(r, mesh.Size().X / 2.0f, 0.2f);
}

TEST(UnclippedSphere10)   // MarchingCubesTests.cs:65-79
{
    int n = 10;
    auto volume = Voxels::SampleSdf(Sdfs::Sphere(2.0f), -1.0f * Vector3::One(), Vector3::One(), n, n, n);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(0, (int)mesh.Vertices.size());
    ARE_EQUAL(0, (int)mesh.Triangles.size());
}

TEST(ClippedSphere10)   // MarchingCubesTests.cs:82-98
{
    int n = 10;
    auto volume = Voxels::SampleSdf(Sdfs::Sphere(2.0f), -1.0f * Vector3::One(), Vector3::One(), n, n, n);
    volume.ClipToBounds();
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(384, (int)mesh.Vertices.size());
    ARE_EQUAL_TOL(mesh.Center().Length(), 0.0f, 1e-6f);
    ARE_EQUAL_TOL(2.0f, mesh.Size().X, 1e-1f);
}

TEST(Box10)   // MarchingCubesTests.cs:101-115
{
    float r = 2.0f;
    auto volume = Voxels::SampleSdf(Sdfs::Box(r), -2.5f * Vector3::One(), 2.5f * Vector3::One(), 10, 10, 10);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(384, (int)mesh.Vertices.size());
    ARE_EQUAL_TOL(mesh.Center().Length(), 0.0f, 1e-6f);
    ARE_EQUAL_TOL(r, mesh.Size().X / 2.0f, 3e-1f);
}

TEST(Cylinder50)   // MarchingCubesTests.cs:118-138
{
    int n = 50;
    auto volume = Voxels::SampleSdf(Sdfs::Cylinder(1, 3), Vector3(-1.5f, -3.5f, -1.5f), Vector3(1.5f, 3.5f, 1.5f), n, n, n);
    ARE_EQUAL(n, volume.NX);
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1);
    ARE_EQUAL(7456, (int)mesh.Vertices.size());
    ARE_EQUAL_TOL(0.0f, mesh.Center().X, 1e-6f);
    ARE_EQUAL_TOL(0.0f, mesh.Center().Y, 1e-6f);
    ARE_EQUAL_TOL(0.0f, mesh.Center().Z, 1e-6f);
    ARE_EQUAL_TOL(1, mesh.Size().X / 2.0f, 1e-1f);
}

TEST(Sphere128Progress)   // MarchingCubesTests.cs:141-171
{
    float r = 3.0f;
    auto volume = Voxels::SampleSdf(Sdfs::Sphere(r), -3.1f * Vector3::One(), 3.1f * Vector3::One(), 128, 128, 128);
    bool gotZero = false, gotOne = false, inRange = true;
    auto progress = [&](float f) {
        if (!(f >= 0.0f && f <= 1.0f)) inRange = false;
        if (f < 1e-6f) gotZero = true;
        else if (1.0f - f < 1e-6f) gotOne = true;
    };
    auto mesh = MarchingCubes::CreateMesh(volume, 0.0f, 1, progress);
    ARE_EQUAL(72240, (int)mesh.Vertices.size());
    IS_TRUE(inRange); IS_TRUE(gotZero); IS_TRUE(gotOne);
    ARE_EQUAL_TOL(mesh.Center().Length(), 0.0f, 1e-6f);
    ARE_EQUAL_TOL(r, mesh.Size().X / 2.0f, 0.1f);
}

TEST(CreateVolumeSphere)   // SdfTests.cs:12-26 (the delegate there is (1,1,1, p.Length() - r))
{
    float r = 0.5f;
    auto sdf = Sdfs::Solid([r](Vec3 p) { return p.Length() - Val(r); });
    auto v = sdf.ToVoxels(Vector3(-1, -1, -1), Vector3(1, 1, 1), 128, 128, 128);
    ARE_EQUAL_TOL(-0.5f, v(63, 63, 63), 2.0e-2f);
}

TEST(CreateMeshSphere)   // SdfTests.cs:29-39
{
    int n = 32;
    auto mesh = Sdfs::Sphere(0.5f).ToMesh(Vector3(-1, -1, -1), Vector3(1, 1, 1), n, n, n);
    ARE_EQUAL(1248, (int)mesh.Vertices.size());
}

TEST(SolidSphere)   // SdfTests.cs:42-52
{
    float r = 0.5f;
    int n = 32;
    auto sdf = SdfExprs::Solid([r](Vec3 p) { return p.Length() - Val(r); }).ToSdf();
    auto mesh = sdf.ToMesh(Vector3(-1, -1, -1), Vector3(1, 1, 1), n, n, n);
    ARE_EQUAL(1248, (int)mesh.Vertices.size());
}

TEST(EmptyVolumeDims)   // VolumeTests.cs:11-38
{
    Voxels v(-1.0f * Vector3::One(), Vector3::One(), 5, 7, 11);
    ARE_EQUAL(5, v.NX); ARE_EQUAL(7, v.NY); ARE_EQUAL(11, v.NZ);
    ARE_EQUAL_TOL(2.0f, v.Size().X, 1e-6f);
}

TEST(SphereCenterValue)   // VolumeTests.cs:83-106
{
    auto v = Voxels::SampleSdf(Sdfs::Sphere(0.5f), -1.0f * Vector3::One(), Vector3::One(), 5, 5, 5);
    ARE_EQUAL_TOL(-0.5f, v(2, 2, 2), 1e-3f);
}

TEST(ReadmeSceneRepeatXY)   // README.md:24-30 scene through the expression API, 64^3
{
    auto sdf = SdfExprs::Sphere(0.5f)
                   .RepeatXY(1.125f, 1.125f, [](Vec3 i, Vec3 p, Vec4 d) { return Val(0.9f) * Vec3(Vector3::One()) - Vec3::Abs(i) / Val(6.0f); })
                   .ToSdf();
    auto mesh = sdf.ToMesh(-2.8125f * Vector3::One(), 2.8125f * Vector3::One(), 64, 64, 64);
    IS_TRUE(mesh.Vertices.size() > 1000);
    IS_TRUE(mesh.Triangles.size() % 3 == 0);
    for (auto& c : mesh.Colors) { if (!(c.X >= 0.9f - 0.5f - 1e-5f && c.X <= 0.9f + 1e-5f)) { IS_TRUE(false); break; } }
}

TEST(RayMarcherSphereDepth)   // RayMarcherTests.cs:10-24
{
    const int w = 50, h = 30;
    RayMarcher rt(w, h, Sdfs::Sphere(1.0f));
    auto img = rt.RenderDepth();
    ARE_EQUAL(w, img.Width);
    ARE_EQUAL(h, img.Height);
    ARE_EQUAL_TOL(4.0f, img(w / 2, h / 2), 1.0e-2f);
    IS_TRUE(img(0, 0) > 9.0f);
}

TEST(RayMarcherBoxDepth)   // RayMarcherTests.cs:27-41
{
    const int w = 50, h = 30;
    auto img = RayMarcher(w, h, Sdfs::Box(1.0f)).RenderDepth();
    ARE_EQUAL_TOL(4.0f, img(w / 2, h / 2), 1.0e-2f);
    IS_TRUE(img(0, 0) > 9.0f);
}

TEST(RayMarcherCylinderDepth)   // RayMarcherTests.cs:44-62
{
    const int w = 50, h = 30;
    const float r = 0.25f;
    auto img = RayMarcher(w, h, SdfExprs::Cylinder(r, r * 2).RepeatX(4 * r).ToSdf()).RenderDepth();
    ARE_EQUAL_TOL(5 - r, img(w / 2, h / 2 - 2), 1.0e-1f);
    IS_TRUE(img(0, 0) > 9.0f);
}

TEST(RayMarcherPlaneDepth)   // RayMarcherTests.cs:65-78
{
    const int w = 50, h = 30;
    auto img = RayMarcher(w, h, Sdfs::PlaneXY()).RenderDepth();
    ARE_EQUAL_TOL(5.0f, img(w / 2, h / 2), 1.0e-2f);
    IS_TRUE(img(0, 0) < 9.0f);
}

TEST(RayMarcherSphereRepeat)   // RayMarcherTests.cs:96-108 (192 x 108, camera (-2,2,4) -> origin)
{
    const float r = 0.5f;
    auto sdf = SdfExprs::Sphere(r).RepeatXY(2.25f * r, 2.25f * r, [](Vec3 i, Vec3, Vec4) { return Val(0.9f) * Vec3(Vector3::One()) - Vec3::Abs(i) / Val(6.0f); }).ToSdf();
    RayMarcher rm(192, 108, sdf);
    rm.ViewTransform = Matrix4x4::CreateLookAt(Vector3(-2, 2, 4), Vector3(0, 0, 0), Vector3(0, 1, 0));
    auto img = rm.Render();
    ARE_EQUAL(192, img.Width);
    ARE_EQUAL(108, img.Height);
    const Vector3 c = img(96, 54);
    IS_TRUE(c.X > 0.1f && c.X <= 1.0f);
}

TEST(MeshWriteObjFormat)   // Mesh.cs:66-97 + invariant-culture Single.ToString()
{
    ARE_EQUAL(true, Mesh::FormatSingle(1.5f) == "1.5");
    ARE_EQUAL(true, Mesh::FormatSingle(-0.1f) == "-0.1");
    ARE_EQUAL(true, Mesh::FormatSingle(1e-5f) == "1E-05");
    ARE_EQUAL(true, Mesh::FormatSingle(0.0001f) == "0.0001");
    ARE_EQUAL(true, Mesh::FormatSingle(1234567.0f) == "1234567");
    // .NET Core 3.0+: scientific iff the decimal-point position > max(digits, 7) or < -3 (Number.Formatting.cs)
    ARE_EQUAL(true, Mesh::FormatSingle(12345678.0f) == "12345678");          // 8 digits, position 8: fixed
    ARE_EQUAL(true, Mesh::FormatSingle(16777216.0f) == "16777216");
    ARE_EQUAL(true, Mesh::FormatSingle(1e7f) == "1E+07");                    // 1 digit, position 8 > 7
    ARE_EQUAL(true, Mesh::FormatSingle(1.5e7f) == "1.5E+07");
    ARE_EQUAL(true, Mesh::FormatSingle(123456792.0f) == "1.2345679E+08");    // 8 digits, position 9 > 8: scientific
    ARE_EQUAL(true, Mesh::FormatSingle(100000008.0f) == "1.0000001E+08");
    ARE_EQUAL(true, Mesh::FormatSingle(105242136.0f) == "105242136");        // 9 digits, position 9: fixed
    ARE_EQUAL(true, Mesh::FormatSingle(1052421360.0f) == "1.0524214E+09");
    ARE_EQUAL(true, Mesh::FormatSingle(-12345678.0f) == "-12345678");
    ARE_EQUAL(true, Mesh::FormatSingle(0.00012345678f) == "0.00012345678");  // position -3: fixed
    ARE_EQUAL(true, Mesh::FormatSingle(0.000012345678f) == "1.2345678E-05");
    ARE_EQUAL(true, Mesh::FormatSingle(0.33333334f) == "0.33333334");
    ARE_EQUAL(true, Mesh::FormatSingle(100.0f) == "100");
    auto mesh = Sdfs::Sphere(1.0f).ToMesh(Vector3(-1.5f, -1.5f, -1.5f), Vector3(1.5f, 1.5f, 1.5f), 5, 5, 5, 2048, -1, false);
    std::ostringstream os;
    mesh.WriteObj(os);
    const std::string txt = os.str();
    size_t lines = 0;
    for (char c : txt) lines += c == '\n';
    ARE_EQUAL(mesh.Vertices.size() * 2 + mesh.Triangles.size() / 3, lines);
    IS_TRUE(txt.rfind("v ", 0) == 0);
    IS_TRUE(txt.find("\nvn ") != std::string::npos && txt.find("\nf 1//1 ") != std::string::npos);
}

TEST(SdfSamplePoints)   // SdfEx.Sample (Sdf.cs:22-47): the value at the centre of a sphere of radius 0.5 is -0.5 (VolumeTests.cs:83-106 asserts the same of the grid)
{
    auto sdf = Sdfs::Sphere(0.5f);
    std::vector<Vector3> pts = {Vector3(0, 0, 0), Vector3(0.5f, 0, 0), Vector3(0, 2, 0), Vector3(3, 4, 0)};
    std::vector<Vector4> out(pts.size(), Vector4(9, 8, 7, 6));
    sdf.Sample(pts, out);
    IS_TRUE(out[0].W == -0.5f && out[1].W == 0.0f && out[2].W == 1.5f && out[3].W == 4.5f);
    IS_TRUE(out[0].X == 9 && out[2].Y == 8 && out[3].Z == 7);      // a .W-only delegate leaves X, Y, Z alone (Sdf.cs:211)
    auto col = sdf.WithColor(0.25f, 0.5f, 0.75f);
    col.Sample(pts, out);
    IS_TRUE(out[3].W == 4.5f && out[3].X == 0.25f && out[3].Y == 0.5f && out[3].Z == 0.75f);
}

TEST(NodeOfTwoRanksSharingTheGpu)   // SdfEx.ToMesh (Sdf.cs:29-39: 1248 vertices) through sdfk_node_*: two rank threads of the library on GPU 0
{
    Node node({0, 0});
    ARE_EQUAL(2, node.World());
    for (int rep = 0; rep < 2; rep++) {
        auto mesh = node.ToMesh(Sdfs::Sphere(0.5f), Vector3(-1, -1, -1), Vector3(1, 1, 1), 32, 32, 32);
        ARE_EQUAL((size_t)1248, mesh.Vertices.size());
        auto one = Sdfs::Sphere(0.5f).ToMesh(Vector3(-1, -1, -1), Vector3(1, 1, 1), 32, 32, 32);
        ARE_EQUAL(one.Triangles.size(), mesh.Triangles.size());
        bool same = one.Triangles == mesh.Triangles;
        for (size_t i = 0; same && i < one.Vertices.size(); i++)
            same = one.Vertices[i].X == mesh.Vertices[i].X && one.Vertices[i].Y == mesh.Vertices[i].Y && one.Vertices[i].Z == mesh.Vertices[i].Z &&
                   one.Normals[i].X == mesh.Normals[i].X && one.Normals[i].Y == mesh.Normals[i].Y && one.Normals[i].Z == mesh.Normals[i].Z;
        IS_TRUE(same);
        IS_TRUE(one.Min.X == mesh.Min.X && one.Max.Z == mesh.Max.Z);
    }
}

int main()
{
    run_MeshWriteObjFormat(); run_SdfSamplePoints(); run_NodeOfTwoRanksSharingTheGpu();
    run_RayMarcherSphereDepth(); run_RayMarcherBoxDepth(); run_RayMarcherCylinderDepth(); run_RayMarcherPlaneDepth(); run_RayMarcherSphereRepeat();
    run_ColoredSpheres(); run_Sphere5(); run_Sphere10(); run_UnclippedSphere10(); run_ClippedSphere10(); run_Box10();
    run_Cylinder50(); run_Sphere128Progress(); run_CreateVolumeSphere(); run_CreateMeshSphere(); run_SolidSphere();
    run_EmptyVolumeDims(); run_SphereCenterValue(); run_ReadmeSceneRepeatXY();
    printf("%d tests, %d failures\n", g_run, g_fail);
    sdfk_shutdown();
    return g_fail ? 1 : 0;
}
