// format_host.cpp -- test shim (g++ -shared): Mesh::FormatSingle / Mesh::WriteObj of the C++ host layer (include/SdfKit.hpp,
// Mesh.cs:66-97) without a device: tests/test_write_obj.py holds the Python mirror's writer to the same text, line by line.
#include <sstream>
#include <string>
#include "SdfKit.hpp"

extern "C" {

// invariant-culture Single.ToString() of `x` into out (NUL-terminated); returns the length
int fmt_single(float x, char* out, int cap)
{
    const std::string s = SdfKit::Mesh::FormatSingle(x);
    if ((int)s.size() + 1 > cap) return -1;
    std::copy(s.begin(), s.end(), out);
    out[s.size()] = 0;
    return (int)s.size();
}

// Mesh.WriteObj of the given arrays to `path`
int write_obj(const float* vertices3, const float* normals3, long nv, const int* triangles, long ni, const char* path)
{
    SdfKit::Mesh m;
    m.Vertices.resize((size_t)nv);
    m.Normals.resize((size_t)nv);
    for (long i = 0; i < nv; i++) {
        m.Vertices[(size_t)i] = SdfKit::Vector3(vertices3[3 * i], vertices3[3 * i + 1], vertices3[3 * i + 2]);
        m.Normals[(size_t)i] = SdfKit::Vector3(normals3[3 * i], normals3[3 * i + 1], normals3[3 * i + 2]);
    }
    m.Triangles.assign(triangles, triangles + ni);
    m.WriteObj(std::string(path));
    return 0;
}

}  // extern "C"
