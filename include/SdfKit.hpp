// SdfKit.hpp -- header-only C++ host layer over the C ABI (sdfkit_hip.h) that mirrors the
// reference's public API for the hot path: same type and member names, argument meaning,
// defaults and error behaviour as praeclarum/SdfKit (C#), so that code written against
//   Sdfs / SdfFuncs / SdfExprs / Voxels / MarchingCubes / Mesh
// ports line by line (tests/cpp/reference_suite.cpp restates the reference's NUnit tests).
// The C# shim of INTEGRATION.md is the same layer in the reference's own language; .NET is
// not available in the build image, C++ is.
//
// All compute runs in libsdfkit_hip.so on the GPU.  SDFs are symbolic: arithmetic on
// SdfKit::Val records float32 SSA instructions (the role LINQ expression trees play in the
// reference, GlobalUsings.cs:16), lowered by sdfk_program_create (JIT, like SdfExpr.cs:225-273).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sdfkit_hip.h"

namespace SdfKit {

struct Vector3 {
    float X = 0, Y = 0, Z = 0;
    Vector3() = default;
    Vector3(float x, float y, float z) : X(x), Y(y), Z(z) {}
    explicit Vector3(float v) : X(v), Y(v), Z(v) {}
    static Vector3 One() { return Vector3(1, 1, 1); }
    static Vector3 Zero() { return Vector3(0, 0, 0); }
    float Length() const { return std::sqrt((X * X + Y * Y) + Z * Z); }
};
struct Vector4 {   // (colour, distance) of a sample: what an Sdf delegate writes per point (Sdf.cs:8)
    float X = 0, Y = 0, Z = 0, W = 0;
    Vector4() = default;
    Vector4(float x, float y, float z, float w) : X(x), Y(y), Z(z), W(w) {}
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
inline Vector3 operator-(Vector3 a, Vector3 b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
inline Vector3 operator*(Vector3 a, float s) { return {a.X * s, a.Y * s, a.Z * s}; }
inline Vector3 operator*(float s, Vector3 a) { return a * s; }

inline void Check(int status)
{
    if (status != SDFK_OK) throw std::runtime_error(std::string("sdfkit_hip: ") + sdfk_last_error());
}
inline void EnsureInit()
{
    static bool done = false;
    if (!done) {
        // The library's streams want 8 hardware queues (sdfk_init in sdfkit_hip.hip says why); the HIP runtime reads the
        // variable when it initialises, and the library never edits the environment itself: a host binding's job.
        setenv("GPU_MAX_HW_QUEUES", "8", 0);
        Check(sdfk_init(0));
        done = true;
    }
}
// sdfk_set_option / sdfk_get_option (enum sdfk_option): run-time switches that used to be environment variables
inline void SetOption(sdfk_option key, int64_t value) { Check(sdfk_set_option((int32_t)key, value)); }
inline int64_t GetOption(sdfk_option key)
{
    int64_t v = 0;
    Check(sdfk_get_option((int32_t)key, &v));
    return v;
}

// ---------------------------------------------------------------------------------------
// symbolic float32 values (one SSA instruction each)
// ---------------------------------------------------------------------------------------
struct Builder {
    std::vector<sdfk_op> ops;
    int emit(int opcode, int a = -1, int b = -1, int c = -1, int d = -1, float imm = 0.f)
    {
        ops.push_back(sdfk_op{opcode, a, b, c, d, imm});
        return (int)ops.size() - 1;
    }
};

struct Val {
    Builder* b = nullptr;
    int id = -1;
    float k = 0;            // constant payload while not yet bound to a builder
    Val() = default;
    Val(float c) : k(c) {}  // NOLINT: literals convert implicitly, like C# float literals
    Val(Builder* bb, int i) : b(bb), id(i) {}
    int bind(Builder* bb) const { return b ? id : bb->emit(SDFK_OP_CONST, -1, -1, -1, -1, k); }
};
inline Builder* builder_of(const Val& a, const Val& c) { return a.b ? a.b : c.b; }
inline Val bin(int op, const Val& a, const Val& c)
{
    Builder* b = builder_of(a, c);
    if (!b) {   // two literals: the same IEEE binary32 operation, done here
        switch (op) {
        case SDFK_OP_ADD: return Val(a.k + c.k);
        case SDFK_OP_SUB: return Val(a.k - c.k);
        case SDFK_OP_MUL: return Val(a.k * c.k);
        case SDFK_OP_DIV: return Val(a.k / c.k);
        default: throw std::logic_error("SdfKit::Val: this operation needs a symbolic operand");
        }
    }
    const int ia = a.bind(b), ic = c.bind(b);
    return Val(b, b->emit(op, ia, ic));
}
inline Val un(int op, const Val& a)
{
    if (!a.b) {
        switch (op) {
        case SDFK_OP_NEG: return Val(-a.k);
        case SDFK_OP_ABS: return Val(std::fabs(a.k));
        case SDFK_OP_SQRT: return Val(std::sqrt(a.k));
        case SDFK_OP_FLOOR: return Val(std::floor(a.k));
        default: throw std::logic_error("SdfKit::Val: this operation needs a symbolic operand");
        }
    }
    return Val(a.b, a.b->emit(op, a.id));
}
inline Val operator+(const Val& a, const Val& c) { return bin(SDFK_OP_ADD, a, c); }
inline Val operator-(const Val& a, const Val& c) { return bin(SDFK_OP_SUB, a, c); }
inline Val operator*(const Val& a, const Val& c) { return bin(SDFK_OP_MUL, a, c); }
inline Val operator/(const Val& a, const Val& c) { return bin(SDFK_OP_DIV, a, c); }
inline Val operator-(const Val& a) { return un(SDFK_OP_NEG, a); }

struct MathF {   // System.MathF members used by the catalogue
    static Val Sqrt(const Val& a) { return un(SDFK_OP_SQRT, a); }
    static Val Abs(const Val& a) { return un(SDFK_OP_ABS, a); }
    static Val Floor(const Val& a) { return un(SDFK_OP_FLOOR, a); }
    static Val Max(const Val& a, const Val& c) { return bin(SDFK_OP_MAX_IEEE, a, c); }
    static Val Min(const Val& a, const Val& c) { return bin(SDFK_OP_MIN_IEEE, a, c); }
};

struct Vec3 {   // System.Numerics.Vector3 over symbolic components
    Val X, Y, Z;
    Vec3() = default;
    Vec3(Val x, Val y, Val z) : X(x), Y(y), Z(z) {}
    Vec3(Vector3 v) : X(v.X), Y(v.Y), Z(v.Z) {}  // NOLINT
    Val Length() const { return MathF::Sqrt((X * X + Y * Y) + Z * Z); }
    static Vec3 Abs(const Vec3& a) { return {MathF::Abs(a.X), MathF::Abs(a.Y), MathF::Abs(a.Z)}; }
    static Vec3 Max(const Vec3& a, const Vec3& c) { return {bin(SDFK_OP_MAX_SEL, a.X, c.X), bin(SDFK_OP_MAX_SEL, a.Y, c.Y), bin(SDFK_OP_MAX_SEL, a.Z, c.Z)}; }
    static Vec3 Min(const Vec3& a, const Vec3& c) { return {bin(SDFK_OP_MIN_SEL, a.X, c.X), bin(SDFK_OP_MIN_SEL, a.Y, c.Y), bin(SDFK_OP_MIN_SEL, a.Z, c.Z)}; }
    static Val Dot(const Vec3& a, const Vec3& c) { return (a.X * c.X + a.Y * c.Y) + a.Z * c.Z; }
};
inline Vec3 operator+(const Vec3& a, const Vec3& c) { return {a.X + c.X, a.Y + c.Y, a.Z + c.Z}; }
inline Vec3 operator-(const Vec3& a, const Vec3& c) { return {a.X - c.X, a.Y - c.Y, a.Z - c.Z}; }
inline Vec3 operator*(const Val& s, const Vec3& a) { return {s * a.X, s * a.Y, s * a.Z}; }
inline Vec3 operator/(const Vec3& a, const Val& s) { return {a.X / s, a.Y / s, a.Z / s}; }

struct Vec4 {   // SdfOutput = Vector4(colour, distance)
    Val X, Y, Z, W;
    Vec4() = default;
    Vec4(const Vec3& c, Val w) : X(c.X), Y(c.Y), Z(c.Z), W(w) {}
    Vec4(Val x, Val y, Val z, Val w) : X(x), Y(y), Z(z), W(w) {}
};

inline Val Mod(const Val& a, const Val& c) { return a - c * MathF::Floor(a / c); }                       // VectorData.cs:697-698
inline Val VMax(const Vec3& v) { return MathF::Max(MathF::Max(v.X, v.Y), v.Z); }                         // VectorData.cs:860-861
inline Val SelectLt(const Val& a, const Val& c, const Val& t, const Val& f)
{
    Builder* b = a.b ? a.b : (c.b ? c.b : (t.b ? t.b : f.b));
    return Val(b, b->emit(SDFK_OP_SEL_LT, a.bind(b), c.bind(b), t.bind(b), f.bind(b)));
}

// ---------------------------------------------------------------------------------------
// Sdf / SdfFunc
// ---------------------------------------------------------------------------------------
using PointFn = std::function<Vec4(Vec3)>;
using SdfIndexedOutputModifierFunc = std::function<Vec3(Vec3 /*index*/, Vec3 /*position*/, Vec4 /*output*/)>;

class Mesh;
class Voxels;

// The reference's `Sdf` delegate (Sdf.cs:8), restricted to SDFs that have a GPU program.
class Sdf {
public:
    Sdf(PointFn fn, bool writesColor) : st_(std::make_shared<State>()) { st_->fn = std::move(fn); st_->writesColor = writesColor; }
    bool WritesColor() const { return st_->writesColor; }
    // the SDF as the flat op list the library compiles (what sdfk_program_create and sdfk_node_* take)
    void Lower(std::vector<sdfk_op>& ops, int32_t out[4]) const
    {
        Builder b;
        Vec3 p(Val(&b, b.emit(SDFK_OP_X)), Val(&b, b.emit(SDFK_OP_Y)), Val(&b, b.emit(SDFK_OP_Z)));
        Vec4 o = st_->fn(p);
        out[0] = out[1] = out[2] = -1;
        out[3] = o.W.bind(&b);
        if (st_->writesColor) { out[0] = o.X.bind(&b); out[1] = o.Y.bind(&b); out[2] = o.Z.bind(&b); }
        ops = b.ops;
    }
    sdfk_program* Program() const
    {
        if (!st_->prog) {
            EnsureInit();
            std::vector<sdfk_op> ops;
            int32_t out[4];
            Lower(ops, out);
            Check(sdfk_program_create(ops.data(), (int32_t)ops.size(), out, st_->writesColor ? 1 : 0, &st_->prog));
        }
        return st_->prog;
    }
    // SdfEx.WithColor (Sdf.cs:101-115)
    Sdf WithColor(Vector3 color) const { PointFn f = st_->fn; return Sdf([f, color](Vec3 p) { return Vec4(Vec3(color), f(p).W); }, true); }
    Sdf WithColor(float r, float g, float b) const { return WithColor(Vector3(r, g, b)); }
    // SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points; out[i] = (colour, distance) of points[i] -- an SDF that only assigns .W leaves
    // X, Y, Z of the caller's elements alone, like the reference's delegates.  batchSize / maxDegreeOfParallelism have no GPU meaning.
    void Sample(const std::vector<Vector3>& points, std::vector<Vector4>& out, int /*batchSize*/ = 2048, int /*maxDegreeOfParallelism*/ = -1) const
    {
        if (out.size() != points.size()) out.resize(points.size());
        static_assert(sizeof(Vector3) == 12 && sizeof(Vector4) == 16, "plain float triples / quadruples");
        Check(sdfk_eval_points(Program(), points.empty() ? nullptr : &points[0].X, (int64_t)points.size(), out.empty() ? nullptr : &out[0].X));
    }
    // SdfEx.ToVoxels / ToMesh (Sdf.cs:49-63)
    inline Voxels ToVoxels(Vector3 min, Vector3 max, int nx, int ny, int nz, int batchSize = 2048, int maxDegreeOfParallelism = -1, bool clipToBounds = true) const;
    inline Mesh ToMesh(Vector3 min, Vector3 max, int nx, int ny, int nz, int batchSize = 2048, int maxDegreeOfParallelism = -1,
                       bool clipToBounds = true, float isoValue = 0.0f, int step = 1, const std::function<void(float)>& progress = nullptr) const;

private:
    struct State {
        PointFn fn;
        bool writesColor = true;
        sdfk_program* prog = nullptr;
        ~State() { if (prog) sdfk_program_destroy(prog); }
    };
    std::shared_ptr<State> st_;
};

// per-point SDF (SdfFunc delegate / SdfExpr tree): SdfFuncEx + SdfExprEx members
class SdfFunc {
public:
    SdfFunc(PointFn f) : fn(std::move(f)) {}  // NOLINT
    PointFn fn;
    Sdf ToSdf() const { return Sdf(fn, true); }                                                    // Sdf.cs:301-313, SdfExpr.cs:208-211
    SdfFunc Translate(Vector3 off) const { PointFn f = fn; return SdfFunc([f, off](Vec3 p) { return f(p - Vec3(off)); }); }     // Sdf.cs:315-326
    SdfFunc Translate(float x, float y, float z) const { return Translate(Vector3(x, y, z)); }
    SdfFunc WithColor(Vector3 c) const { PointFn f = fn; return SdfFunc([f, c](Vec3 p) { return Vec4(Vec3(c), f(p).W); }); }    // Sdf.cs:328-340
    SdfFunc WithColor(float r, float g, float b) const { return WithColor(Vector3(r, g, b)); }
    SdfFunc Color(float r, float g, float b) const { return WithColor(r, g, b); }                  // SdfExpr.cs:143-147
    SdfFunc ModifyInput(std::function<Vec3(Vec3)> m) const { PointFn f = fn; return SdfFunc([f, m](Vec3 p) { return f(m(p)); }); }  // SdfExpr.cs:79-89
    static Val Rep(const Val& c, float s) { Val sv(s); return Mod(c + sv * Val(0.5f), sv) - sv * Val(0.5f); }
    static Val Idx(const Val& c, float s) { Val sv(s); return MathF::Floor((c + sv * Val(0.5f)) / sv); }
    SdfFunc RepeatX(float sx) const { return ModifyInput([sx](Vec3 p) { return Vec3(Rep(p.X, sx), p.Y, p.Z); }); }              // SdfExpr.cs:149-153
    SdfFunc RepeatY(float sy) const { return ModifyInput([sy](Vec3 p) { return Vec3(p.X, Rep(p.Y, sy), p.Z); }); }              // SdfExpr.cs:197-201
    SdfFunc RepeatXY(float sx, float sy) const { return ModifyInput([sx, sy](Vec3 p) { return Vec3(Rep(p.X, sx), Rep(p.Y, sy), p.Z); }); }  // SdfExpr.cs:155-161
    SdfFunc RepeatXY(float sx, float sy, SdfIndexedOutputModifierFunc mod) const                  // SdfExpr.cs:163-178
    {
        PointFn f = fn;
        return SdfFunc([f, sx, sy, mod](Vec3 p) {
            Vec3 mp(Rep(p.X, sx), Rep(p.Y, sy), p.Z);
            Vec3 index(Idx(p.X, sx), Idx(p.Y, sy), Val(0.0f));
            Vec4 d = f(mp);
            return Vec4(mod(index, mp, d), d.W);
        });
    }
    SdfFunc RepeatXZ(float sx, float sz, SdfIndexedOutputModifierFunc mod) const                  // SdfExpr.cs:180-195
    {
        PointFn f = fn;
        return SdfFunc([f, sx, sz, mod](Vec3 p) {
            Vec3 mp(Rep(p.X, sx), p.Y, Rep(p.Z, sz));
            Vec3 index(Idx(p.X, sx), Val(0.0f), Idx(p.Z, sz));
            Vec4 d = f(mp);
            return Vec4(mod(index, mp, d), d.W);
        });
    }
};

inline Val BoxDistance(Vec3 p, Vector3 bounds)
{   // Vector3.Max(wd, Zero).Length() + VMax(Vector3.Min(wd, Zero))     Sdf.cs:134-136
    Vec3 wd = Vec3::Abs(p) - Vec3(bounds);
    return Vec3::Max(wd, Vec3(Vector3::Zero())).Length() + VMax(Vec3::Min(wd, Vec3(Vector3::Zero())));
}

struct SdfFuncs {   // Sdf.cs:217-249
    static SdfFunc Box(Vector3 bounds) { return SdfFunc([bounds](Vec3 p) { return Vec4(Vec3(Vector3::One()), BoxDistance(p, bounds)); }); }
    static SdfFunc Box(float b) { return Box(Vector3(b)); }
    static SdfFunc Sphere(float r) { return SdfFunc([r](Vec3 p) { return Vec4(Vec3(Vector3::One()), p.Length() - Val(r)); }); }
    static SdfFunc Union(SdfFunc a, SdfFunc b)
    {
        return SdfFunc([a, b](Vec3 p) {
            Vec4 da = a.fn(p), db = b.fn(p);   // da.W < db.W ? da : db
            return Vec4(SelectLt(da.W, db.W, da.X, db.X), SelectLt(da.W, db.W, da.Y, db.Y), SelectLt(da.W, db.W, da.Z, db.Z), SelectLt(da.W, db.W, da.W, db.W));
        });
    }
};

struct SdfExprs {   // SdfExpr.cs:16-69
    static SdfFunc Box(Vector3 bounds) { return SdfFuncs::Box(bounds); }
    static SdfFunc Box(float b) { return SdfFuncs::Box(b); }
    static SdfFunc Cylinder(float r, float h, Vector3 color = Vector3::One())
    {
        return SdfFunc([r, h, color](Vec3 p) { return Vec4(Vec3(color), MathF::Max(MathF::Sqrt(p.X * p.X + p.Z * p.Z) - Val(r), MathF::Abs(p.Y) - Val(h))); });
    }
    static SdfFunc Solid(std::function<Val(Vec3)> dist, Vector3 color = Vector3::One()) { return SdfFunc([dist, color](Vec3 p) { return Vec4(Vec3(color), dist(p)); }); }
    static SdfFunc Sphere(float r, Vector3 color = Vector3::One()) { return SdfFunc([r, color](Vec3 p) { return Vec4(Vec3(color), p.Length() - Val(r)); }); }
    static SdfFunc Union(SdfFunc a, SdfFunc b) { return SdfFuncs::Union(a, b); }
};

struct Sdfs {   // Sdf.cs:118-215 (batched delegates that only assign .W leave colour zero)
    static Sdf Box(Vector3 bounds) { return Sdf([bounds](Vec3 p) { return Vec4(Val(0.f), Val(0.f), Val(0.f), BoxDistance(p, bounds)); }, false); }
    static Sdf Box(float b) { return Box(Vector3(b)); }
    static Sdf Cylinder(float radius, float height) { return SdfExprs::Cylinder(radius, height).ToSdf(); }
    static Sdf Plane(Vector3 n, float d) { return Sdf([n, d](Vec3 p) { return Vec4(Val(0.f), Val(0.f), Val(0.f), Vec3::Dot(p, Vec3(n)) + Val(d)); }, false); }
    static Sdf PlaneXY(float z = 0) { return Plane(Vector3(0, 0, 1), z); }
    static Sdf PlaneXZ(float y = 0) { return Plane(Vector3(0, 1, 0), y); }
    static Sdf Solid(SdfFunc f) { return Sdf(f.fn, true); }
    static Sdf Solid(std::function<Val(Vec3)> dist, Vector3 color = Vector3::One()) { return Sdf([dist, color](Vec3 p) { return Vec4(Vec3(color), dist(p)); }, true); }
    static Sdf Sphere(float radius) { return Sdf([radius](Vec3 p) { return Vec4(Val(0.f), Val(0.f), Val(0.f), p.Length() - Val(radius)); }, false); }
};

// ---------------------------------------------------------------------------------------
// Mesh (Mesh.cs:8-64)
// ---------------------------------------------------------------------------------------
class Mesh {
public:
    std::vector<Vector3> Vertices, Colors, Normals;
    std::vector<int32_t> Triangles;
    Vector3 Min, Max;
    Vector3 Center() const { return (Min + Max) * 0.5f; }
    Vector3 Size() const { return Max - Min; }
    float Radius() const { return (Max - Min).Length() * 0.5f; }
    // Mesh.WriteObj (Mesh.cs:66-97): `v` lines, then `vn`, then `f a//a b//b c//c`, 1-based
    void WriteObj(std::ostream& w) const
    {
        for (const auto& v : Vertices) w << "v " << FormatSingle(v.X) << ' ' << FormatSingle(v.Y) << ' ' << FormatSingle(v.Z) << '\n';
        for (const auto& v : Normals) w << "vn " << FormatSingle(v.X) << ' ' << FormatSingle(v.Y) << ' ' << FormatSingle(v.Z) << '\n';
        for (size_t i = 0; i + 2 < Triangles.size(); i += 3) {
            const int a = Triangles[i] + 1, b = Triangles[i + 1] + 1, c = Triangles[i + 2] + 1;
            w << "f " << a << "//" << a << ' ' << b << "//" << b << ' ' << c << "//" << c << '\n';
        }
    }
    void WriteObj(const std::string& path) const { std::ofstream f(path); WriteObj(f); }
    // invariant-culture System.Single.ToString() of .NET Core 3.0+ (the runtime global.json pins): shortest round-trip
    // digits through format 'G'; scientific ("d.dddE+XX") when the decimal-point position (decimal exponent + 1) is
    // greater than max(number of digits, 7) or below -3 -- Number.Formatting.cs: nMaxDigits = Math.Max(number.DigitsCount,
    // SinglePrecision), then FormatGeneral's `digPos > nMaxDigits || digPos < -3`.  So 12345678f prints "12345678" (8 digits,
    // position 8), 1e7f prints "1E+07" (1 digit, position 8 > 7), 1e-5f prints "1E-05".
    static std::string FormatSingle(float x)
    {
        if (std::isnan(x)) return "NaN";
        if (std::isinf(x)) return x > 0 ? "Infinity" : "-Infinity";
        if (x == 0) return std::signbit(x) ? "-0" : "0";
        char buf[64];
        auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);   // shortest round-trip: d.ddde±XX
        std::string s(buf, r.ptr);
        const size_t ep = s.find('e');
        std::string mant = s.substr(0, ep);
        const int e = std::stoi(s.substr(ep + 1));
        const bool neg = mant[0] == '-';
        std::string digits;
        for (char c : mant) if (c >= '0' && c <= '9') digits.push_back(c);
        std::string out;
        const int max_digits = (int)std::max<size_t>(digits.size(), 7);
        if (e > -5 && e < max_digits) {
            if (e >= 0) {
                std::string ip = digits.substr(0, std::min(digits.size(), (size_t)e + 1));
                ip.append((size_t)e + 1 - ip.size(), '0');
                const std::string fp = digits.size() > (size_t)e + 1 ? digits.substr(e + 1) : "";
                out = ip + (fp.empty() ? "" : "." + fp);
            } else {
                out = "0." + std::string((size_t)(-e - 1), '0') + digits;
            }
        } else {
            char eb[16];
            snprintf(eb, sizeof eb, "%02d", e < 0 ? -e : e);
            out = digits.substr(0, 1) + (digits.size() > 1 ? "." + digits.substr(1) : "") + "E" + (e >= 0 ? "+" : "-") + eb;
        }
        return neg ? "-" + out : out;
    }
    static Mesh FromHandle(sdfk_mesh* h)
    {
        Mesh m;
        int64_t nv = 0, ni = 0;
        Check(sdfk_mesh_counts(h, &nv, &ni));
        m.Vertices.resize(nv); m.Colors.resize(nv); m.Normals.resize(nv); m.Triangles.resize(ni);
        Check(sdfk_mesh_copy(h, &m.Vertices.data()->X, &m.Colors.data()->X, &m.Normals.data()->X, m.Triangles.data()));
        Check(sdfk_mesh_bounds(h, &m.Min.X, &m.Max.X));
        sdfk_mesh_free(h);
        return m;
    }
};

inline void ReportProgress(const std::function<void(float)>& progress, int nz, int step)
{   // MarchingCubes.cs:53-81: one report per z layer, (float)z / (nz - 2*step)
    if (!progress) return;
    const int zb = nz - 2 * step;
    for (int z = -step; z < zb;) { z += step; progress((float)z / (float)zb); }
}

// ---------------------------------------------------------------------------------------
// Voxels (Voxels.cs)
// ---------------------------------------------------------------------------------------
class Voxels {
public:
    int NX, NY, NZ;
    float DX, DY, DZ;
    Vector3 Min, Max;
    Voxels(Vector3 min, Vector3 max, int nx, int ny, int nz) : NX(nx), NY(ny), NZ(nz), Min(min), Max(max)
    {
        DX = nx >= 1 ? (max.X - min.X) / (float)nx : 0.0f;
        DY = ny >= 1 ? (max.Y - min.Y) / (float)ny : 0.0f;
        DZ = nz >= 1 ? (max.Z - min.Z) / (float)nz : 0.0f;
    }
    Voxels(Voxels&& o) noexcept { *this = std::move(o); }
    Voxels& operator=(Voxels&& o) noexcept
    {
        NX = o.NX; NY = o.NY; NZ = o.NZ; DX = o.DX; DY = o.DY; DZ = o.DZ; Min = o.Min; Max = o.Max;
        h_ = o.h_; o.h_ = nullptr; hasColors_ = o.hasColors_; values_ = std::move(o.values_); hostNewer_ = o.hostNewer_;
        return *this;
    }
    ~Voxels() { if (h_) sdfk_volume_free(h_); }
    Vector3 Center() const { return (Min + Max) * 0.5f; }
    Vector3 Size() const { return Max - Min; }

    // Voxels.SampleSdf (instance, Voxels.cs:72-125; static, :169-174).  batchSize and
    // maxDegreeOfParallelism are accepted and ignored on the GPU.
    void SampleSdf(const Sdf& sdf, int batchSize = 2048, int maxDegreeOfParallelism = -1) { Sample(sdf, false); }
    static Voxels SampleSdf(const Sdf& sdf, Vector3 min, Vector3 max, int nx, int ny, int nz, int batchSize = 2048, int maxDegreeOfParallelism = -1)
    {
        Voxels v(min, max, nx, ny, nz);
        v.Sample(sdf, false);
        return v;
    }
    void Sample(const Sdf& sdf, bool clip)
    {
        Ensure(sdf.WritesColor());
        Check(sdfk_sample(sdf.Program(), h_, clip ? 1 : 0));
        values_.clear();
        hostNewer_ = false;
    }
    void ClipToBounds()   // Voxels.cs:133-167
    {
        Sync();
        Check(sdfk_volume_clip_to_bounds(h_));
        values_.clear();
    }
    // indexer v[ix,iy,iz] (Voxels.cs:42-46): host copy of Values, z fastest
    float& operator()(int ix, int iy, int iz)
    {
        if (values_.empty()) {
            values_.assign((size_t)NX * NY * NZ, 0.0f);
            if (h_) Check(sdfk_volume_download(h_, values_.data(), nullptr));
        }
        hostNewer_ = true;
        return values_[((size_t)ix * NY + iy) * NZ + iz];
    }
    inline Mesh ToMesh(float isoValue = 0.0f, int step = 1, const std::function<void(float)>& progress = nullptr);
    sdfk_volume* Sync()
    {
        Ensure(hasColors_);
        if (hostNewer_ && !values_.empty()) { Check(sdfk_volume_upload(h_, values_.data(), nullptr)); hostNewer_ = false; }
        return h_;
    }

private:
    void Ensure(bool colors)
    {
        EnsureInit();
        if (h_ && colors && !hasColors_) { sdfk_volume_free(h_); h_ = nullptr; }
        if (!h_) { Check(sdfk_volume_create(NX, NY, NZ, &Min.X, &Max.X, colors ? 1 : 0, &h_)); hasColors_ = colors; }
    }
    sdfk_volume* h_ = nullptr;
    bool hasColors_ = false;
    std::vector<float> values_;
    bool hostNewer_ = false;
};

struct MarchingCubes {
    // MarchingCubes.CreateMesh(Voxels, isoValue = 0, step = 1, progress = null)  (MarchingCubes.cs:39)
    static Mesh CreateMesh(Voxels& volume, float isoValue = 0.0f, int step = 1, const std::function<void(float)>& progress = nullptr)
    {
        sdfk_mesh* m = nullptr;
        Check(sdfk_march(volume.Sync(), isoValue, step, &m));
        ReportProgress(progress, volume.NZ, step);
        return Mesh::FromHandle(m);
    }
};

inline Mesh Voxels::ToMesh(float isoValue, int step, const std::function<void(float)>& progress) { return MarchingCubes::CreateMesh(*this, isoValue, step, progress); }

inline Voxels Sdf::ToVoxels(Vector3 min, Vector3 max, int nx, int ny, int nz, int, int, bool clipToBounds) const
{
    Voxels v(min, max, nx, ny, nz);
    v.Sample(*this, clipToBounds);
    return v;
}

inline Mesh Sdf::ToMesh(Vector3 min, Vector3 max, int nx, int ny, int nz, int, int, bool clipToBounds, float isoValue, int step,
                        const std::function<void(float)>& progress) const
{
    EnsureInit();
    sdfk_mesh* m = nullptr;
    Check(sdfk_sample_march(Program(), &min.X, &max.X, nx, ny, nz, clipToBounds ? 1 : 0, isoValue, step, &m));
    ReportProgress(progress, nz, step);
    return Mesh::FromHandle(m);
}

// ---------------------------------------------------------------------------------------
// Node: SdfEx.ToMesh (Sdf.cs:59-63) over several GPUs from THIS process (sdfk_node_*)
// ---------------------------------------------------------------------------------------
// Every listed device gets a context and a rank thread of the library's own; ToMesh fills the caller's four exact-length arrays:
// every GPU copies its own slab into its slice over its own PCIe link (the mesh stays sharded until it is on the host).
class Node {
public:
    explicit Node(const std::vector<int32_t>& devices = {})
    {
        Check(sdfk_node_open(devices.empty() ? nullptr : devices.data(), (int32_t)devices.size(), &h_));
        int32_t backend = 0;
        Check(sdfk_node_info(h_, &world_, &backend));
    }
    ~Node() { if (h_) sdfk_node_close(h_); }
    Node(const Node&) = delete;
    Node& operator=(const Node&) = delete;
    int World() const { return world_; }
    Mesh ToMesh(const Sdf& sdf, Vector3 min, Vector3 max, int nx, int ny, int nz, bool clipToBounds = true, float isoValue = 0.0f) const
    {
        std::vector<sdfk_op> ops;
        int32_t out[4];
        sdf.Lower(ops, out);
        int64_t nv = 0, ni = 0;
        int32_t hasColors = 0;
        Check(sdfk_node_mesh_begin(h_, ops.data(), (int32_t)ops.size(), out, sdf.WritesColor() ? 1 : 0, &min.X, &max.X, nx, ny, nz,
                                   clipToBounds ? 1 : 0, isoValue, &nv, &ni, &hasColors));
        Mesh m;
        m.Vertices.resize(nv); m.Colors.resize(nv); m.Normals.resize(nv); m.Triangles.resize(ni);   // (value-initialised: zero colours already)
        Check(sdfk_node_mesh_copy(h_, nv ? &m.Vertices.data()->X : nullptr, (nv && hasColors) ? &m.Colors.data()->X : nullptr,
                                  nv ? &m.Normals.data()->X : nullptr, ni ? m.Triangles.data() : nullptr, &m.Min.X, &m.Max.X));
        return m;
    }

private:
    sdfk_node* h_ = nullptr;
    int32_t world_ = 1;
};

// ---------------------------------------------------------------------------------------
// RayMarcher (RayMarcher.cs) + the image containers it returns (VectorData.cs)
// ---------------------------------------------------------------------------------------
// The members of System.Numerics.Matrix4x4 the RayMarcher uses, software float32 forms
// (row-major M[r][c] = M(r+1)(c+1)); compile with -ffp-contract=off.
struct Matrix4x4 {
    float M[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    static Vector3 Normalize(Vector3 v) { const float l = std::sqrt((v.X * v.X + v.Y * v.Y) + v.Z * v.Z); return {v.X / l, v.Y / l, v.Z / l}; }
    static Vector3 Cross(Vector3 a, Vector3 b) { return {a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X}; }
    static float Dot(Vector3 a, Vector3 b) { return (a.X * b.X + a.Y * b.Y) + a.Z * b.Z; }
    static Matrix4x4 CreateLookAt(Vector3 pos, Vector3 target, Vector3 up)
    {
        const Vector3 z = Normalize(pos - target), x = Normalize(Cross(up, z)), y = Cross(z, x);
        Matrix4x4 r;
        const float m[4][4] = {{x.X, y.X, z.X, 0}, {x.Y, y.Y, z.Y, 0}, {x.Z, y.Z, z.Z, 0}, {-Dot(x, pos), -Dot(y, pos), -Dot(z, pos), 1}};
        std::memcpy(r.M, m, sizeof m);
        return r;
    }
    static Matrix4x4 CreatePerspectiveFieldOfView(float fov, float aspect, float nearp, float farp)
    {
        const float ys = 1.0f / std::tan(fov * 0.5f), xs = ys / aspect;
        const float nfr = (std::isinf(farp) && farp > 0) ? -1.0f : farp / (nearp - farp);
        Matrix4x4 r;
        std::memset(r.M, 0, sizeof r.M);
        r.M[0][0] = xs; r.M[1][1] = ys; r.M[2][2] = nfr; r.M[2][3] = -1.0f; r.M[3][2] = nearp * nfr;
        return r;
    }
    friend Matrix4x4 operator*(const Matrix4x4& a, const Matrix4x4& b)
    {
        Matrix4x4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                r.M[i][j] = ((a.M[i][0] * b.M[0][j] + a.M[i][1] * b.M[1][j]) + a.M[i][2] * b.M[2][j]) + a.M[i][3] * b.M[3][j];
        return r;
    }
    static bool Invert(const Matrix4x4& s, Matrix4x4& out)
    {
        const float a = s.M[0][0], b = s.M[0][1], c = s.M[0][2], d = s.M[0][3], e = s.M[1][0], f = s.M[1][1], g = s.M[1][2], h = s.M[1][3];
        const float i = s.M[2][0], j = s.M[2][1], k = s.M[2][2], l = s.M[2][3], m = s.M[3][0], n = s.M[3][1], o = s.M[3][2], p = s.M[3][3];
        const float kp_lo = k * p - l * o, jp_ln = j * p - l * n, jo_kn = j * o - k * n, ip_lm = i * p - l * m, io_km = i * o - k * m, in_jm = i * n - j * m;
        const float a11 = +(f * kp_lo - g * jp_ln + h * jo_kn), a12 = -(e * kp_lo - g * ip_lm + h * io_km);
        const float a13 = +(e * jp_ln - f * ip_lm + h * in_jm), a14 = -(e * jo_kn - f * io_km + g * in_jm);
        const float det = a * a11 + b * a12 + c * a13 + d * a14;
        if (std::fabs(det) < 1.1920929e-07f) { for (auto& row : out.M) for (float& v : row) v = NAN; return false; }
        const float inv = 1.0f / det;
        float (*R)[4] = out.M;
        R[0][0] = a11 * inv; R[1][0] = a12 * inv; R[2][0] = a13 * inv; R[3][0] = a14 * inv;
        R[0][1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv; R[1][1] = +(a * kp_lo - c * ip_lm + d * io_km) * inv;
        R[2][1] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv; R[3][1] = +(a * jo_kn - b * io_km + c * in_jm) * inv;
        const float gp_ho = g * p - h * o, fp_hn = f * p - h * n, fo_gn = f * o - g * n, ep_hm = e * p - h * m, eo_gm = e * o - g * m, en_fm = e * n - f * m;
        R[0][2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv; R[1][2] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv;
        R[2][2] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv; R[3][2] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv;
        const float gl_hk = g * l - h * k, fl_hj = f * l - h * j, fk_gj = f * k - g * j, el_hi = e * l - h * i, ek_gi = e * k - g * i, ej_fi = e * j - f * i;
        R[0][3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv; R[1][3] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv;
        R[2][3] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv; R[3][3] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv;
        return true;
    }
};

inline void WriteTgaHeader(std::ostream& w, int imageType, int width, int height, int bpp)
{   // VectorData.cs:246-261: 18-byte header, top-down
    const unsigned char h[18] = {0, 0, (unsigned char)imageType, 0, 0, 0, 0, 0, 0, 0, 0, 0, (unsigned char)(width & 255), (unsigned char)(width >> 8),
                                 (unsigned char)(height & 255), (unsigned char)(height >> 8), (unsigned char)bpp, 0x20};
    w.write(reinterpret_cast<const char*>(h), 18);
}
struct FloatData {   // VectorData.cs:137-280; indexer is (x, y)
    int Width = 0, Height = 0;
    std::vector<float> Values;
    float operator()(int x, int y) const { return Values[(size_t)y * Width + x]; }
    void SaveDepthTga(const std::string& path, float nearv, float farv) const   // VectorData.cs:244-279
    {
        std::ofstream w(path, std::ios::binary);
        WriteTgaHeader(w, 3, Width, Height, 8);
        for (float v : Values) {
            const unsigned char b = v >= farv ? 0 : (v <= nearv ? 255 : (unsigned char)(255.0f * (farv - v) / (farv - nearv)));
            w.put((char)b);
        }
    }
};
struct Vec3Data {    // VectorData.cs:343-620
    int Width = 0, Height = 0;
    std::vector<float> Values;   // 3 per pixel
    Vector3 operator()(int x, int y) const { const float* p = &Values[((size_t)y * Width + x) * 3]; return {p[0], p[1], p[2]}; }
    void SaveTga(const std::string& path) const   // VectorData.cs:570-619: 24-bit B, G, R, channel * 255 truncated and clamped
    {
        std::ofstream w(path, std::ios::binary);
        WriteTgaHeader(w, 2, Width, Height, 24);
        for (size_t k = 0; k + 2 < Values.size(); k += 3)
            for (int c = 2; c >= 0; c--) {
                const float v = Values[k + c] * 255.0f;
                w.put((char)(v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (unsigned char)v)));
            }
    }
};

class RayMarcher {   // RayMarcher.cs:7-43: same constructor, properties and defaults
public:
    static constexpr float DefaultNearPlaneDistance = 1.0f, DefaultFarPlaneDistance = 100.0f, DefaultVerticalFieldOfViewDegrees = 60.0f;
    static constexpr int DefaultDepthIterations = 40;
    Matrix4x4 ViewTransform = Matrix4x4::CreateLookAt(Vector3(0, 0, 5), Vector3(0, 0, 0), Vector3(0, 1, 0));
    float NearPlaneDistance = DefaultNearPlaneDistance, FarPlaneDistance = DefaultFarPlaneDistance;
    float VerticalFieldOfViewDegrees = DefaultVerticalFieldOfViewDegrees;
    int DepthIterations = DefaultDepthIterations;
    RayMarcher(int width, int height, Sdf sdf, int /*batchSize*/ = 2048, int /*maxDegreeOfParallelism*/ = -1) : w_(width), h_(height), sdf_(std::move(sdf)) {}
    Vec3Data Render() const { Vec3Data d; d.Width = w_; d.Height = h_; d.Values.resize((size_t)w_ * h_ * 3); Run(nullptr, d.Values.data()); return d; }
    FloatData RenderDepth() const { FloatData d; d.Width = w_; d.Height = h_; d.Values.resize((size_t)w_ * h_); Run(d.Values.data(), nullptr); return d; }

private:
    void Run(float* depth, float* rgb) const
    {
        // host part of GetCameraRays (RayMarcher.cs:97-112)
        Matrix4x4 cam, vpi;
        Matrix4x4::Invert(ViewTransform, cam);
        const float pos[3] = {((0.0f * cam.M[0][0] + 0.0f * cam.M[1][0]) + 0.0f * cam.M[2][0]) + cam.M[3][0],
                              ((0.0f * cam.M[0][1] + 0.0f * cam.M[1][1]) + 0.0f * cam.M[2][1]) + cam.M[3][1],
                              ((0.0f * cam.M[0][2] + 0.0f * cam.M[1][2]) + 0.0f * cam.M[2][2]) + cam.M[3][2]};
        const Matrix4x4 proj = Matrix4x4::CreatePerspectiveFieldOfView(VerticalFieldOfViewDegrees * 3.14159274f / 180.0f, (float)w_ / (float)h_,
                                                                       NearPlaneDistance, FarPlaneDistance);
        Matrix4x4::Invert(ViewTransform * proj, vpi);
        EnsureInit();
        Check(sdfk_raymarch(sdf_.Program(), w_, h_, pos, &vpi.M[0][0], NearPlaneDistance, FarPlaneDistance, DepthIterations, depth, rgb));
    }
    int w_, h_;
    Sdf sdf_;
};

}  // namespace SdfKit
