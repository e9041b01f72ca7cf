// mc_decide.h -- the Lewiner case dispatcher as pure functions of a cell's eight corner values: the lookup blob, the
// decision-table blob, TestFace / TestInternal / TheBigSwitch (MarchingCubes.cs:94-546).  Included by mc_device.h for the
// kernels.  The function and table qualifiers (__device__, __forceinline__, __constant__) come from whoever includes this
// file: <hip/hip_runtime.h> in the product (mc_device.h); tests/cpp/dispatch_host.cpp defines them away itself to instantiate
// the very same text with a host compiler for the CPU suite (tests/test_dispatch_manifest.py).  Nothing here is HIP-specific.
//
// All arithmetic is double precision exactly like the reference (Cell.cs:74,191-208: float voxels are widened, the iso value
// is subtracted in double); compile with -ffp-contract=off: `A*C - B*D` must round as two multiplies and a subtract.
//
// Corner order v0..v7 (Luts.cs:30-52): v0=(x,y,z) v1=(x+1,y,z) v2=(x+1,y+1,z) v3=(x,y+1,z),
// v4..v7 the same at z+1.  "Bit order" index = dz*4+dy*2+dx (Cell.cs:318-319).
#pragma once
#include <stdint.h>
#include "mc_luts.h"

namespace sdfk {

// The packed table blob lives in __constant__ memory; kernels copy it into LDS once per
// workgroup (13.5 KB) because the per-cell decisions are chains of dependent table reads
// with run-time (divergent) indices -- LDS latency instead of vector-cache latency.
constexpr int MCLUT_PADDED = (MCLUT_BLOB_SIZE + 15) & ~15;   // copied in 16-byte pieces
__constant__ __attribute__((aligned(16))) int8_t c_lut[MCLUT_PADDED] = {MCLUT_BLOB_VALUES};


// ---- the DECISION tables: what the dispatcher reads -------------------------------------------------------------------
// mc_resolve / mc_test_internal read the case table, the test tables, subconfig13 and (one entry per row of) tiling13_5_1 --
// 1.1 KB of the 13.5 KB blob; the tiling rows themselves are only read by k_triangles.  They get a blob of their own, built
// at compile time from the same values, so that k_resolve copies 1.1 KB to LDS per workgroup instead of 13.5 KB (2280
// workgroups at 512^3: 2.5 instead of 31 MB of L2 -> LDS traffic, 15 instead of 27.5 KB of LDS per workgroup).
// Triangle-row OFFSETS (Tiling::lut_off) stay offsets into the full blob.
constexpr int MCDEC_OFF_cases = 0;
constexpr int MCDEC_OFF_test3 = MCDEC_OFF_cases + MCLUT_DIM0_cases * MCLUT_DIM1_cases;
constexpr int MCDEC_OFF_test4 = MCDEC_OFF_test3 + MCLUT_DIM0_test3;
constexpr int MCDEC_OFF_test6 = MCDEC_OFF_test4 + MCLUT_DIM0_test4;
constexpr int MCDEC_OFF_test7 = MCDEC_OFF_test6 + MCLUT_DIM0_test6 * MCLUT_DIM1_test6;
constexpr int MCDEC_OFF_test10 = MCDEC_OFF_test7 + MCLUT_DIM0_test7 * MCLUT_DIM1_test7;
constexpr int MCDEC_OFF_test12 = MCDEC_OFF_test10 + MCLUT_DIM0_test10 * MCLUT_DIM1_test10;
constexpr int MCDEC_OFF_test13 = MCDEC_OFF_test12 + MCLUT_DIM0_test12 * MCLUT_DIM1_test12;
constexpr int MCDEC_OFF_subconfig13 = MCDEC_OFF_test13 + MCLUT_DIM0_test13 * MCLUT_DIM1_test13;
constexpr int MCDEC_OFF_tiling13_5_1 = MCDEC_OFF_subconfig13 + MCLUT_DIM0_subconfig13;
constexpr int MCDEC_SIZE = MCDEC_OFF_tiling13_5_1 + MCLUT_DIM0_tiling13_5_1 * MCLUT_DIM1_tiling13_5_1 * MCLUT_DIM2_tiling13_5_1;
constexpr int MCDEC_PADDED = (MCDEC_SIZE + 15) & ~15;   // copied in 16-byte pieces
struct alignas(16) McDecBlob { int8_t v[MCDEC_PADDED]; };
constexpr McDecBlob mc_make_dec()
{
    constexpr int8_t full[] = {MCLUT_BLOB_VALUES};
    McDecBlob d{};
    constexpr int src[10] = {MCLUT_OFF_cases, MCLUT_OFF_test3, MCLUT_OFF_test4, MCLUT_OFF_test6, MCLUT_OFF_test7, MCLUT_OFF_test10,
                             MCLUT_OFF_test12, MCLUT_OFF_test13, MCLUT_OFF_subconfig13, MCLUT_OFF_tiling13_5_1};
    constexpr int dst[11] = {MCDEC_OFF_cases, MCDEC_OFF_test3, MCDEC_OFF_test4, MCDEC_OFF_test6, MCDEC_OFF_test7, MCDEC_OFF_test10,
                             MCDEC_OFF_test12, MCDEC_OFF_test13, MCDEC_OFF_subconfig13, MCDEC_OFF_tiling13_5_1, MCDEC_SIZE};
    for (int t = 0; t < 10; t++)
        for (int i = 0; i < dst[t + 1] - dst[t]; i++) d.v[dst[t] + i] = full[src[t] + i];
    return d;
}
__constant__ McDecBlob c_dec = mc_make_dec();


// every function below takes `lut` = base of the DECISION blob (LDS or constant) for what it reads; triangle-row offsets
// (MC_ROW2 / MC_ROW3) are offsets into the full blob c_lut
#define MC_L1(name, i) (lut[MCDEC_OFF_##name + (i)])
#define MC_L2(name, i, j) (lut[MCDEC_OFF_##name + (i) * MCLUT_DIM1_##name + (j)])
#define MC_DROW3(name, i, j) (MCDEC_OFF_##name + ((i) * MCLUT_DIM1_##name + (j)) * MCLUT_DIM2_##name)
#define MC_ROW2(name, i) (MCLUT_OFF_##name + (i) * MCLUT_DIM1_##name)
#define MC_ROW3(name, i, j) (MCLUT_OFF_##name + ((i) * MCLUT_DIM1_##name + (j)) * MCLUT_DIM2_##name)

// MarchingCubes.cs:37 and Cell.cs:63: a DOUBLE literal, not C's FLT_EPSILON.
#define MC_EPS 0.0000001

struct Tiling {
    int lut_off;  // start of the triangle row in the blob, -1 when nothing is emitted
    int row;      // row id (index into c_rowocc / MCLUT_ROWOFF), valid when nt > 0
    int nt;       // triangles
    int index;    // 8-bit corner sign word (Cell.cs:220-229)
};
#define MC_PICK2(name, i, ntv) do { r.lut_off = MC_ROW2(name, i); r.row = MCLUT_ROWBASE_##name + (i); r.nt = (ntv); } while (0)
#define MC_PICK3(name, i, j, ntv) do { r.lut_off = MC_ROW3(name, i, j); r.row = MCLUT_ROWBASE_##name + (i) * MCLUT_DIM1_##name + (j); r.nt = (ntv); } while (0)

// Per triangle row: how often it references each vertex id 0..12 (4 bits each).  Derived
// from the tiling tables by tools/gen_luts.py; replaces scanning the row (Cell.cs:238-265
// visits every entry) when only the NUMBER of references to one edge is needed.
__constant__ uint64_t c_rowocc[MCLUT_NROWS] = {MCLUT_ROWOCC_VALUES};
__constant__ uint8_t c_rownt[MCLUT_NROWS] = {MCLUT_ROWNT_VALUES};   // triangles of each row
// distinct vertex ids of each row in the order of their first reference (4 bits each from bit 0, their
// number in bits 60..63): the order in which a cell creates the vertices it owns (Cell.cs:272-359)
__constant__ uint64_t c_roword[MCLUT_NROWS] = {MCLUT_ROWORD_VALUES};

// Corner accessors.  The Lewiner tables index the eight corners with run-time indices; a
// per-thread register array indexed that way is demoted to scratch memory by the compiler,
// so kernels keep each thread's corners in its own LDS column instead (conflict-free:
// consecutive lanes -> consecutive banks) and hand the decision functions an accessor.
// (ISO0: the iso value is known to be +0.0 -- x - 0.0 is x for every x, -0.0 and NaN included, so the subtraction, one f64
// operation per corner read, is left out; k_vertices reads ~58 corners per vertex)
template <bool ISO0>
struct CornersLdsT {          // float voxels in LDS, [corner][thread] with `stride` threads
    const float* p;
    int stride;
    double iso;
    __device__ __forceinline__ double operator[](int k) const { return ISO0 ? (double)p[k * stride] : (double)p[k * stride] - iso; }
};
using CornersLds = CornersLdsT<false>;
// Corners of one cell inside a 3x3x3 voxel block staged per thread in LDS
// ([(lx*3+ly)*3+lz][256 threads]); (ox,oy,oz) = the cell's origin inside the block.
struct CornersNbr {
    const float* p;
    int ox, oy, oz;
    double iso;
    __device__ __forceinline__ double operator[](int k) const
    {
        const int dx = ((k + 1) >> 1) & 1, dy = (k >> 1) & 1, dz = k >> 2;   // corner k of Luts.cs:30-52
        return (double)p[(((ox + dx) * 3 + (oy + dy)) * 3 + (oz + dz)) * 256] - iso;
    }
};
struct CornersPtr {          // plain array of iso-subtracted doubles
    const double* p;
    __device__ __forceinline__ double operator[](int k) const { return p[k]; }
};

// MarchingCubes.cs:376-407
template <class V>
__device__ __forceinline__ bool mc_test_face(const V& v, int face)
{
    const int af = face < 0 ? -face : face;
    double A = 0, B = 0, C = 0, D = 0;
    switch (af) {
    case 1: A = v[0]; B = v[4]; C = v[5]; D = v[1]; break;
    case 2: A = v[1]; B = v[5]; C = v[6]; D = v[2]; break;
    case 3: A = v[2]; B = v[6]; C = v[7]; D = v[3]; break;
    case 4: A = v[3]; B = v[7]; C = v[4]; D = v[0]; break;
    case 5: A = v[0]; B = v[3]; C = v[2]; D = v[1]; break;
    case 6: A = v[4]; B = v[7]; C = v[6]; D = v[5]; break;
    default: break;
    }
    const double acbd = A * C - B * D;
    if (acbd > -MC_EPS && acbd < MC_EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

// Reference-edge lerp table of MarchingCubes.cs:440-511:
// t = v[a]/(v[a]-v[b]+eps); Bt = v[B0]+(v[B1]-v[B0])*t; Ct, Dt likewise.
__constant__ int8_t c_interior_edges[12][8] = {
    {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
    {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
    {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
    {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

// MarchingCubes.cs:412-546
template <class V>
__device__ __forceinline__ bool mc_test_internal(const int8_t* lut, const V& v, int cas, int config, int subconfig, int s)
{
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0;
    if (cas == 4 || cas == 10) {
        const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else {
        int edge;
        if (cas == 6) edge = MC_L2(test6, config, 2);
        else if (cas == 7) edge = MC_L2(test7, config, 4);
        else if (cas == 12) edge = MC_L2(test12, config, 3);
        else edge = lut[MC_DROW3(tiling13_5_1, config, subconfig)];
        if (edge >= 0 && edge < 12) {
            const int8_t* e = c_interior_edges[edge];
            t = v[e[0]] / (v[e[0]] - v[e[1]] + MC_EPS);
            At = 0;
            Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
            Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
            Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
        }
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) {  // MarchingCubes.cs:526-545
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 5: if (At * Ct - Bt * Dt < MC_EPS) return s > 0; break;
    case 10: if (At * Ct - Bt * Dt >= MC_EPS) return s > 0; break;
    default: return s < 0;  // 7, 11, 13, 14, 15
    }
    return s < 0;
}

// The 33-case dispatcher of MarchingCubes.cs:94-371 as a pure function of the corners.
template <class V>
__device__ __forceinline__ Tiling mc_resolve(const int8_t* lut, const V& v)
{
    Tiling r;
    int index = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) index |= (v[k] > 0.0) ? (1 << k) : 0;
    r.index = index;
    r.lut_off = -1;
    r.row = 0;
    r.nt = 0;
    const int cas = MC_L2(cases, index, 0);
    const int cfg = MC_L2(cases, index, 1);
    int sub = 0;
    switch (cas) {
    case 1: MC_PICK2(tiling1, cfg, 1); break;
    case 2: MC_PICK2(tiling2, cfg, 2); break;
    case 3:
        if (mc_test_face(v, MC_L1(test3, cfg))) { MC_PICK2(tiling3_2, cfg, 4); }
        else { MC_PICK2(tiling3_1, cfg, 2); }
        break;
    case 4:
        if (mc_test_internal(lut, v, cas, cfg, 0, MC_L1(test4, cfg))) { MC_PICK2(tiling4_1, cfg, 2); }
        else { MC_PICK2(tiling4_2, cfg, 6); }
        break;
    case 5: MC_PICK2(tiling5, cfg, 3); break;
    case 6:
        if (mc_test_face(v, MC_L2(test6, cfg, 0))) { MC_PICK2(tiling6_2, cfg, 5); }
        else if (mc_test_internal(lut, v, cas, cfg, 0, MC_L2(test6, cfg, 1))) { MC_PICK2(tiling6_1_1, cfg, 3); }
        else { MC_PICK2(tiling6_1_2, cfg, 9); }
        break;
    case 7:
        if (mc_test_face(v, MC_L2(test7, cfg, 0))) sub += 1;
        if (mc_test_face(v, MC_L2(test7, cfg, 1))) sub += 2;
        if (mc_test_face(v, MC_L2(test7, cfg, 2))) sub += 4;
        switch (sub) {
        case 0: MC_PICK2(tiling7_1, cfg, 3); break;
        case 1: MC_PICK3(tiling7_2, cfg, 0, 5); break;
        case 2: MC_PICK3(tiling7_2, cfg, 1, 5); break;
        case 3: MC_PICK3(tiling7_3, cfg, 0, 9); break;
        case 4: MC_PICK3(tiling7_2, cfg, 2, 5); break;
        case 5: MC_PICK3(tiling7_3, cfg, 1, 9); break;
        case 6: MC_PICK3(tiling7_3, cfg, 2, 9); break;
        default:
            if (mc_test_internal(lut, v, cas, cfg, sub, MC_L2(test7, cfg, 3))) { MC_PICK2(tiling7_4_2, cfg, 9); }
            else { MC_PICK2(tiling7_4_1, cfg, 5); }
            break;
        }
        break;
    case 8: MC_PICK2(tiling8, cfg, 2); break;
    case 9: MC_PICK2(tiling9, cfg, 4); break;
    case 10:
        if (mc_test_face(v, MC_L2(test10, cfg, 0))) {
            if (mc_test_face(v, MC_L2(test10, cfg, 1))) { MC_PICK2(tiling10_1_1_, cfg, 4); }
            else { MC_PICK2(tiling10_2, cfg, 8); }
        } else {
            if (mc_test_face(v, MC_L2(test10, cfg, 1))) { MC_PICK2(tiling10_2_, cfg, 8); }
            else if (mc_test_internal(lut, v, cas, cfg, 0, MC_L2(test10, cfg, 2))) { MC_PICK2(tiling10_1_1, cfg, 4); }
            else { MC_PICK2(tiling10_1_2, cfg, 8); }
        }
        break;
    case 11: MC_PICK2(tiling11, cfg, 4); break;
    case 12:
        if (mc_test_face(v, MC_L2(test12, cfg, 0))) {
            if (mc_test_face(v, MC_L2(test12, cfg, 1))) { MC_PICK2(tiling12_1_1_, cfg, 4); }
            else { MC_PICK2(tiling12_2, cfg, 8); }
        } else {
            if (mc_test_face(v, MC_L2(test12, cfg, 1))) { MC_PICK2(tiling12_2_, cfg, 8); }
            else if (mc_test_internal(lut, v, cas, cfg, 0, MC_L2(test12, cfg, 2))) { MC_PICK2(tiling12_1_1, cfg, 4); }
            else { MC_PICK2(tiling12_1_2, cfg, 8); }
        }
        break;
    case 13: {
#pragma unroll 1
        for (int k = 0; k < 6; k++)
            if (mc_test_face(v, MC_L2(test13, cfg, k))) sub += 1 << k;
        sub = MC_L1(subconfig13, sub);
        if (sub == 0) { MC_PICK2(tiling13_1, cfg, 4); }
        else if (sub >= 1 && sub <= 6) { MC_PICK3(tiling13_2, cfg, sub - 1, 6); }
        else if (sub >= 7 && sub <= 18) { MC_PICK3(tiling13_3, cfg, sub - 7, 10); }
        else if (sub >= 19 && sub <= 22) { MC_PICK3(tiling13_4, cfg, sub - 19, 12); }
        else if (sub >= 23 && sub <= 26) {
            const int s5 = sub - 23;
            if (mc_test_internal(lut, v, cas, cfg, s5, MC_L2(test13, cfg, 6))) { MC_PICK3(tiling13_5_1, cfg, s5, 6); }
            else { MC_PICK3(tiling13_5_2, cfg, s5, 10); }
        }
        else if (sub >= 27 && sub <= 38) { MC_PICK3(tiling13_3_, cfg, sub - 27, 10); }
        else if (sub >= 39 && sub <= 44) { MC_PICK3(tiling13_2_, cfg, sub - 39, 6); }
        else if (sub == 45) { MC_PICK2(tiling13_1_, cfg, 4); }
        // else: "Impossible case 13?" (MarchingCubes.cs:365) -- the cell emits nothing
        break;
    }
    case 14: MC_PICK2(tiling14, cfg, 4); break;
    default: break;
    }
    return r;
}

}  // namespace sdfk
