// sample_codegen.h -- lowers an SDF program (flat SSA list of float32 ops, see
// include/sdfkit_hip.h) to HIP source for hiprtc.  This is the GPU counterpart of the
// reference's SdfExprCompiler (SdfExpr.cs:225-273), which wraps a per-point expression
// in a batch loop and JIT-compiles it; here the "batch loop" is the grid-sampling kernel
// of Voxels.SampleSdf (Voxels.cs:72-125) with ClipToBounds (Voxels.cs:133-167) fused in.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdfkit_hip.h"

#define SDFK_STR_(...) #__VA_ARGS__
#define SDFK_STR(...) SDFK_STR_(__VA_ARGS__)
// One definition, used both by the host (below) and pasted into the generated source.
#define SDFK_SAMPLE_ARGS_BODY                                                                   \
    float* values; float* colors; float mx, my, mz, dx, dy, dz; int nx, ny, nz; int z0, nz_global; \
    int clip; float outside; int pitch8 /* row pitch of values, colors (x3) and bits8 */; int pad0; unsigned char* bits8; int nx8; float iso;

struct SampleArgs { SDFK_SAMPLE_ARGS_BODY };
struct RayArgs { float* depth; float* rgb; float cam[3]; float m[16]; int width, height; float nearp, farp; int iters; };

namespace sdfk {

static const char* const kSamplePrelude =
    "struct SampleArgs { " SDFK_STR(SDFK_SAMPLE_ARGS_BODY) " };\n"
    R"SRC(
// Math.Max / MathF.Max and Math.Min / MathF.Min: IEEE 754:2019 maximum / minimum (NaN if either
// operand is NaN, -0 < +0) -- on gfx950 one instruction each (v_maximum3_f32 / v_minimum3_f32).
__device__ __forceinline__ float sdfk_max_ieee(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float sdfk_min_ieee(float a, float b) { return __builtin_elementwise_minimum(a, b); }
// MathF.Sqrt, correctly rounded.  The general expansion (operand scaling for results near the
// denormal range, pass-through of 0 / inf / NaN) costs 17 instructions per call, and the sampling
// kernel of a cheap SDF is bound by its VALU work as much as by its stores.  When EVERY lane's
// operand is a finite number >= 2^-96 -- any sample point that is not within 4e-15 of a primitive's
// centre -- five operations give the same result: y = v_rsq_f32(x), g = x y, h = y / 2, then ONE
// correction step with exact FMA residuals, g + (x - g g) h.  That this is the correctly rounded
// root for every float in [2^-96, FLT_MAX] is not a theorem (LLVM's own lowering spends a Goldschmidt
// step more to have one) but a fact about gfx950's v_rsq_f32, established exhaustively:
// tools/ubench/ub_sqrt.hip compares all 1 879 048 192 operands with the fp64 root rounded once
// (0 mismatches; the same check runs in tests/test_gpu_parity.py::test_sqrt_whole_float_range).
// 8 issue slots (the transcendental counts 4) instead of the 12 of "v_sqrt_f32, then test the two
// neighbours" (round 1).  A wavefront with any other operand takes the general expansion as a whole
// (wave-uniform branch).
__device__ __forceinline__ float sdfk_sqrt(float x)
{
    // (false for NaN, zero, negatives, +inf -- rsq(inf) = 0 would turn the root into NaN)
    const bool easy = (x >= 0x1p-96f) & (x < __builtin_inff());
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!easy) != 0, 0)) return __builtin_sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// ---- interval form of the program's operations (SDFK_OPT_ELIDE_VOLUME: block culling, sdfk_cull_blocks below) ---------------
// An interval [lo, hi] that CONTAINS every float32 value the operation can produce for operands inside the operand intervals.
// No outward rounding is needed: round-to-nearest is monotone, so for + - * / sqrt floor the extreme FLOAT results are reached at
// the corners of the operand box and are computed here with the very same float operations.  NaN means "unknown": minimum /
// maximum propagate it (IEEE 754:2019 minimum / maximum), an interval whose divisor contains zero is made NaN, a square root of
// an interval reaching below zero is NaN by itself, and the caller evaluates every voxel of a block whose result is NaN.
struct sdfk_iv { float lo, hi; };
// "Unknown" is a property of the WHOLE interval: an operation that can produce NaN for SOME operand pair inside the box poisons both
// endpoints, never one (a half-poisoned [NaN, 2] would pass `hi < x` tests although the points whose value is NaN compare false).
// That covers: NaN endpoints; sqrt of an interval reaching below zero; a divisor interval containing zero; inf - inf, inf + -inf and
// 0 * inf, which need not show at the corners of the operand box ([-1, 1] * [inf, inf] has corner products -inf and +inf only).
__device__ __forceinline__ sdfk_iv iv_nan() { sdfk_iv r; r.lo = __builtin_nanf(""); r.hi = r.lo; return r; }
__device__ __forceinline__ sdfk_iv iv_whole(sdfk_iv r) { if (r.lo != r.lo || r.hi != r.hi) return iv_nan(); return r; }
__device__ __forceinline__ bool iv_unknown(sdfk_iv a) { return a.lo != a.lo || a.hi != a.hi; }
__device__ __forceinline__ bool iv_has_zero(sdfk_iv a) { return a.lo <= 0.0f && a.hi >= 0.0f; }
__device__ __forceinline__ bool iv_has_inf(sdfk_iv a) { return __builtin_isinf(a.lo) || __builtin_isinf(a.hi); }
__device__ __forceinline__ sdfk_iv iv_make(float a, float b) { sdfk_iv r; r.lo = sdfk_min_ieee(a, b); r.hi = sdfk_max_ieee(a, b); return r; }
__device__ __forceinline__ sdfk_iv iv_const(float c) { sdfk_iv r; r.lo = c; r.hi = c; return r; }
__device__ __forceinline__ sdfk_iv iv_add(sdfk_iv a, sdfk_iv b)
{
    const float inf = __builtin_inff();
    if ((a.hi == inf && b.lo == -inf) || (a.lo == -inf && b.hi == inf)) return iv_nan();   // inf + -inf somewhere in the box
    sdfk_iv r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi; return iv_whole(r);
}
__device__ __forceinline__ sdfk_iv iv_sub(sdfk_iv a, sdfk_iv b)
{
    const float inf = __builtin_inff();
    if ((a.hi == inf && b.hi == inf) || (a.lo == -inf && b.lo == -inf)) return iv_nan();   // inf - inf somewhere in the box
    sdfk_iv r; r.lo = a.lo - b.hi; r.hi = a.hi - b.lo; return iv_whole(r);
}
__device__ __forceinline__ sdfk_iv iv_mul(sdfk_iv a, sdfk_iv b)
{
    if ((iv_has_inf(a) && iv_has_zero(b)) || (iv_has_inf(b) && iv_has_zero(a))) return iv_nan();   // 0 * inf somewhere in the box
    const float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    sdfk_iv r;
    r.lo = sdfk_min_ieee(sdfk_min_ieee(p0, p1), sdfk_min_ieee(p2, p3));
    r.hi = sdfk_max_ieee(sdfk_max_ieee(p0, p1), sdfk_max_ieee(p2, p3));
    return iv_whole(r);
}
// a * a of ONE value (x * x in every Length): the product form would treat the two factors as independent and give [-|lo hi|, ..]
// for an interval around zero -- a negative lower bound under the square root, i.e. "unknown" for every block that a coordinate
// plane of some primitive's centre passes through.  fl(x * x) is monotone in |x|.
__device__ __forceinline__ sdfk_iv iv_sqr(sdfk_iv a)
{
    const float p0 = a.lo * a.lo, p1 = a.hi * a.hi;
    sdfk_iv r;
    r.hi = sdfk_max_ieee(p0, p1);
    r.lo = (a.lo >= 0.0f || a.hi <= 0.0f) ? sdfk_min_ieee(p0, p1) : 0.0f;
    return iv_whole(r);
}
__device__ __forceinline__ sdfk_iv iv_div(sdfk_iv a, sdfk_iv b)
{
    if (!(b.lo > 0.0f) && !(b.hi < 0.0f)) return iv_nan();   // the divisor may be zero (or is unknown)
    if (iv_has_inf(a) && iv_has_inf(b)) return iv_nan();      // inf / inf somewhere in the box
    const float q0 = a.lo / b.lo, q1 = a.lo / b.hi, q2 = a.hi / b.lo, q3 = a.hi / b.hi;
    sdfk_iv r;
    r.lo = sdfk_min_ieee(sdfk_min_ieee(q0, q1), sdfk_min_ieee(q2, q3));
    r.hi = sdfk_max_ieee(sdfk_max_ieee(q0, q1), sdfk_max_ieee(q2, q3));
    return iv_whole(r);
}
__device__ __forceinline__ sdfk_iv iv_neg(sdfk_iv a) { sdfk_iv r; r.lo = -a.hi; r.hi = -a.lo; return iv_whole(r); }
__device__ __forceinline__ sdfk_iv iv_abs(sdfk_iv a)
{
    const float x = __builtin_fabsf(a.lo), y = __builtin_fabsf(a.hi);
    sdfk_iv r;
    r.hi = sdfk_max_ieee(x, y);
    r.lo = (a.lo >= 0.0f || a.hi <= 0.0f) ? sdfk_min_ieee(x, y) : 0.0f;   // zero inside: [0, max]
    return iv_whole(r);
}
// (an interval that reaches below zero holds points whose root is NaN: unknown as a whole, not [NaN, sqrt(hi)])
__device__ __forceinline__ sdfk_iv iv_sqrt(sdfk_iv a) { sdfk_iv r; r.lo = sdfk_sqrt(a.lo); r.hi = sdfk_sqrt(a.hi); return iv_whole(r); }
__device__ __forceinline__ sdfk_iv iv_floor(sdfk_iv a) { sdfk_iv r; r.lo = __builtin_floorf(a.lo); r.hi = __builtin_floorf(a.hi); return iv_whole(r); }
// (a < b) ? a : b, Math.Min: both are min(a, b) when no NaN is involved, and minimum() makes the result unknown when one is
__device__ __forceinline__ sdfk_iv iv_min(sdfk_iv a, sdfk_iv b) { sdfk_iv r; r.lo = sdfk_min_ieee(a.lo, b.lo); r.hi = sdfk_min_ieee(a.hi, b.hi); return iv_whole(r); }
__device__ __forceinline__ sdfk_iv iv_max(sdfk_iv a, sdfk_iv b) { sdfk_iv r; r.lo = sdfk_max_ieee(a.lo, b.lo); r.hi = sdfk_max_ieee(a.hi, b.hi); return iv_whole(r); }
// (a < b) ? c : d
__device__ __forceinline__ sdfk_iv iv_sel_lt(sdfk_iv a, sdfk_iv b, sdfk_iv c, sdfk_iv d)
{
    if (!iv_unknown(a) && !iv_unknown(b)) {   // (an unknown operand may be NaN at some points -- the comparison is false there, which is d -- and anything elsewhere)
        if (a.hi < b.lo) return c;            // a < b for every operand pair
        if (a.lo >= b.hi) return d;           // never
    }
    sdfk_iv r;                                // either
    r.lo = sdfk_min_ieee(c.lo, d.lo);
    r.hi = sdfk_max_ieee(c.hi, d.hi);
    return iv_whole(r);
}
)SRC";

static const char* const kSampleKernels = R"SRC(
// Voxels.SampleSdf (Voxels.cs:72-125): sample point of voxel (ix,iy,iz) is
//   p = (min + 0.5*D) + (float)i * D   per axis (Voxels.cs:81,104-106),
// value -> Values[ix,iy,iz] (z fastest), colour -> Colors[ix,iy,iz].
//
// Fused form: sampling AND the marching-cubes sign bits in one pass over the grid.
// A workgroup owns 8 consecutive x rows x 256 z of one y: 8 / RPW wavefronts, RPW rows each
// (default RPW = 2: 256 lanes).  A lane evaluates 4 consecutive z of a row and issues one 16-byte
// nontemporal store for it (1 KiB contiguous per wavefront instruction).  Few stores per lane
// and 4..8-wavefront workgroups are what reaches the plain-fill store rate of the MI355X
// (tools/ubench/ub_store.hip: one store per lane 80 us, two 83 us, sixteen 100+ us for 512 MiB);
// two rows per lane halve the per-lane address / coordinate arithmetic, which matters once the
// SDF itself costs some ALU (sphere: 112 us with RPW = 1, 98 us with RPW = 2).
// Sign bits (value > iso): each lane leaves a nibble per row in LDS; wavefront 0 turns the 8 rows'
// nibbles into 4 bytes per lane (bit r of byte k = row r at z + k) and stores 256 contiguous
// bytes of bits8[y][x/8][z].  k_bits_transpose (mc_kernels.hip) regroups those bytes into the
// X-packed words the marching-cubes classifier reads; the volume is never re-read densely.
typedef float sdfk_f4 __attribute__((ext_vector_type(4)));
// (rows of the volume are A.pitch8 floats long -- nz rounded up to a multiple of 4 -- so every 4-voxel group is 16-byte aligned)
// `base` is UNIFORM over the wavefront (a row / run start), `elem` this lane's first float in it.
// SDFK_SAMPLE_NT = 2: the store leaves as "sc1 nt" -- written through and dropped from the XCD's L2 instead of kept there.  Measured
// with tools/ubench/ub_store2.hip (512 MiB in this kernel's pattern, four buffers in rotation): nt 81.2-82.0 us, sc1 nt 78.5-79.2 us
// (plain 84.0, sc1 81.3, sc0 sc1 81.5; as a linear fill: plain 79, nt 80, every sc1 form 76.0-76.6).  The compiler has no spelling for
// it (a volatile store is "sc0 sc1" but flat and not nt; __builtin_nontemporal_store drops the volatile): inline assembly, in the
// scalar-base form so that the address costs no vector instructions.
template <bool COLOR = false>
__device__ __forceinline__ void sdfk_store4_nt(float* base, int elem, float a, float b, float c, float d)
{
    const sdfk_f4 t = {a, b, c, d};
#if SDFK_SAMPLE_NT == 3   // (values sc1 nt, colours nt)
    if (COLOR) { __builtin_nontemporal_store(t, reinterpret_cast<sdfk_f4*>(base + elem)); return; }
#endif
#if SDFK_SAMPLE_NT >= 2
    // (s_nop 1: a store of more than 64 bits needs two wait states before a vector instruction may overwrite its data registers on
    // gfx940+, and the compiler's hazard recogniser does not look inside an asm statement -- without it the next row's values landed
    // in this row's store)
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" : : "v"((unsigned)elem * 4u), "v"(t), "s"(base) : "memory");
#elif SDFK_SAMPLE_NT
    __builtin_nontemporal_store(t, reinterpret_cast<sdfk_f4*>(base + elem));
#else
    *reinterpret_cast<sdfk_f4*>(base + elem) = t;
#endif
}
#ifndef SDFK_SAMPLE_RPW
#define SDFK_SAMPLE_RPW 2   // x rows per wavefront (stores per lane); workgroup = 8 / RPW wavefronts
#endif
// Two shapes of the 256 voxels a wavefront covers per x row (MODE); P = A.pitch8 = the row pitch of the volume
// (nz rounded up to a multiple of 4):
//   SDFK_ROWS  nz % 256 == 0 (P = nz): z in [256 b, 256 b + 256) of ONE y (blockIdx = (b, y, x/8)); y is a scalar
//   SDFK_FLAT  any nz:        voxels [256 b, 256 b + 256) of the whole (y, z) PLANE of the x row (ny * P floats,
//              contiguous in memory; blockIdx = (b, 0, x/8)): a wavefront's 1 KiB store is line-aligned
//              whatever nz is (when ny P % 16 == 0), may span two y rows, and only the last chunk of the
//              plane has idle lanes.  With z tiles per row instead, rows of 500 voxels make every store
//              end in two partial lines (139 instead of 85 us at 500^3) and rows of 520 use a third tile
//              for 8 planes.  The up to 3 padding voxels of a row are stored like any other (nobody reads
//              them) and give 0 sign bits.  (Round 1 stored rows of nz % 4 != 0 unpadded, at 4-byte alignment:
//              130 instead of 80 us at 510^3.)
#define SDFK_ROWS 0
#define SDFK_FLAT 1
// STORE = false (sdfk_sample_signs[_flat], SDFK_OPT_ELIDE_VOLUME): the sign bytes ONLY -- no Values, no Colors.  A volume that
// sdfk_sample_march creates for itself is a temporary (Sdf.cs:59-63): with the sign bits from here, the corners of the active
// cells re-evaluated (sdfk_corners_eval) and the vertex colours re-evaluated (sdfk_vertex_colors) the meshing chain never reads
// it, so the 4 (16) bytes per voxel need not be written at all; the colour arithmetic is dead code the compiler removes.
// ClipToBounds is then a run-time flag (A.clip) instead of a second instantiation.
// COLS = false (sdfk_sample_bits_nc*): a colour program's VALUES and sign bytes only -- the first of two passes, the colours follow in
// sdfk_sample_colors below (its colour arithmetic is dead code here).  Why two passes can be faster than one: see sdfk_sample_colors.
template <bool CLIP, int MODE, bool STORE = true, bool COLS = true>
__device__ __forceinline__ void sdfk_sample_bits_body(const SampleArgs& A, const SdfkK& K)
{
    constexpr int RPW = SDFK_SAMPLE_RPW;
    __shared__ unsigned char nib[8][64];
#ifdef SDFK_SAMPLE_VREG
    // (occupancy cap of the sampling kernels: naming vector register N as clobbered makes the kernel's allocation N + 1 registers,
    // i.e. at most 512 / (N + 1) wavefronts per SIMD -- 95: five, 127: four -- so that the marching-cubes kernels of the jobs on the
    // other lanes find wavefront slots and registers next to a running sampler; see DESIGN.md "places")
#define SDFK_S2_(x) #x
#define SDFK_S1_(x) SDFK_S2_(x)
    asm volatile("" ::: "v" SDFK_S1_(SDFK_SAMPLE_VREG));
#endif
#if SDFK_WRITES_COLOR
    __shared__ __attribute__((aligned(16))) float cbuf[STORE && COLS ? 8 / RPW : 1][STORE && COLS ? 768 : 1];   // colour staging, one slice per wavefront
#endif
    // (a program that only assigns .W has no staging buffer: 0.5 instead of 12.5 KB of LDS per workgroup -- at 8 wavefronts
    // per SIMD the sampler would otherwise hold 100 of a CU's 160 KB, and the marching-cubes kernels of the job before it
    // on the other stream, 28-41 KB per workgroup, could keep only one workgroup per CU next to it)
    // (the wavefront index as a scalar: row index, row base address and x coordinate stay off the VALU)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (grid dimensions y and z hold at most 65535 workgroups: the plane-chunk form, which has no y in its grid, takes the
    // groups of 8 x rows beyond that in blockIdx.y -- extreme aspect ratios only)
    const int x8 = MODE == SDFK_FLAT ? (int)(blockIdx.z + blockIdx.y * 65535u) : (int)blockIdx.z;
    if (x8 >= A.nx8) return;
    const int P = A.pitch8;                     // floats per row
    const int plane = A.ny * P;                 // floats per x row (< 2^31: Voxels.cs:82, + <= 3 per row)
    const int f0 = blockIdx.x * 256;            // SDFK_FLAT: first voxel of the chunk within the plane
    int iy = blockIdx.y, z = blockIdx.x * 256 + 4 * lane;
    bool zok = z < A.nz;
    if (MODE == SDFK_FLAT) {
        const int iy0 = f0 / P;                 // (scalar)
        iy = iy0;
        z = (f0 - iy0 * P) + 4 * lane;
        if (P >= 256) {                         // a chunk spans at most two rows
            if (z >= P) { z -= P; iy++; }
        } else {
            const int q = z / P;
            z -= q * P;
            iy += q;
        }
        zok = f0 + 4 * lane < plane;
    }
    const float py = A.my + (float)iy * A.dy;
    const bool edge_y = (iy == 0) | (iy == A.ny - 1);
    float pz[4];
    bool edge_z[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int zg = A.z0 + z + k;
        pz[k] = A.mz + (float)zg * A.dz;
        edge_z[k] = (zg == 0) | (zg == A.nz_global - 1);
    }
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        const int r = wave * RPW + rr;
        const int ix = x8 * 8 + r;
        unsigned n = 0;
        if (ix < A.nx && zok) {
            const float px = A.mx + (float)ix * A.dx;
            const bool edge_xy = edge_y | (ix == 0) | (ix == A.nx - 1);
            float w[4], cr[4], cg[4], cb[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                sdf_eval(K, px, py, pz[k], cr[k], cg[k], cb[k], w[k]);
                if ((CLIP || (!STORE && A.clip)) && (edge_xy || edge_z[k])) w[k] = A.outside;
            }
            // (the wavefront's run of the row starts at a uniform address: voxel f0 of the x row's plane, or z tile blockIdx.x of row (ix, y))
            const long o0 = MODE == SDFK_FLAT ? (long)ix * plane + f0 : ((long)ix * A.ny + (int)blockIdx.y) * P + (int)blockIdx.x * 256;
            if (STORE) sdfk_store4_nt(A.values + o0, 4 * lane, w[0], w[1], w[2], w[3]);
#if SDFK_WRITES_COLOR
            if (STORE && COLS && A.colors) {   // rgb of the lane's 4 voxels -> the wavefront's LDS slice (stored below)
                sdfk_f4* mine = reinterpret_cast<sdfk_f4*>(cbuf[wave] + 12 * lane);
                mine[0] = sdfk_f4{cr[0], cg[0], cb[0], cr[1]};
                mine[1] = sdfk_f4{cg[1], cb[1], cr[2], cg[2]};
                mine[2] = sdfk_f4{cb[2], cr[3], cg[3], cb[3]};
            }
#endif
            n = (w[0] > A.iso ? 1u : 0u) | (w[1] > A.iso ? 2u : 0u) | (w[2] > A.iso ? 4u : 0u) | (w[3] > A.iso ? 8u : 0u);
            if (MODE == SDFK_FLAT && z + 3 >= A.nz) n &= z < A.nz ? (1u << (A.nz - z)) - 1u : 0u;   // row padding: 0 bits
        }
        nib[r][lane] = (unsigned char)n;
        if (STORE && COLS && A.colors && ix < A.nx) {
            // A lane produced 48 contiguous bytes (4 voxels x rgb) of the wavefront's 3 KiB run.
            // Stored as they are, every instruction would touch a third of every line; through
            // the LDS slice they become three full 1 KiB nontemporal stores (lane L writes
            // floats 4 L .. 4 L + 3 of each KiB) -- ALL lanes store, also those whose own
            // voxels lie beyond nz: what they store belongs to the lanes before them.
            const int z0 = blockIdx.x * 256;
            const int left = MODE == SDFK_FLAT ? plane - f0 : P - z0;
            const int run = (left < 256 ? left : 256) * 3;   // floats of the run that exist
#if SDFK_WRITES_COLOR
            __builtin_amdgcn_wave_barrier();
            const float* cw = cbuf[wave];
            const sdfk_f4 t0 = *reinterpret_cast<const sdfk_f4*>(cw + 4 * lane);
            const sdfk_f4 t1 = *reinterpret_cast<const sdfk_f4*>(cw + 256 + 4 * lane);
            const sdfk_f4 t2 = *reinterpret_cast<const sdfk_f4*>(cw + 512 + 4 * lane);
            __builtin_amdgcn_wave_barrier();
#else
            const sdfk_f4 t0 = {0.0f, 0.0f, 0.0f, 0.0f}, t1 = t0, t2 = t0;   // a .W-only delegate leaves (0,0,0) everywhere (Voxels.cs:88-92)
#endif
            float* c = A.colors + (MODE == SDFK_FLAT ? (long)ix * plane + f0 : ((long)ix * A.ny + blockIdx.y) * P + z0) * 3;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const sdfk_f4 t = q == 0 ? t0 : (q == 1 ? t1 : t2);
                const int e = 256 * q + 4 * lane;   // first float of this lane's piece
                if (e + 3 < run) sdfk_store4_nt<true>(c, e, t.x, t.y, t.z, t.w);
                else {           // the run ends inside the piece
                    if (e < run) c[e] = t.x;
                    if (e + 1 < run) c[e + 1] = t.y;
                    if (e + 2 < run) c[e + 2] = t.z;
                }
            }
        }
    }
    __syncthreads();
    if (wave == 0 && zok) {
        unsigned out = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned q = nib[r][lane];
            out |= ((q & 1u) << r) | (((q >> 1) & 1u) << (8 + r)) | (((q >> 2) & 1u) << (16 + r)) | (((q >> 3) & 1u) << (24 + r));
        }
        *reinterpret_cast<unsigned*>(A.bits8 + ((long)iy * A.nx8 + x8) * P + z) = out;   // (the byte rows of bits8 have the same pitch)
    }
}
// SDFK_KERNELS: bit mask of the entry points this module contains (the host compiles a program's kernels on demand:
// bit 0 / 1 = the two instantiations without ClipToBounds, 3 / 4 = with it, 5 = sdfk_vertex_colors, 6 = sdfk_corners_eval, 7 = sdfk_raymarch,
// 2 / 8 = the sign-bits-only samplers of SDFK_OPT_ELIDE_VOLUME, 9 / 10 = its block culling: sdfk_cull_blocks / sdfk_eval_blocks, 11 = sdfk_eval_points,
// 12..15 = the four fused samplers without their colour half (sdfk_sample_bits_nc*), 16 = sdfk_sample_colors: two-pass sampling of colour volumes)
#ifndef SDFK_KERNELS
#define SDFK_KERNELS 0x1ffff
#endif
#if SDFK_KERNELS & 0x04
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_signs(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_ROWS, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x100
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_signs_flat(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_FLAT, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x01
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_ROWS>(A, K); }
#endif
#if SDFK_KERNELS & 0x02
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_flat(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_FLAT>(A, K); }
#endif
#if SDFK_KERNELS & 0x08
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_clip(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<true, SDFK_ROWS>(A, K); }
#endif
#if SDFK_KERNELS & 0x10
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_clip_flat(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<true, SDFK_FLAT>(A, K); }
#endif

// ---- two-pass sampling of a COLOUR volume (SDFK_OPT_COLOR_PASSES) -------------------------------------------------------------------
// One pass writes 16 B per voxel into TWO arrays (values, colours) from workgroups that own 8 x rows each: 16 address streams per
// workgroup, four 1-KiB stores per lane and row.  That pattern runs at 0.72-0.75 of the HBM peak WITHOUT any arithmetic, and no
// re-mapping inside one kernel gets past 0.81 (tools/ubench/ub_mixstore.hip, round 6: the tile 0.72-0.75, one mixed-address store per
// lane in a linear mapping 0.78-0.81, padded planes / longer runs: no gain) -- while the values alone go out at 0.81-0.82 in this
// very tile (the distance-only sampler) and the colour array alone, written as ONE linear stream with one full-KiB store per
// wavefront, at 0.85-0.86: 0.84-0.85 for the two launches together.  So a program whose arithmetic is cheap enough to be done twice
// samples in two passes: sdfk_sample_bits_nc* = the kernel above without its colour half, then sdfk_sample_colors = the colour array
// as a plain fill -- a workgroup evaluates 256 consecutive voxels of the volume (one per lane; the volume [nx][ny][pitch] is one
// contiguous run), leaves r, g, b in LDS, and three of its four wavefronts store one contiguous KiB each.  Same sdf_eval, same
// coordinates: bit-identical colours (tests/test_gpu_color_passes.py).  ClipToBounds touches values only (Voxels.cs:133-167).
#if SDFK_KERNELS & 0x1000
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_nc(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_ROWS, true, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x2000
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_nc_flat(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<false, SDFK_FLAT, true, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x4000
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_nc_clip(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<true, SDFK_ROWS, true, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x8000
extern "C" __global__ __launch_bounds__(512 / SDFK_SAMPLE_RPW) void sdfk_sample_bits_nc_clip_flat(SampleArgs A, SdfkK K) { sdfk_sample_bits_body<true, SDFK_FLAT, true, false>(A, K); }
#endif
#if SDFK_KERNELS & 0x10000
extern "C" __global__ __launch_bounds__(256) void sdfk_sample_colors(SampleArgs A, SdfkK K)
{
#if SDFK_WRITES_COLOR
    __shared__ __attribute__((aligned(16))) float cb[768];
    const unsigned tid = threadIdx.x;
    const unsigned P = (unsigned)A.pitch8, plane = (unsigned)A.ny * P;   // floats per row / per x row (the padded layout of the volume)
    const unsigned long long total = (unsigned long long)A.nx * plane;    // voxels of the padded volume (< 2^32)
    const unsigned long long chunk0 = (unsigned long long)blockIdx.x * 256ull;   // (uniform; at most 2^24 chunks: nx ny nz < 2^31, Voxels.cs:82)
    const unsigned long long me = chunk0 + tid;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (me < total) {
        // (the chunk's first voxel: uniform divisions; a lane then wraps at most once per level unless rows / planes are shorter than 256)
        unsigned ix = (unsigned)(chunk0 / plane);
        unsigned q = (unsigned)(chunk0 - (unsigned long long)ix * plane) + tid;
        if (q >= plane) { const unsigned k = q / plane; q -= k * plane; ix += k; }
        const unsigned iy = q / P, z = q - iy * P;
        const float px = A.mx + (float)(int)ix * A.dx, py = A.my + (float)(int)iy * A.dy, pz = A.mz + (float)(A.z0 + (int)z) * A.dz;
        float w;
        sdf_eval(K, px, py, pz, r, g, b, w);
    }
    cb[3 * tid] = r; cb[3 * tid + 1] = g; cb[3 * tid + 2] = b;
    __syncthreads();
    if (tid < 192) {   // three wavefronts, one contiguous KiB each (total and chunk0 are multiples of 4 voxels: no piece straddles the end)
        const unsigned long long e = chunk0 * 3ull + 4ull * tid;
        if (e + 3 < total * 3ull) {
            const sdfk_f4 t = *reinterpret_cast<const sdfk_f4*>(cb + 4 * tid);
            __builtin_nontemporal_store(t, reinterpret_cast<sdfk_f4*>(A.colors + e));
        }
    }
#endif
}
#endif

// ---- SDFK_OPT_ELIDE_VOLUME = 2, block culling: the sign bits without evaluating most voxels ---------------------------------
// The sign-only sampler above is pure arithmetic (74 us for the sphere at 512^3).  But a block of voxels whose values provably
// all lie on one side of the iso value needs no evaluation at all: sdf_interval() below is the program evaluated over a BOX of
// sample points in the interval form of every operation (prelude), and a box whose result interval excludes the iso value gets
// its sign bits written as constants.  No assumption about the SDF (no Lipschitz bound): the intervals contain every float
// the per-voxel evaluation can produce, NaN = unknown = evaluate.
// A block = 64 x (ONE word of the X-packed sign array the marching-cubes classifier reads) x 4 y x 4 z = 1024 voxels, 16 words.
// sdfk_cull_blocks: coarse boxes of 2 x 2 x 2 blocks first (one interval evaluation decides eight blocks), then eight LANES per block,
// one per 8-x sub-box = one byte of the block's word: the words go out with the bytes of the decided sub-boxes, and a block with
// undecided ones is appended to a work list with their mask (order irrelevant: every word has its place).  sdfk_eval_blocks: one
// WAVEFRONT per listed block evaluates its undecided sub-boxes voxel by voxel -- the ballot of "value > iso" holds a sub-box's bytes of
// eight sign words (all eight undecided: lane = x, the ballot IS the word).  Both write bits[z][y][xw] directly: this path has no byte
// form and no k_bits_transpose.
// The work list is kept in SDFK_CULL_LISTS sub-lists, each with a counter on a cache line of its own (workgroup w of the culling kernel
// appends to sub-list w % SDFK_CULL_LISTS, region `region` words long): every append is one atomic, same-address atomics serialise at
// ~10 ns each, and a thousand workgroups appending to ONE counter were 10 of the kernel's 16 us at 512^3.
#define SDFK_CULL_LISTS 64
struct CullArgs { unsigned long long* bits; unsigned* worklist; unsigned* counter; int nbx, nby, nbz; int cpw; unsigned region; };
#if SDFK_KERNELS & 0x600
__device__ __forceinline__ float sdfk_coord(float m, int i, float d) { return m + (float)i * d; }
#endif
#if SDFK_KERNELS & 0x200
// EIGHT lanes per block, one per 8-x sub-box -- one BYTE of the block's sign word each: a box of 64 x is too wide for scenes that repeat
// along x (the interval of mod(x, period) covers the whole period).  (Round 4 gave a block ONE lane that walked the eight sub-boxes
// in a row: 131 072 lanes at 512^3, two wavefronts per SIMD, each a chain of eight interval evaluations -- 19 us of latency for 3 us
// of vector work.)  The eight lanes combine their bytes with three butterfly steps; each then stores two of the block's sixteen rows.
//
// COARSE boxes first (round 5): a wavefront owns C.cpw (1..32) consecutive coarse boxes of 2 x 2 x 2 blocks (128 x 8 x 8 voxels).  Lane l
// evaluates the program over the whole of coarse box l -- ONE interval evaluation per wavefront for all of them -- and a box whose
// interval excludes the iso value gives its eight blocks their constant words without the 64 sub-box evaluations (sound for the
// same reason: the interval contains every float a voxel of the box can produce; an unknown or straddling interval decides nothing).
// Only the coarse boxes the surface may pass through take the per-sub-box pass: a fifth of them for the 512^3 sphere (3.9x fewer
// interval evaluations), all of them for a scene that repeats along x with a period below 128 voxels (1 / cpw more).  C.cpw < 0: the
// coarse pass is skipped (A/B: SDFK_CULL_COARSE=0), -C.cpw boxes per wavefront.
extern "C" __global__ __launch_bounds__(256) void sdfk_cull_blocks(SampleArgs A, CullArgs C, SdfkK K)
{
    __shared__ unsigned s_list[4][32];       // the blocks a wavefront lists (at most 8 per coarse box, cpw <= 4) ...
    __shared__ unsigned char s_msk[4][32];   // ... and which of their eight sub-boxes are undecided
    __shared__ unsigned s_cnt[4], s_base;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cpw = C.cpw < 0 ? -C.cpw : C.cpw;   // (<= 4: s_list)
    const int ncx = (C.nbx + 1) >> 1, ncy = (C.nby + 1) >> 1, ncz = (C.nbz + 1) >> 1;
    const int ncoarse = ncx * ncy * ncz;
    const int c0 = (blockIdx.x * 4 + wave) * cpw;   // this wavefront's first coarse box
    // phase 1: lane l decides coarse box c0 + l as a whole.  -1: no such box; 0 / 1: all its voxels <= iso / > iso; 2: look closer;
    // + 4: every one of its eight blocks exists, is whole and away from the clipped faces (the loop below then needs no tests)
    int ccls = -1, pbx = 0, pby = 0, pbz = 0;
    if (lane < cpw && c0 + lane < ncoarse) {
        const int c = c0 + lane;
        pbx = c % ncx;
        const int t = c / ncx;
        pby = t % ncy;
        pbz = t / ncy;
        const int x0 = pbx * 128, y0 = pby * 8, z0 = pbz * 8;
        const bool whole = x0 + 128 <= A.nx && y0 + 8 <= A.ny && z0 + 8 <= A.nz;
        const bool faces = A.clip && (x0 == 0 || x0 + 128 >= A.nx || y0 == 0 || y0 + 8 >= A.ny || A.z0 + z0 == 0 || A.z0 + z0 + 8 >= A.nz_global);
        ccls = 2;
        if (C.cpw > 0) {
            const int x1 = x0 + 127 < A.nx ? x0 + 127 : A.nx - 1, y1 = y0 + 7 < A.ny ? y0 + 7 : A.ny - 1, z1 = z0 + 7 < A.nz ? z0 + 7 : A.nz - 1;
            const sdfk_iv X = iv_make(sdfk_coord(A.mx, x0, A.dx), sdfk_coord(A.mx, x1, A.dx));
            const sdfk_iv Y = iv_make(sdfk_coord(A.my, y0, A.dy), sdfk_coord(A.my, y1, A.dy));
            const sdfk_iv Z = iv_make(sdfk_coord(A.mz, A.z0 + z0, A.dz), sdfk_coord(A.mz, A.z0 + z1, A.dz));
            const sdfk_iv W = sdf_interval(K, X, Y, Z);
            if (W.lo > A.iso && W.hi >= W.lo) ccls = 1;
            else if (W.hi <= A.iso && W.lo <= W.hi) ccls = 0;
        }
        if (whole && !faces) ccls += 4;
    }
    // phase 2: coarse box by coarse box, its eight blocks side by side (block j = lane / 8 at (j & 1, j / 2 & 1, j / 4), sub-box lane & 7;
    // the lane stores rows 2 sx and 2 sx + 1 of its block's sixteen (zz, yy) rows: their place relative to the coarse box's first word)
    const int j = lane >> 3, sx = lane & 7;
    const int bi = j & 1, bj = (j >> 1) & 1, bk = j >> 2;
    long off[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int row = 2 * sx + q, zz = row >> 2, yy = row & 3;
        off[q] = ((long)(4 * bk + zz) * A.ny + (4 * bj + yy)) * C.nbx + bi;
    }
    unsigned nlisted = 0;
    for (int k = 0; k < cpw; k++) {
        const int cc = __builtin_amdgcn_readlane(ccls, k);   // (all uniform over the wavefront)
        if (cc < 0) break;
        const int cbx = __builtin_amdgcn_readlane(pbx, k), cby = __builtin_amdgcn_readlane(pby, k), cbz = __builtin_amdgcn_readlane(pbz, k);
        unsigned long long* base = C.bits + ((long)(8 * cbz) * A.ny + 8 * cby) * C.nbx + 2 * cbx;
        if (cc == 4 || cc == 5) {   // decided as a whole, nothing to test: 128 constant words
            const unsigned long long word = cc == 5 ? ~0ull : 0ull;
            base[off[0]] = word;
            base[off[1]] = word;
            continue;
        }
        const int bx = 2 * cbx + bi, by = 2 * cby + bj, bz = 2 * cbz + bk;
        int cls = -1;   // -1: not a block; 0 / 1: every voxel of this lane's sub-box is <= iso / > iso; 2: evaluate the block
        if (bx < C.nbx && by < C.nby && bz < C.nbz) {
            const int x0 = bx * 64, y0 = by * 4, z0 = bz * 4;
            cls = 2;
            // only whole blocks away from the clipped faces are candidates (the others are O(n^2) few)
            const bool whole = x0 + 64 <= A.nx && y0 + 4 <= A.ny && z0 + 4 <= A.nz;
            const bool faces = A.clip && (x0 == 0 || x0 + 64 >= A.nx || y0 == 0 || y0 + 4 >= A.ny || A.z0 + z0 == 0 || A.z0 + z0 + 4 >= A.nz_global);
            if (whole && !faces) {
                if ((cc & 3) != 2) cls = cc & 3;   // decided with its coarse box
                else {
                    const sdfk_iv X = iv_make(sdfk_coord(A.mx, x0 + 8 * sx, A.dx), sdfk_coord(A.mx, x0 + 8 * sx + 7, A.dx));
                    const sdfk_iv Y = iv_make(sdfk_coord(A.my, y0, A.dy), sdfk_coord(A.my, y0 + 3, A.dy));
                    const sdfk_iv Z = iv_make(sdfk_coord(A.mz, A.z0 + z0, A.dz), sdfk_coord(A.mz, A.z0 + z0 + 3, A.dz));
                    const sdfk_iv W = sdf_interval(K, X, Y, Z);
                    // (decided only by an interval that is known as a whole: both comparisons are false for NaN)
                    if (W.lo > A.iso && W.hi >= W.lo) cls = 1;
                    else if (W.hi <= A.iso && W.lo <= W.hi) cls = 0;
                }
            }
        }
        // the block's word (byte sx = 0xff where the sub-box lies above the iso value) and "some sub-box is undecided", over the 8 lanes of the block
        unsigned lo = (cls == 1 && sx < 4) ? 0xffu << (8 * sx) : 0u, hi = (cls == 1 && sx >= 4) ? 0xffu << (8 * (sx - 4)) : 0u;
        unsigned und = cls == 2 ? 1u << sx : 0u;   // (the block's undecided sub-boxes, a bit each)
        // (ds_swizzle in bit mode: and 0x1f, or 0, xor 1 / 2 / 4 -- lanes 8 k .. 8 k + 7 exchange among themselves)
#define SDFK_OR8(v)                                                              \
    v |= (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (1 << 10) | 0x1f);        \
    v |= (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (2 << 10) | 0x1f);        \
    v |= (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1f);
        SDFK_OR8(lo) SDFK_OR8(hi) SDFK_OR8(und)
#undef SDFK_OR8
        // the word goes out whether or not the block is decided: the bytes of its undecided sub-boxes (0 here) are what
        // sdfk_eval_blocks fills in afterwards (rows of a block that sticks out of the grid do not exist)
        if (cls >= 0) {
            const unsigned long long word = (unsigned long long)lo | ((unsigned long long)hi << 32);
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int row = 2 * sx + q;
                if (by * 4 + (row & 3) < A.ny && bz * 4 + (row >> 2) < A.nz) base[off[q]] = word;
            }
        }
        // a block with undecided sub-boxes is listed with their mask: for the 512^3 sphere 9.4 % of the blocks, but only 1.8 % of the sub-boxes
        const bool listed = cls >= 0 && und && sx == 0;
        const unsigned long long need = __builtin_amdgcn_ballot_w64(listed);
        if (listed) {
            const unsigned at = nlisted + (unsigned)__builtin_popcountll(need & ((1ull << lane) - 1ull));
            s_list[wave][at] = (unsigned)((bz * C.nby + by) * C.nbx + bx);
            s_msk[wave][at] = (unsigned char)und;
        }
        nlisted += (unsigned)__builtin_popcountll(need);
    }
    // the undecided blocks of this WORKGROUP: ONE atomic for all of them, to one of SDFK_CULL_LISTS counters (one atomic per wavefront
    // of 8 blocks to one counter made this kernel 73 us at 512^3)
    if (lane == 0) s_cnt[wave] = nlisted;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const unsigned n = s_cnt[w];
        before += w < wave ? n : 0u;
        total += n;
    }
    const unsigned sub = blockIdx.x % SDFK_CULL_LISTS;
    if (threadIdx.x == 0 && total) s_base = atomicAdd(C.counter + 32u * sub, total);
    __syncthreads();
    if ((unsigned)lane < nlisted) {
        const size_t at = (size_t)sub * C.region + s_base + before + (unsigned)lane;
        C.worklist[at] = s_list[wave][lane];
        reinterpret_cast<unsigned char*>(C.worklist + (size_t)SDFK_CULL_LISTS * C.region)[at] = s_msk[wave][lane];   // (the masks: behind the 64 regions)
    }
}
#endif
#if SDFK_KERNELS & 0x400
extern "C" __global__ __launch_bounds__(256) void sdfk_eval_blocks(SampleArgs A, CullArgs C, SdfkK K)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the sub-lists' lengths, lane s that of sub-list s, and their running sum: item e of the whole is item e - (sum before s) of
    // sub-list s = the number of running sums <= e
    static_assert(SDFK_CULL_LISTS == 64, "one lane per sub-list");
    const unsigned mine = C.counter[32 * lane];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned n = (unsigned)__shfl_up((int)incl, o);
        if (lane >= o) incl += n;
    }
    const unsigned count = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    // one wavefront per listed block.  All eight sub-boxes undecided: lane = x, sixteen evaluations, the 64-bit ballot of "value > iso" IS
    // the sign word of a (y, z) row.  Otherwise sub-box by sub-box (8 x 4 x 4 voxels, lane = (x, yy, upper / lower z pair), two evaluations):
    // the ballot holds one BYTE per (yy, z) row -- the sub-box's byte of that row's word (sdfk_cull_blocks wrote the decided ones).
    const int xs = lane & 7, yy8 = (lane >> 3) & 3, zh = lane >> 5;
    unsigned char* const bytes = reinterpret_cast<unsigned char*>(C.bits);
    const unsigned char* const masks = reinterpret_cast<const unsigned char*>(C.worklist + (size_t)SDFK_CULL_LISTS * C.region);
    for (unsigned e = blockIdx.x * 4u + (unsigned)wave; e < count; e += gridDim.x * 4u) {
        const int sub = __builtin_popcountll(__builtin_amdgcn_ballot_w64(incl <= e));
        const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)(incl - mine), sub);
        const size_t at = (size_t)sub * C.region + (e - first);
        const int b = (int)C.worklist[at];
        unsigned mask = masks[at];
        const int bx = b % C.nbx, t = b / C.nbx, by = t % C.nby, bz = t / C.nby;
        if (mask == 0xffu) {
            const int ix = bx * 64 + lane;
            const float px = sdfk_coord(A.mx, ix, A.dx);
            const bool edge_x = (ix == 0) | (ix == A.nx - 1);
            unsigned long long word = 0;   // (this lane's word of the block: lane 4 zz + yy keeps row (yy, zz))
#pragma unroll
            for (int zz = 0; zz < 4; zz++) {
                const int iz = bz * 4 + zz, zg = A.z0 + iz;
                const float pz = sdfk_coord(A.mz, zg, A.dz);
                const bool edge_z = (zg == 0) | (zg == A.nz_global - 1);
#pragma unroll
                for (int yy = 0; yy < 4; yy++) {
                    const int iy = by * 4 + yy;
                    float r, g, bl, w;
                    sdf_eval(K, px, sdfk_coord(A.my, iy, A.dy), pz, r, g, bl, w);
                    if (A.clip && (edge_x || edge_z || iy == 0 || iy == A.ny - 1)) w = A.outside;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(ix < A.nx && w > A.iso);   // bits of x >= nx stay 0
                    if (lane == 4 * zz + yy) word = m;
                }
            }
            if (lane < 16) {
                const int yy = lane & 3, zz = lane >> 2;
                if (by * 4 + yy < A.ny && bz * 4 + zz < A.nz)
                    C.bits[((long)(bz * 4 + zz) * A.ny + (by * 4 + yy)) * C.nbx + bx] = word;
            }
            continue;
        }
        const int iy = by * 4 + yy8;
        const float py = sdfk_coord(A.my, iy, A.dy);
        while (mask) {
            const int sx = __builtin_ctz(mask);
            mask &= mask - 1u;
            const int ix = bx * 64 + sx * 8 + xs;
            const float px = sdfk_coord(A.mx, ix, A.dx);
            const bool edge_xy = (ix == 0) | (ix == A.nx - 1) | (iy == 0) | (iy == A.ny - 1);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int iz = bz * 4 + 2 * i + zh, zg = A.z0 + iz;
                float r, g, bl, w;
                sdf_eval(K, px, py, sdfk_coord(A.mz, zg, A.dz), r, g, bl, w);
                if (A.clip && (edge_xy || zg == 0 || zg == A.nz_global - 1)) w = A.outside;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(ix < A.nx && w > A.iso);   // bits of x >= nx stay 0
                if (lane < 8) {   // row (yy = lane & 3, z = 2 i + lane / 4) of the sub-box
                    const int ry = by * 4 + (lane & 3), rz = bz * 4 + 2 * i + (lane >> 2);
                    if (ry < A.ny && rz < A.nz)
                        bytes[(((long)rz * A.ny + ry) * C.nbx + bx) * 8 + sx] = (unsigned char)(m >> (8 * lane));
                }
            }
        }
    }
}
#endif

// Corner values of the active cells, RE-EVALUATED instead of gathered: for a volume this very
// program has just sampled, the 8 corners of cell (x,y,z) are 8 more evaluations of the same
// float32 expression (same point arithmetic, same clip rule), bit-identical to what the
// sampling kernel stored -- and 8 evaluations per ACTIVE cell (a surface, not a volume) are
// far cheaper than 4 scattered 8-byte loads per cell from a [x][y][z] grid.
// Corner order v0..v7 = (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1) (Cell.cs:24-31).
__device__ __forceinline__ float sdfk_voxel(const SampleArgs& A, const SdfkK& K, int ix, int iy, int iz)
{
    const float px = A.mx + (float)ix * A.dx;
    const float py = A.my + (float)iy * A.dy;
    const int zg = A.z0 + iz;
    const float pz = A.mz + (float)zg * A.dz;
    float r, g, b, w;
    sdf_eval(K, px, py, pz, r, g, b, w);
    if (A.clip && ((ix == 0) | (ix == A.nx - 1) | (iy == 0) | (iy == A.ny - 1) | (zg == 0) | (zg == A.nz_global - 1)))
        w = A.outside;
    return w;
}

#if SDFK_KERNELS & 0x40
extern "C" __global__ __launch_bounds__(256) void sdfk_corners_eval(SampleArgs A, const unsigned* __restrict__ rec_xy,
                                                                     const unsigned* __restrict__ rec_z,
                                                                     float* __restrict__ rec_corners,
                                                                     const unsigned* __restrict__ n_active, unsigned cap, int xbits, SdfkK K)
{
    const unsigned n = *n_active < cap ? *n_active : cap;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned xy = rec_xy[i];
        const int x = (int)(xy & ((1u << xbits) - 1u)), y = (int)(xy >> xbits), z = (int)rec_z[i];
        const float c0 = sdfk_voxel(A, K, x, y, z), c1 = sdfk_voxel(A, K, x + 1, y, z);
        const float c2 = sdfk_voxel(A, K, x + 1, y + 1, z), c3 = sdfk_voxel(A, K, x, y + 1, z);
        const float c4 = sdfk_voxel(A, K, x, y, z + 1), c5 = sdfk_voxel(A, K, x + 1, y, z + 1);
        const float c6 = sdfk_voxel(A, K, x + 1, y + 1, z + 1), c7 = sdfk_voxel(A, K, x, y + 1, z + 1);
        float4* o = reinterpret_cast<float4*>(rec_corners + (size_t)i * 8);
        o[0] = make_float4(c0, c1, c2, c3);
        o[1] = make_float4(c4, c5, c6, c7);
    }
}
#endif

// Vertex colours, RE-EVALUATED instead of gathered: k_vertices (mc_kernels.hip) left (creator record, edge) for every
// emitted vertex; the colour of a vertex on edge (c1, c2) of its creator cell is the blend of the two corner colours
// weighted by 1 / (eps + |value - iso|) (Cell.AddFaceFromEdgeIndex, Cell.cs:314-350), that of a centre vertex the same
// over all 8 corners (Cell.CalculateCenterVertex, Cell.cs:501-549) -- and for a volume this program has just sampled the
// corner colours AND values are two (eight) more evaluations of the same float32 expression, bit-identical to what the
// sampling kernel stored, instead of two 12-byte gathers 1 MiB apart from a 1.6 GB colour volume (k_vertices of the
// README scene at 512^3: 121 us with the gathers, of which ~60 us are the gathers).  Arithmetic and order exactly as in
// k_vertices' gather path: float products and sum, then ONE double division.
#if (SDFK_KERNELS & 0x20) && SDFK_WRITES_COLOR
struct VColArgs { const unsigned* vdesc; const unsigned* rec_xy; const unsigned* rec_z; const unsigned* counters; float* colors;
                  unsigned cap_vertices; int xbits; float iso; };
__device__ __forceinline__ void sdfk_voxel_rgbw(const SampleArgs& A, const SdfkK& K, int ix, int iy, int iz, float& r, float& g, float& b, float& w)
{
    const float px = A.mx + (float)ix * A.dx;
    const float py = A.my + (float)iy * A.dy;
    const int zg = A.z0 + iz;
    const float pz = A.mz + (float)zg * A.dz;
    sdf_eval(K, px, py, pz, r, g, b, w);
    if (A.clip && ((ix == 0) | (ix == A.nx - 1) | (iy == 0) | (iy == A.ny - 1) | (zg == 0) | (zg == A.nz_global - 1)))
        w = A.outside;
}
extern "C" __global__ __launch_bounds__(256) void sdfk_vertex_colors(SampleArgs A, VColArgs V, SdfkK K)
{
    // McCounters (mc_params.h): total_v = word 3, nghost = word 5; published by k_vertices (stream order)
    unsigned nv = V.counters[3] - V.counters[5];
    if (nv > V.cap_vertices) nv = V.cap_vertices;
    const double iso = (double)V.iso;
    for (unsigned o = blockIdx.x * 256u + threadIdx.x; o < nv; o += gridDim.x * 256u) {
        const unsigned rec = V.vdesc[2u * o], e = V.vdesc[2u * o + 1u];
        const unsigned xy = V.rec_xy[rec];
        const int x = (int)(xy & ((1u << V.xbits) - 1u)), y = (int)(xy >> V.xbits), z = (int)V.rec_z[rec];
        float out[3];
        if (e == 12u) {
            double ff = 0.0;
            float fc[3] = {0.0f, 0.0f, 0.0f};
            for (int k = 0; k < 8; k++) {
                float c[3], w;
                sdfk_voxel_rgbw(A, K, x + (((k + 1) >> 1) & 1), y + ((k >> 1) & 1), z + (k >> 2), c[0], c[1], c[2], w);
                const double wk = 1.0 / (0.0000001 + __builtin_fabs((double)w - iso));
                ff += wk;
                const float wf = (float)wk;
                if (k == 0) { fc[0] = c[0] * wf; fc[1] = c[1] * wf; fc[2] = c[2] * wf; }
                else { fc[0] = fc[0] + c[0] * wf; fc[1] = fc[1] + c[1] * wf; fc[2] = fc[2] + c[2] * wf; }
            }
            for (int j = 0; j < 3; j++) out[j] = (float)((double)fc[j] / ff);
        } else {
            // the two end corners of cube edge e (Luts.cs:30-52)
            const int c1 = e < 8u ? (int)e : (int)e - 8, c2 = e < 8u ? (int)((e & 4u) | ((e + 1u) & 3u)) : (int)e - 4;
            float ca[3], cb[3], wa, wb;
            sdfk_voxel_rgbw(A, K, x + (((c1 + 1) >> 1) & 1), y + ((c1 >> 1) & 1), z + (c1 >> 2), ca[0], ca[1], ca[2], wa);
            sdfk_voxel_rgbw(A, K, x + (((c2 + 1) >> 1) & 1), y + ((c2 >> 1) & 1), z + (c2 >> 2), cb[0], cb[1], cb[2], wb);
            const double w1 = 1.0 / (0.0000001 + __builtin_fabs((double)wa - iso));
            const double w2 = 1.0 / (0.0000001 + __builtin_fabs((double)wb - iso));
            const double ff = w1 + w2;
            const float w1f = (float)w1, w2f = (float)w2;
            for (int j = 0; j < 3; j++) {
                const float cj = ca[j] * w1f + cb[j] * w2f;
                out[j] = (float)((double)cj / ff);
            }
        }
        float* c = V.colors + 3ul * o;
        c[0] = out[0]; c[1] = out[1]; c[2] = out[2];
    }
}
#endif

// SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points, one lane per point.  The delegate writes into the caller's Vector4 buffer:
// a delegate that only assigns .W (Sdfs.Sphere, Sdf.cs:211) leaves X, Y, Z of every element as the caller had them -- here too.
struct PointArgs { const float* points; float* rgbw; long n; };
#if SDFK_KERNELS & 0x800
extern "C" __global__ __launch_bounds__(256) void sdfk_eval_points(PointArgs A, SdfkK K)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    float r, g, b, w;
    sdf_eval(K, A.points[3 * i], A.points[3 * i + 1], A.points[3 * i + 2], r, g, b, w);
#if SDFK_WRITES_COLOR
    A.rgbw[4 * i] = r; A.rgbw[4 * i + 1] = g; A.rgbw[4 * i + 2] = b;
#endif
    A.rgbw[4 * i + 3] = w;
}
#endif

// RayMarcher.RenderDepth / Render (RayMarcher.cs:45-95, 134-169): sphere tracing, one lane per
// pixel.  The reference runs every step as a whole-image batch operation (VectorData.cs); per
// pixel that is the same float32 arithmetic, reproduced op for op (mul then add, never fused;
// Vector3.Normalize = v / length for the rays, NormalizeInplace = v * (1 / length) for the
// shading vectors).  The camera position and the inverse view-projection are derived by the
// caller from ViewTransform (GetCameraRays, RayMarcher.cs:97-112).
struct RayArgs { float* depth; float* rgb; float cam[3]; float m[16]; int width, height; float nearp, farp; int iters; };

__device__ __forceinline__ float sdfk_scene_w(const SdfkK& K, float x, float y, float z, float& r, float& g, float& b)
{
    float w;
    sdf_eval(K, x, y, z, r, g, b, w);
    return w;
}

__device__ __forceinline__ void sdfk_normalize_inplace(float& x, float& y, float& z)   // VectorData.cs:490-508
{
    const float len = __builtin_sqrtf((x * x + y * y) + z * z);
    if (len > 0.0f) {
        const float r = 1.0f / len;
        x = x * r; y = y * r; z = z * r;
    }
}

#if SDFK_KERNELS & 0x80
extern "C" __global__ __launch_bounds__(256) void sdfk_raymarch(RayArgs A, SdfkK K)
{
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= (long)A.width * A.height) return;
    const int j = (int)(k / A.width), i = (int)(k - (long)j * A.width);
    // GetCameraRays, RayMarcher.cs:113-128
    const float y = 1.0f - 2.0f * (float)j / (float)(A.height - 1);
    const float x = -1.0f + 2.0f * (float)i / (float)(A.width - 1);
    float v4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) v4[q] = ((x * A.m[q] + y * A.m[4 + q]) + 0.0f * A.m[8 + q]) + 1.0f * A.m[12 + q];
    const float dx = v4[0] / v4[3] - A.cam[0], dy = v4[1] / v4[3] - A.cam[1], dz = v4[2] / v4[3] - A.cam[2];
    const float dl = __builtin_sqrtf((dx * dx + dy * dy) + dz * dz);
    const float rx = dx / dl, ry = dy / dl, rz = dz / dl;
    float depth = A.nearp - 0.1f;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    for (int it = 0; it < A.iters; it++) {
        const float w = sdfk_scene_w(K, rx * depth + A.cam[0], ry * depth + A.cam[1], rz * depth + A.cam[2], cr, cg, cb);
        depth = depth + w;
    }
    if (A.depth) A.depth[k] = depth;
    if (!A.rgb) return;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f;
    if (A.iters > 0) { d0 = 0.0f + cr; d1 = 0.0f + cg; d2 = 0.0f + cb; }
    const float sx = A.cam[0] + rx * depth, sy = A.cam[1] + ry * depth, sz = A.cam[2] + rz * depth;
    const float go = 1e-5f;
    float t0, t1, t2;
    const float wpx = sdfk_scene_w(K, sx + go * 1.0f, sy + go * 0.0f, sz + go * 0.0f, t0, t1, t2);
    const float wpy = sdfk_scene_w(K, sx + go * 0.0f, sy + go * 1.0f, sz + go * 0.0f, t0, t1, t2);
    const float wpz = sdfk_scene_w(K, sx + go * 0.0f, sy + go * 0.0f, sz + go * 1.0f, t0, t1, t2);
    const float wnx = sdfk_scene_w(K, sx + -go * 1.0f, sy + -go * 0.0f, sz + -go * 0.0f, t0, t1, t2);
    const float wny = sdfk_scene_w(K, sx + -go * 0.0f, sy + -go * 1.0f, sz + -go * 0.0f, t0, t1, t2);
    const float wnz = sdfk_scene_w(K, sx + -go * 0.0f, sy + -go * 0.0f, sz + -go * 1.0f, t0, t1, t2);
    float nx = wpx - wnx, ny = wpy - wny, nz = wpz - wnz;
    sdfk_normalize_inplace(nx, ny, nz);
    float lx = 5.0f - sx, ly = 5.0f - sy, lz = 10.0f - sz;
    sdfk_normalize_inplace(lx, ly, lz);
    const float dv = sdfk_max_ieee((nx * lx + ny * ly) + nz * lz, 0.0f);
    const float bgm = depth > A.farp ? 1.0f : 0.0f;
    const float fgm = bgm == 0.0f ? 1.0f : 0.0f;
    float* o = A.rgb + 3 * k;
    o[0] = 0.0f + ((dv * d0 + 0.1f) * fgm + bgm * 0.5f);
    o[1] = 0.0f + ((dv * d1 + 0.1f) * fgm + bgm * 0.75f);
    o[2] = 0.0f + ((dv * d2 + 0.1f) * fgm + bgm * 1.0f);
}
#endif

)SRC";

// Constants of a program are kernel ARGUMENTS, not literals: the generated source -- hence the hiprtc module and the
// on-disk cache entry -- depends on the program's STRUCTURE (opcodes, operand ids, outputs) only, and a program with the same
// structure and other constants (another radius, another period: Sdf.cs:202-214 is a closure, a new radius costs the
// reference nothing) reuses the loaded module.  `params` receives the constants in the order of their K.k[] slots.
// Baked as literals (part of the structure): a constant one of whose uses lets the compiler simplify EXACTLY -- x * +-1,
// x / +-1, x / 2^k (an exact reciprocal), x + -0, x - +0, -0 - x -- so that the common cases cost what they cost as
// literals; with -ffp-contract=off and no fast-math every such folding is IEEE-exact, results are bit-identical either way.
// More than kMaxParams constants: ALL literals, as in rounds 1-3.  Arguments live in scalar registers, every evaluation of the
// program (8 per lane in the sampler) uses all of them, and gfx950 has ~100 SGPRs per wavefront: measured with hipcc on unions of
// k translated primitives, 29 constants compile without a spill, 36 spill 16 SGPRs into VGPR lanes, 58 (BASELINE config C4, the
// 8-primitive union) spill 78 -- and that sampler then takes 6.3 instead of 3.6 ms at 1024^3.  Scenes that big change their
// structure more often than their constants.
constexpr int kMaxParams = 28;

inline bool sdfk_const_is_baked(const sdfk_op* ops, int n_ops, int i)
{
    uint32_t bits;
    memcpy(&bits, &ops[i].imm, 4);
    const uint32_t mag = bits & 0x7fffffffu;
    const bool one = mag == 0x3f800000u, pzero = bits == 0u, nzero = bits == 0x80000000u;
    const uint32_t ex = mag >> 23;
    const bool pow2 = (mag & 0x007fffffu) == 0u && ex >= 2u && ex <= 252u;   // +-2^k whose reciprocal is a normal float too
    if (!(one || pzero || nzero || pow2)) return false;
    for (int u = i + 1; u < n_ops; u++) {
        const sdfk_op& q = ops[u];
        switch (q.opcode) {
        case SDFK_OP_MUL: if (one && (q.a == i || q.b == i)) return true; break;
        case SDFK_OP_DIV: if ((one || pow2) && q.b == i) return true; break;
        case SDFK_OP_ADD: if (nzero && (q.a == i || q.b == i)) return true; break;
        case SDFK_OP_SUB: if ((pzero && q.b == i) || (nzero && q.a == i)) return true; break;
        default: break;
        }
    }
    return false;
}

inline bool generate_sample_source(const sdfk_op* ops, int n_ops, const int32_t out_rgbw[4], int writes_color,
                                   std::string& src, std::string& err, std::vector<float>* params = nullptr)
{
    char buf[256];
    if (n_ops > (1 << 20)) { err = "program too long"; return false; }
    std::string body, ibody;   // the per-point program and its interval form (sdf_interval: block culling of SDFK_OPT_ELIDE_VOLUME)
    body.reserve((size_t)n_ops * 48);
    ibody.reserve((size_t)n_ops * 56);
    int n_params = 0;
    bool parameterise = true;
    {
        int n_const = 0;
        for (int i = 0; i < n_ops; i++) n_const += ops[i].opcode == SDFK_OP_CONST ? 1 : 0;
        if (n_const > kMaxParams) parameterise = false;
    }
    if (params) params->clear();
    for (int i = 0; i < n_ops; i++) {
        const sdfk_op& o = ops[i];
        auto arg = [&](int32_t id, const char* which) -> bool {
            if (id < 0 || id >= i) {
                snprintf(buf, sizeof buf, "op %d: operand %s = %d does not name an earlier value", i, which, id);
                err = buf;
                return false;
            }
            return true;
        };
        int arity = 0;
        const char* fmt = nullptr;
        switch (o.opcode) {
        case SDFK_OP_CONST: {
            uint32_t bits;
            memcpy(&bits, &o.imm, 4);
            char ib[96];
            if (parameterise && !sdfk_const_is_baked(ops, n_ops, i)) {
                snprintf(ib, sizeof ib, "    const sdfk_iv i%d = iv_const(K.k[%d]);\n", i, n_params);
                snprintf(buf, sizeof buf, "    const float v%d = K.k[%d];\n", i, n_params++);
                if (params) params->push_back(o.imm);
            } else {
                snprintf(ib, sizeof ib, "    const sdfk_iv i%d = iv_const(__uint_as_float(0x%08xu));\n", i, bits);
                snprintf(buf, sizeof buf, "    const float v%d = __uint_as_float(0x%08xu);\n", i, bits);
            }
            body += buf;
            ibody += ib;
            continue;
        }
        case SDFK_OP_X: snprintf(buf, sizeof buf, "    const float v%d = X;\n", i); body += buf; snprintf(buf, sizeof buf, "    const sdfk_iv i%d = X;\n", i); ibody += buf; continue;
        case SDFK_OP_Y: snprintf(buf, sizeof buf, "    const float v%d = Y;\n", i); body += buf; snprintf(buf, sizeof buf, "    const sdfk_iv i%d = Y;\n", i); ibody += buf; continue;
        case SDFK_OP_Z: snprintf(buf, sizeof buf, "    const float v%d = Z;\n", i); body += buf; snprintf(buf, sizeof buf, "    const sdfk_iv i%d = Z;\n", i); ibody += buf; continue;
        case SDFK_OP_ADD: arity = 2; fmt = "v%d + v%d"; break;
        case SDFK_OP_SUB: arity = 2; fmt = "v%d - v%d"; break;
        case SDFK_OP_MUL: arity = 2; fmt = "v%d * v%d"; break;
        case SDFK_OP_DIV: arity = 2; fmt = "v%d / v%d"; break;
        case SDFK_OP_NEG: arity = 1; fmt = "-v%d"; break;
        case SDFK_OP_ABS: arity = 1; fmt = "__builtin_fabsf(v%d)"; break;
        case SDFK_OP_SQRT: arity = 1; fmt = "sdfk_sqrt(v%d)"; break;
        case SDFK_OP_FLOOR: arity = 1; fmt = "__builtin_floorf(v%d)"; break;
        case SDFK_OP_MIN_SEL: arity = 3; fmt = "(v%d < v%d) ? v%d : v%d"; break;
        case SDFK_OP_MAX_SEL: arity = 3; fmt = "(v%d > v%d) ? v%d : v%d"; break;
        case SDFK_OP_MIN_IEEE: arity = 2; fmt = "sdfk_min_ieee(v%d, v%d)"; break;
        case SDFK_OP_MAX_IEEE: arity = 2; fmt = "sdfk_max_ieee(v%d, v%d)"; break;
        case SDFK_OP_SEL_LT: arity = 4; fmt = "(v%d < v%d) ? v%d : v%d"; break;
        default:
            snprintf(buf, sizeof buf, "op %d: unknown opcode %d", i, o.opcode);
            err = buf;
            return false;
        }
        if (arity >= 1 && !arg(o.a, "a")) return false;
        if (arity >= 2 && !arg(o.b, "b")) return false;
        if (arity >= 4 && (!arg(o.c, "c") || !arg(o.d, "d"))) return false;
        char ex[160];
        if (arity == 1) snprintf(ex, sizeof ex, fmt, o.a);
        else if (arity == 2) snprintf(ex, sizeof ex, fmt, o.a, o.b);
        else if (arity == 3) snprintf(ex, sizeof ex, fmt, o.a, o.b, o.a, o.b);  // compare-select on (a,b)
        else snprintf(ex, sizeof ex, fmt, o.a, o.b, o.c, o.d);
        snprintf(buf, sizeof buf, "    const float v%d = %s;\n", i, ex);
        body += buf;
        {   // the same operation on intervals
            static const char* const ifn[] = {nullptr, nullptr, nullptr, nullptr, "iv_add", "iv_sub", "iv_mul", "iv_div", "iv_neg", "iv_abs", "iv_sqrt",
                                              "iv_floor", "iv_min", "iv_max", "iv_min", "iv_max", "iv_sel_lt"};
            if (o.opcode == SDFK_OP_MUL && o.a == o.b) snprintf(buf, sizeof buf, "    const sdfk_iv i%d = iv_sqr(i%d);\n", i, o.a);
            else if (arity == 1) snprintf(buf, sizeof buf, "    const sdfk_iv i%d = %s(i%d);\n", i, ifn[o.opcode], o.a);
            else if (arity <= 3) snprintf(buf, sizeof buf, "    const sdfk_iv i%d = %s(i%d, i%d);\n", i, ifn[o.opcode], o.a, o.b);
            else snprintf(buf, sizeof buf, "    const sdfk_iv i%d = %s(i%d, i%d, i%d, i%d);\n", i, ifn[o.opcode], o.a, o.b, o.c, o.d);
            ibody += buf;
        }
    }
    for (int k = 0; k < 4; k++) {
        if (k < 3 && !writes_color) continue;
        if (out_rgbw[k] < 0 || out_rgbw[k] >= n_ops) { err = "output id out of range"; return false; }
    }
    src.clear();
    src += kSamplePrelude;
    snprintf(buf, sizeof buf, "struct SdfkK { float k[%d]; };\n", n_params > 0 ? n_params : 1);
    src += buf;
    if (params && params->empty()) params->push_back(0.0f);   // (the argument always exists: one unused slot)
    src += "__device__ __forceinline__ void sdf_eval(const SdfkK& K, float X, float Y, float Z, float& R, float& G, float& B, float& W)\n{\n";
    src += body;
    if (writes_color) {
        snprintf(buf, sizeof buf, "    R = v%d; G = v%d; B = v%d;\n", out_rgbw[0], out_rgbw[1], out_rgbw[2]);
        src += buf;
    } else {
        // delegates that only assign .W leave colour at the zeroed scratch value (Voxels.cs:88-92)
        src += "    R = 0.0f; G = 0.0f; B = 0.0f;\n";
    }
    snprintf(buf, sizeof buf, "    W = v%d;\n}\n", out_rgbw[3]);
    src += buf;
    // the distance over a box of sample points, in intervals (only compiled into the block-culling kernels)
    src += "#if SDFK_KERNELS & 0x200\n__device__ __forceinline__ sdfk_iv sdf_interval(const SdfkK& K, sdfk_iv X, sdfk_iv Y, sdfk_iv Z)\n{\n";
    src += ibody;
    snprintf(buf, sizeof buf, "    return i%d;\n}\n#endif\n", out_rgbw[3]);
    src += buf;
    src += writes_color ? "#define SDFK_WRITES_COLOR 1\n" : "#define SDFK_WRITES_COLOR 0\n";
    src += kSampleKernels;
    return true;
}

}  // namespace sdfk
