// mc_kernels.hip -- marching-cubes pipeline for MI355X (gfx950, wave64).
//
// The reference (MarchingCubes.cs:39-92 + Cell.cs) is one serial z->y->x sweep whose
// output order (vertex numbering, triangle order, float32 normal accumulation order)
// depends on that sweep.  This pipeline reproduces the same output in parallel:
//
//   K1  signbits  volume[x][y][z] (z fastest) -> 1 bit per voxel packed along X:
//                 bits[z][y][xw], bit b = (value(64*xw+b, y, z) > iso), via the byte form
//                 bits8[y][x/8][z] (k_signbits8 -> k_bits_transpose).  The only dense pass over
//                 the volume (4 B/voxel read); skipped when the fused sampling kernel
//                 (sample_codegen.h, sdfk_sample_bits) already produced the bytes.
//   K2  compact   workgroups own chunks of 64-cell X-runs in sweep order; "8 corners not all
//                 equal" is 64-bit-parallel on the sign words.  Count pass -> one-workgroup
//                 scan of the per-chunk counts -> write pass (wave-shuffle prefix inside the
//                 workgroup): the active-cell list comes out in EXACT serial-sweep order
//                 without sorting and without atomics.
//   K3  resolve   one lane per active cell: gathers the 8 corners into its LDS column, runs
//                 the 33-case dispatcher, decides which vertices the cell CREATES in the sweep
//                 (it is the first live cell of the sweep touching that grid edge); per-256-cell
//                 totals of (created vertices, triangles) are scanned by one workgroup, so
//                 every cell's first vertex id / first triangle = chunk prefix + in-chunk prefix.
//   K4  vertices  one lane per created vertex: position/colour in the creator's frame, normal
//                 as a gather over the <=4 cells around the edge in sweep order (bit-exact
//                 float32 accumulation order, no atomics).
//   K5  triangles one lane per triangle index: the vertex id of an edge is its creator cell's
//                 first vertex id + the edge's rank in the creator's list.
//
// After K3 the volume is not touched again: records carry the cells' corner values, and
// neighbouring / creator cells are found through a per-row start table + a short search in
// the sorted record list, so K4/K5 work on compact, cache-resident arrays only (no per-voxel
// maps).  Lookup tables are copied from __constant__ to LDS per workgroup.  No MFMA: nothing here
// is a contraction.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>
#include "mc_device.h"
#include "mc_kernels.h"

namespace sdfk {

// k_vertices stages, per chunk of MC_CHUNK records, two windows of the record list (K4_WMAX slots of 40 B together) and the
// rowstart[] slices that map a row to its records (2 x K4_RMAX entries).  Whatever falls outside is still found -- through
// global memory, at several dependent round trips per lookup -- so the sizes decide how often that happens
// (tools/experiments/r04_vertices_counters.patch counts it):
//   rows: a chunk that straddles two layers spans the EMPTY rows between the last surface row of one layer and the first of the
//         next, ncy - (rows the surface occupies) of them.  With 288 entries 7 % of the 512^3 sphere's vertices (the second
//         layer's part of every such chunk) took the global path; with ncy + 100 or more none do.  1152 covers ncy <= ~1050.
//   slots: rows near a horizontal tangent hold long runs of records, so the next layer's rows can hold far more than the
//         chunk's own (README scene: 1.4 % of the vertices outside 640 slots, 0.5 % outside 752).
// Round 5: 576 slots and 1024 row starts (16-bit, relative to their window) = 40.6 KB of LDS, and the register allocation held to 128
// VGPRs (one spill): FOUR workgroups per CU instead of three -- the kernel is bound by what is resident.  Same box, us: 752 slots /
// 1152 rows / 3 per CU: sphere 46.1, README scene 83.8, 8-primitive union at 1024^3 261; 576 / 384 / 4 per CU: 42.0 / 83.4 / 246 (576 /
// 384 at 3 per CU: 46.3 / 92.8 -- the smaller windows alone cost the README scene 9 us, the fourth workgroup gives them back).
#ifndef SDFK_K4_WMAX
#define SDFK_K4_WMAX 576
#endif
#ifndef SDFK_K4_RMAX
#define SDFK_K4_RMAX 1024
#endif
constexpr int K4_WMAX = SDFK_K4_WMAX;   // k_vertices: record-window slots staged in LDS (both windows together)
constexpr int K4_RMAX = SDFK_K4_RMAX;   // k_vertices: rowstart entries staged per window
static_assert(K4_WMAX <= 768 && K4_WMAX > 512, "k_vertices stages the windows in three rounds of 256 slots");
// k_triangles stages the chunk's block of vertex ids (rec_vid, MC_VSTRIDE slots at most) in LDS: a smooth surface uses 3.5 slots per
// record (840 per chunk), 1536 cover 6.4 per record; what a noisier chunk has beyond that is read from global memory (19.4 KB of
// LDS in all: eight workgroups per CU)
#ifndef SDFK_K5_VMAX
#define SDFK_K5_VMAX 1536
#endif
constexpr int K5_VMAX = SDFK_K5_VMAX;
static_assert(K5_VMAX % 256 == 0 && K5_VMAX >= 256, "k_triangles loads the block in rounds of 256 slots");

// ---------------------------------------------------------------------------
// K1: sign bits
// ---------------------------------------------------------------------------
// For volumes that did not come from the fused sampling kernel (uploaded arrays, another iso value, sub-sampled
// steps): the same shape as that kernel with loads in place of stores.  Rows of the volume are `pitch` floats long
// (nz rounded up to a multiple of 4: every row starts 16-byte aligned; the byte rows of bits8 have the same pitch).
// A workgroup owns 8 consecutive x rows x 256 z of one y (4 wavefronts, 2 rows each); a lane loads 4 consecutive z of
// a row with one 16-byte load (1 KiB contiguous per wavefront instruction), leaves a sign nibble in LDS; wavefront 0
// turns the 8 rows' nibbles into 4 bytes per lane and stores 256 contiguous bytes of bits8[y][x/8][z].
// k_bits_transpose then regroups the bytes into the X-packed words.
//
// FLAT (nz not a multiple of 256; blockIdx = (chunk, 0, x/8)): a wavefront covers voxels [256 b, 256 b + 256) of the
// contiguous (y, z) plane of an x row (ny * pitch floats) instead of a z tile of one row, exactly like the sampling
// kernel (sample_codegen.h): no idle z tile, line-aligned loads.  Voxels of the row padding give 0 bits.
template <bool FLAT>
__global__ __launch_bounds__(256) void k_signbits8(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny,
                                                   int nz, int nx8, int pitch, float iso)
{
    __shared__ unsigned char nib[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x8 = FLAT ? (int)(blockIdx.z + blockIdx.y * 65535u) : (int)blockIdx.z;   // (FLAT: x groups beyond 65535 in blockIdx.y)
    if (x8 >= nx8) return;
    int iy = blockIdx.y, z = blockIdx.x * 256 + 4 * lane;
    bool zok = z < pitch;
    if (FLAT) {
        const int f0 = blockIdx.x * 256, iy0 = f0 / pitch;
        iy = iy0;
        z = (f0 - iy0 * pitch) + 4 * lane;
        const int q = z / pitch;
        z -= q * pitch;
        iy += q;
        zok = f0 + 4 * lane < ny * pitch;
    }
    float v[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int ix = x8 * 8 + wave * 2 + rr;
        v[rr][0] = v[rr][1] = v[rr][2] = v[rr][3] = -INFINITY;   // (never > iso: voxels beyond the row end give 0 bits)
        if (ix < nx && zok) {
            const float4 q = *reinterpret_cast<const float4*>(values + ((size_t)ix * ny + iy) * pitch + z);
            v[rr][0] = q.x;
            if (z + 1 < nz) v[rr][1] = q.y;
            if (z + 2 < nz) v[rr][2] = q.z;
            if (z + 3 < nz) v[rr][3] = q.w;
            if (z >= nz) v[rr][0] = -INFINITY;
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const unsigned n = (v[rr][0] > iso ? 1u : 0u) | (v[rr][1] > iso ? 2u : 0u) | (v[rr][2] > iso ? 4u : 0u) | (v[rr][3] > iso ? 8u : 0u);
        nib[wave * 2 + rr][lane] = (unsigned char)n;
    }
    __syncthreads();
    if (wave == 0 && zok) {
        unsigned out = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned q = nib[r][lane];
            out |= ((q & 1u) << r) | (((q >> 1) & 1u) << (8 + r)) | (((q >> 2) & 1u) << (16 + r)) | (((q >> 3) & 1u) << (24 + r));
        }
        *reinterpret_cast<unsigned*>(bits8 + ((size_t)iy * nx8 + x8) * pitch + z) = out;   // pitch % 4 == 0
    }
}

// bits8[y][x/8][z] (one byte = the sign bits of 8 consecutive x, written by the fused sampling
// kernel) -> bits[z][y][xw]: word xw is the 8 bytes x/8 = 8*xw .. 8*xw+7 of the same (y, z).
// Workgroup = (128 z, one y, 8 words): 64 byte-rows x 128 B go through LDS, every z then
// leaves as 64 contiguous bytes.
__global__ __launch_bounds__(256) void k_bits_transpose(const uint8_t* __restrict__ bits8, uint64_t* __restrict__ bits,
                                                        int nx8, int ny, int nz, int nxw, int pitch8)
{
    __shared__ __attribute__((aligned(4))) uint8_t t[64][132];
    // blockIdx.x = (y, z tile) -- ny may exceed what grid dimension y holds --, blockIdx.z (+ 65535 blockIdx.y) = group of 8 words
    const int ztiles = (nz + 127) / 128;
    const int iy = (int)(blockIdx.x / (unsigned)ztiles), z0 = (int)(blockIdx.x % (unsigned)ztiles) * 128;
    const int xw0 = (int)(blockIdx.z + blockIdx.y * 65535u) * 8, row0 = xw0 * 8;
    if (xw0 >= nxw) return;
    // (all eight loads of a lane are issued before the first LDS store: as a "load; store" loop the compiler keeps one load
    // in flight per trip -- eight dependent L2 round trips, 8 us whatever the size of the volume; a slab of 68 planes took as
    // long as 512 planes)
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = (int)threadIdx.x + 256 * i;
        const int row = k >> 5, c = (k & 31) * 4;
        v[i] = 0;   // (byte rows are pitch8 long, a multiple of 4: a 4-byte group never straddles a row)
        if (row0 + row < nx8 && z0 + c < nz)
            v[i] = *reinterpret_cast<const unsigned*>(bits8 + ((size_t)iy * nx8 + row0 + row) * pitch8 + z0 + c);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = (int)threadIdx.x + 256 * i;
        *reinterpret_cast<unsigned*>(&t[k >> 5][(k & 31) * 4]) = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = threadIdx.x; k < 128 * 8; k += 256) {
        const int zz = k >> 3, xw = k & 7;
        if (z0 + zz < nz && xw0 + xw < nxw) {
            uint64_t w = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) w |= (uint64_t)t[xw * 8 + b][zz] << (8 * b);
            bits[((size_t)(z0 + zz) * ny + iy) * nxw + xw0 + xw] = w;
        }
    }
}

// ---------------------------------------------------------------------------
// workgroup / grid scan helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// exclusive prefix of v over the 256 lanes of the workgroup, *total = workgroup sum; `red` is
// summed over the workgroup on the side (same barriers) and returned through the reference
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t& red, uint32_t* total)
{
    __shared__ uint32_t s_incl[4], s_red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    const uint32_t r = wave_sum_u32(red);
    __syncthreads();
    if (lane == 63) { s_incl[wave] = incl; s_red[wave] = r; }
    __syncthreads();
    uint32_t pre = incl - v;
    for (int w = 0; w < wave; w++) pre += s_incl[w];
    *total = s_incl[0] + s_incl[1] + s_incl[2] + s_incl[3];
    red = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    return pre;
}

// exclusive prefix of v over the 256 lanes of the workgroup; *total = workgroup sum
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t* s_wave /*[4]*/, uint64_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t incl = wave_incl_scan_u64(v);
    __syncthreads();   // s_wave may still be read from a previous call
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t pre = incl - v;
    for (int w = 0; w < wave; w++) pre += s_wave[w];
    *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return pre;
}

// Which chunk a workgroup starts with.  Workgroups are dealt round-robin over the eight XCDs (b and b + 8 share an L2), and a chunk's
// second window in k_vertices is the first window of a chunk a few chunks further on (the next layer's cells): with G > 0 the q-th
// workgroup of an XCD takes chunk ((q / G) * 8 + xcd) * G + q % G -- runs of G consecutive chunks per XCD -- so that the second
// fetch of a window hits that XCD's L2.  G = 0: chunk b.  Measured at 512^3 (round 5, profiles/r05_ab_chunk_xcd_groups.txt):
// k_vertices fetches 57.3 MB with G = 0, 51.3 (8), 41.6 (16), 36.7 (32), 34.3 (64) and takes 47.1 / 46.3 / 45.8-46.0 / 46.9 / 47.4 us
// (long runs leave the XCDs unevenly loaded); k_resolve with the same dealing 15.8 -> 15.0 us; k_triangles reads nothing twice
// and does not care.  (Round 2 had tried one contiguous EIGHTH of the chunks per XCD: 18 % slower.)
#ifndef SDFK_KV_XCD_GROUP
#define SDFK_KV_XCD_GROUP 16
#endif
#ifndef SDFK_KR_XCD_GROUP
#define SDFK_KR_XCD_GROUP 16
#endif
#ifndef SDFK_KT_XCD_GROUP
#define SDFK_KT_XCD_GROUP 0
#endif
template <uint32_t G>
__device__ __forceinline__ uint32_t first_chunk_of_block()
{
    const uint32_t b = blockIdx.x;
    if (G == 0) return b;
    const uint32_t span = 8u * (G ? G : 1u), whole = (gridDim.x / span) * span;
    if (b >= whole) return b;
    const uint32_t xcd = b & 7u, q = b >> 3;
    return ((q / (G ? G : 1u)) * 8u + xcd) * G + q % (G ? G : 1u);
}

// The compaction's logical blocks (layer, part) for the same dealing: a run of G workgroups of one XCD takes G consecutive LAYERS of one
// part -- sign plane z + 1 of a layer is plane z of the next, so the run fetches G + 1 planes instead of 2 G (with block = workgroup
// index the two fetches of a plane come from different XCDs: 34 MB through the fabric for a 16.8 MB array).  Layers beyond the last
// whole group of G keep the plain order.  Returns layer * bpl + part.
#ifndef SDFK_K2_XCD_GROUP
#define SDFK_K2_XCD_GROUP 16
#endif
__device__ __forceinline__ int compact_block_of_workgroup(int nlay, int bpl)
{
    constexpr int G = SDFK_K2_XCD_GROUP;
    if (G == 0) return (int)blockIdx.x;
    const int v = (int)first_chunk_of_block<(uint32_t)G>();
    const int GG = G ? G : 1, grouped = (nlay / GG) * GG * bpl;
    if (v >= grouped) return v;
    const int run = v / GG, k = v - run * GG;
    const int lg = run / bpl, part = run - lg * bpl;
    return (lg * GG + k) * bpl + part;
}

// ---------------------------------------------------------------------------
// K2: ordered compaction of active cells
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t shr1_in(uint64_t w, uint64_t next) { return (w >> 1) | (next << 63); }

typedef unsigned long long u64x2u __attribute__((ext_vector_type(2), aligned(8)));

// five consecutive sign words (two 16-byte loads + one 8-byte load; the buffer is padded)
__device__ __forceinline__ void load_words5(const uint64_t* p, uint64_t* w)
{
    const u64x2u lo = *reinterpret_cast<const u64x2u*>(p);
    const u64x2u hi = *reinterpret_cast<const u64x2u*>(p + 2);
    w[0] = lo.x; w[1] = lo.y; w[2] = hi.x; w[3] = hi.y; w[4] = p[4];
}

// activity of the 64 cells x = 64*xw .. 64*xw+63 of one cell row: a/b/c/d = sign words of the
// voxel rows (y,z) (y+1,z) (y,z+1) (y+1,z+1), *n = the following word of the same row.
// `rem` = number of cells of this word that exist (x < ncx); the word after the last one of a
// row may hold anything: it only reaches bit 63, a cell that never exists.
// Corners: v0=a v1=as v2=bs v3=b v4=c v5=cs v6=ds v7=d.  A cell is active when some corner
// differs from v0; its sign word is 0xA5 or 0x5A (case 13) when v1,v3,v4,v6 differ from v0 and
// v2,v5,v7 equal it.
__device__ __forceinline__ uint64_t segment_mask(uint64_t a, uint64_t an, uint64_t b, uint64_t bn, uint64_t c, uint64_t cn,
                                                 uint64_t d, uint64_t dn, int rem, uint64_t& m13)
{
    const uint64_t as = shr1_in(a, an), bs = shr1_in(b, bn), cs = shr1_in(c, cn), ds = shr1_in(d, dn);
    const uint64_t valid = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    const uint64_t x1 = a ^ as, x3 = a ^ b, x4 = a ^ c, x6 = a ^ ds;
    const uint64_t eq = (a ^ bs) | (a ^ cs) | (a ^ d);
    m13 = (x1 & x3 & x4 & x6) & ~eq & valid;
    return (x1 | x3 | x4 | x6 | eq) & valid;
}

// A sign plane is a flat array of ny*nxw words and cell row y uses voxel rows y and y+1, so
// segment i = y*nxw + xw (64 cells) reads words i, i+1, i+nxw, i+nxw+1 of planes z and z+1:
// sweep order == i order.  Logical block b = (layer, 1024 consecutive segments); each lane owns
// 4 consecutive segments and keeps their words in registers (20 words, five 16/8-byte loads per
// voxel row).  WRITE = false: count the active cells of the block; WRITE = true: write them at
// (sum of the earlier blocks) + in-block prefix.
// A WORKGROUP takes the same 1024 segments of K2_LPB consecutive layers -- K2_LPB logical blocks -- and keeps all K2_LPB + 1
// sign planes it needs in registers: plane z + 1 of one layer is plane z of the next, so a plane is fetched (K2_LPB + 1) / K2_LPB
// times per pass instead of twice (with one workgroup per logical block the second fetch comes from another XCD's workgroup
// and misses its L2: 34 MB through the fabric for a 16.8 MB array).  All loads of a lane are issued up front.
// K2_LPB = 1 is the product: measured at 512^3 (round 5, profiles/r05_ab_compact_layers.txt), both passes together take 25.9 us
// with 1, 30.1 with 2 and 41.3 with 4 layers per workgroup -- a quarter of the workgroups, each lane four layers in a row with a
// workgroup scan per layer -- and the pipelined step does not move (0.1399 / 0.1391 / 0.1449 ms): the fabric serves the second
// fetch at no visible cost, the longer workgroups cost what they cost.

// Totals of the compaction and the layer marks, from the count pass's blockcnt[] (one workgroup: the write pass's first block, or
// k_blockscan).
// (out of line, the few fields it needs by value: inlined into the write pass its registers cost every workgroup of that kernel a
// wavefront of occupancy per SIMD -- 102 instead of 94 VGPRs --, and a reference to the kernel-argument block would force a scratch
// copy of it at the call)
__device__ __noinline__ void publish_compaction_impl(const uint64_t* blockcnt, McCounters* counters, uint32_t* rowstart_end, int gb, int ge, int nlog)
{
    uint64_t ghost = 0, upto_emit_end = 0, all13 = 0, all = 0;
    const int nt = (int)blockDim.x;
    for (int i0 = threadIdx.x; i0 < nlog; i0 += 4 * nt) {   // (four independent loads per trip)
        uint64_t wv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wv[k] = i0 + k * nt < nlog ? blockcnt[i0 + k * nt] : 0ull;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * nt;
            const uint64_t c = wv[k] & 0xffffffffull;
            all += c;
            if (i < gb) ghost += c;
            if (i < ge) upto_emit_end += c;
            all13 += wv[k] >> 32;
        }
    }
    __shared__ uint64_t s_sum[4][16];
    ghost = wave_sum_u64(ghost); upto_emit_end = wave_sum_u64(upto_emit_end); all13 = wave_sum_u64(all13); all = wave_sum_u64(all);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
    if (lane == 0) { s_sum[0][wave] = ghost; s_sum[1][wave] = upto_emit_end; s_sum[2][wave] = all13; s_sum[3][wave] = all; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t[4] = {0, 0, 0, 0};
        for (int w = 0; w < nw; w++)
            for (int q = 0; q < 4; q++) t[q] += s_sum[q][w];
        const uint32_t nall = (uint32_t)t[3];
        counters->n_active = nall;
        counters->n_ghost_cells = (uint32_t)t[0];
        counters->n_emit_cells = (uint32_t)(t[1] - t[0]);
        counters->n_case13 = (uint32_t)t[2];
        *rowstart_end = nall;   // sentinel after the last layer's rows
    }
}

__device__ __forceinline__ void publish_compaction(const McParams& P, int nlay, int nlog)
{
    publish_compaction_impl(P.blockcnt, P.counters, P.rowstart + (size_t)nlay * P.ncy, (P.lay_emit_begin - P.lay_count_begin) * P.bpl,
                            (P.lay_emit_end - P.lay_count_begin) * P.bpl, nlog);
}

template <bool WRITE>
__global__ __launch_bounds__(256) void k_compact(McParams P)
{
    const int nlay = P.lay_list_end - P.lay_count_begin;
    const int nlog = nlay * P.bpl;                           // logical blocks: [layer][part]
    const int wg = K2_LPB == 1 ? compact_block_of_workgroup(nlay, P.bpl) : (int)blockIdx.x;   // (the workgroup's place in (layer group, part) order)
    const int part = wg % P.bpl, lay0 = (wg / P.bpl) * K2_LPB;
    const int nl = min(K2_LPB, nlay - lay0);                 // layers of this workgroup
    const int b0 = lay0 * P.bpl + part;                      // logical block of its first layer; layer l: b0 + l * bpl
    const int nseg = P.ncy * P.nxw;
    const int i0 = part * 1024 + 4 * (int)threadIdx.x;
    const int wave = (int)(threadIdx.x >> 6);
    // the count pass recorded which wavefronts found nothing in which layer: those skip the sign words here
    bool look[K2_LPB];
#pragma unroll
    for (int l = 0; l < K2_LPB; l++) look[l] = l < nl && (!WRITE || P.wavecnt[(b0 + l * P.bpl) * 4 + wave] != 0);
    // write pass: this lane's share of the counts of all blocks before b0, loaded HERE (four independent loads per trip, in
    // flight together with the sign words below) -- a plain "load; add" loop after the words cost up to eight dependent L2
    // round trips for the last blocks of a 512^3 grid --, and of the bpl blocks between two consecutive layers of this workgroup
    uint32_t before[K2_LPB];
#pragma unroll
    for (int l = 0; l < K2_LPB; l++) before[l] = 0;
    if (WRITE) {
        for (int i = threadIdx.x; i < b0; i += 1024) {
            const uint32_t c0 = (uint32_t)P.blockcnt[i];
            const uint32_t c1 = i + 256 < b0 ? (uint32_t)P.blockcnt[i + 256] : 0u;
            const uint32_t c2 = i + 512 < b0 ? (uint32_t)P.blockcnt[i + 512] : 0u;
            const uint32_t c3 = i + 768 < b0 ? (uint32_t)P.blockcnt[i + 768] : 0u;
            before[0] += (c0 + c1) + (c2 + c3);
        }
#pragma unroll
        for (int l = 1; l < K2_LPB; l++)
            if (l < nl)
                for (int i = threadIdx.x; i < P.bpl; i += 256) before[l] += (uint32_t)P.blockcnt[b0 + (l - 1) * P.bpl + i];
    }
    int y0 = 0, xw0 = 0;
    if (i0 < nseg) {
        y0 = i0 / P.nxw;
        xw0 = i0 - y0 * P.nxw;
    }
    // sign planes lay0 .. lay0 + nl of this lane's two voxel rows: plane p serves layers p - 1 and p
    uint64_t w[K2_LPB + 1][2][5];
    {
        const size_t plane = (size_t)P.ny * P.nxw;
        const uint64_t* f0 = P.bits + (size_t)(P.lay_count_begin + lay0) * plane + i0;
#pragma unroll
        for (int p = 0; p <= K2_LPB; p++) {
            const bool need = i0 < nseg && ((p > 0 && look[p - 1]) || (p < K2_LPB && look[p]));
            if (need) {
                load_words5(f0 + (size_t)p * plane, w[p][0]);
                load_words5(f0 + (size_t)p * plane + P.nxw, w[p][1]);
            } else {
#pragma unroll
                for (int k = 0; k < 5; k++) w[p][0][k] = w[p][1][k] = 0;
            }
        }
    }
    uint32_t cnts[K2_LPB], n13s[K2_LPB];
    uint32_t base_acc = 0;   // (write pass) records before the current layer's block
#pragma unroll
    for (int l = 0; l < K2_LPB; l++) {
        cnts[l] = n13s[l] = 0;
        if (l >= nl) continue;   // (uniform over the workgroup)
        uint64_t m[4] = {0, 0, 0, 0};
        uint32_t cnt = 0, n13 = 0;
        if (look[l] && i0 < nseg) {
            int xk = xw0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int rem = (i0 + k < nseg) ? P.ncx - xk * 64 : 0;
                uint64_t m13;
                m[k] = segment_mask(w[l][0][k], w[l][0][k + 1], w[l][1][k], w[l][1][k + 1], w[l + 1][0][k], w[l + 1][0][k + 1],
                                    w[l + 1][1][k], w[l + 1][1][k + 1], rem, m13);
                cnt += (uint32_t)__popcll(m[k]);
                n13 += (uint32_t)__popcll(m13);
                if (++xk == P.nxw) xk = 0;
            }
        }
        if (!WRITE) {
            cnts[l] = cnt; n13s[l] = n13;
            // (the masks of a wavefront that found something, for the write pass: its lane t holds segments 4 t .. 4 t + 3 of the block)
            if (P.segmask && __builtin_amdgcn_ballot_w64(cnt != 0u) != 0ull) {
                uint64_t* q = P.segmask + ((size_t)(b0 + l * P.bpl) * 1024u + 4u * threadIdx.x);
                *reinterpret_cast<u64x2u*>(q) = u64x2u{m[0], m[1]};
                *reinterpret_cast<u64x2u*>(q + 2) = u64x2u{m[2], m[3]};
            }
            continue;
        }
        // exclusive prefix of this block = sum of the counts of all earlier blocks (a few thousand
        // words, read cooperatively: cheaper than a separate scan launch)
        uint32_t total, red = before[l];
        const uint32_t pre = block_excl_scan_u32(cnt, red, &total);   // also reduces `red` over the workgroup
        base_acc += red;
        const int lay = lay0 + l, z = P.lay_count_begin + lay;
        if (b0 == 0 && l == 0) {
            // The FIRST block publishes the totals and the layer marks: everything here follows from the count pass's blockcnt[], the
            // first block starts first and its extra work hides behind the rest of the launch (the last block, which did this in
            // rounds 1-3, starts last and was the launch's tail)
            publish_compaction(P, nlay, nlog);
        }
        if (i0 < nseg) {
            uint32_t pos = base_acc + pre;
            uint32_t* rowstart = P.rowstart + (size_t)lay * P.ncy;
            int y = y0, xw = xw0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + k < nseg) {
                    if (xw == 0) rowstart[y] = pos;   // first record of cell row (z, y)
                    const uint32_t yz = (uint32_t)y << P.xbits;
                    uint64_t mk = m[k];
                    while (mk) {
                        const int bit = __builtin_ctzll(mk);
                        mk &= mk - 1;
                        if (pos < P.cap_active) {
                            P.rec_xy[pos] = (uint32_t)(xw * 64 + bit) | yz;
                            P.rec_z[pos] = (uint32_t)z;
                        }
                        pos++;
                    }
                    if (++xw == P.nxw) { xw = 0; y++; }
                }
            }
        }
    }
    if (!WRITE) {
        if (P.zero_cull && blockIdx.x == 0 && threadIdx.x < 64) P.zero_cull[32 * threadIdx.x] = 0u;   // (read for the last time by the kernel before this one)
        // low 32 bits: active cells of the block; high 32 bits: its case-13 sign words (rare)
        __shared__ uint32_t s_cnt[K2_LPB][4], s_n13[K2_LPB];
        if (threadIdx.x < K2_LPB) s_n13[threadIdx.x] = 0;
        uint32_t wsum[K2_LPB];
#pragma unroll
        for (int l = 0; l < K2_LPB; l++) wsum[l] = wave_sum_u32(cnts[l]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int l = 0; l < K2_LPB; l++)
                if (l < nl) {
                    s_cnt[l][wave] = wsum[l];
                    P.wavecnt[(b0 + l * P.bpl) * 4 + wave] = wsum[l];   // lets the write pass skip empty wavefronts
                }
        }
#pragma unroll
        for (int l = 0; l < K2_LPB; l++)
            if (n13s[l]) atomicAdd(&s_n13[l], n13s[l]);
        __syncthreads();
        if ((int)threadIdx.x < nl) {
            const int l = (int)threadIdx.x;
            P.blockcnt[b0 + l * P.bpl] = (uint64_t)(s_cnt[l][0] + s_cnt[l][1] + s_cnt[l][2] + s_cnt[l][3]) | ((uint64_t)s_n13[l] << 32);
        }
    }
}

// The write pass with the lanes' segments INTERLEAVED (SDFK_COMPACT_STRIDED): lane L of a block takes segments L, L + 256, L + 512 and
// L + 768 of the block's 1024 instead of four consecutive ones.  A run of busy segments -- a row near a tangent of the surface
// holds up to 64 records per segment -- then spreads over neighbouring lanes instead of piling up to 256 records on one lane, whose
// serial walk over its set bits was the tail of the launch (round 4's timeline: writing 1.0 us for the first wavefront, up to 5 us of
// waiting for the slowest).  Quarter k of the block is exactly what wavefront k of the COUNT pass covered, so "nothing there" is a
// wave-uniform test per quarter.  One 64-bit workgroup scan of the four packed per-quarter counts (a quarter holds at most
// 256 x 64 = 2^14 records) gives every segment its place in sweep order.
#ifndef SDFK_KW_MINWAVES
#define SDFK_KW_MINWAVES 1
#endif
template <bool MASKS>   // MASKS: P.segmask holds the count pass's activity masks
__global__ __launch_bounds__(256, MASKS ? SDFK_KW_MINWAVES : 1) void k_compact_write(McParams P)
{
    const int nlay = P.lay_list_end - P.lay_count_begin, nlog = nlay * P.bpl;
    const int b = compact_block_of_workgroup(nlay, P.bpl);   // logical block = (layer, part)
    const int lay = b / P.bpl, part = b - lay * P.bpl;
    const int z = P.lay_count_begin + lay;
    const int nseg = P.ncy * P.nxw;
    const int s0 = part * 1024 + (int)threadIdx.x;
    // this lane's share of the counts of all blocks before b (in flight together with the sign words below)
    // (grids of many blocks: ONE word, left by k_blockscan -- summing 16 K predecessors per workgroup is 16 dependent round trips)
    uint32_t before = 0;
    if (P.blockpre) {
        if (threadIdx.x == 0) before = P.blockpre[b];
    } else {
        for (int i = threadIdx.x; i < b; i += 1024) {
            const uint32_t c0 = (uint32_t)P.blockcnt[i];
            const uint32_t c1 = i + 256 < b ? (uint32_t)P.blockcnt[i + 256] : 0u;
            const uint32_t c2 = i + 512 < b ? (uint32_t)P.blockcnt[i + 512] : 0u;
            const uint32_t c3 = i + 768 < b ? (uint32_t)P.blockcnt[i + 768] : 0u;
            before += (c0 + c1) + (c2 + c3);
        }
    }
    bool look[4];
#pragma unroll
    for (int k = 0; k < 4; k++) look[k] = P.wavecnt[b * 4 + k] != 0;   // (uniform over the workgroup)
    uint64_t m[4] = {0, 0, 0, 0};
    int ys[4] = {0, 0, 0, 0}, xws[4] = {0, 0, 0, 0};
    if (MASKS) {
        // the count pass left the masks of the quarters that hold something: one 8-byte load per segment (512 contiguous bytes per
        // wavefront instruction) instead of four 16-byte loads and the activity logic again -- 40 instead of 96 VGPRs
        const uint64_t* q = P.segmask + (size_t)b * 1024u + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (look[k] && s0 + 256 * k < nseg) m[k] = q[256 * k];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = s0 + 256 * k;
            if (s < nseg) { ys[k] = s / P.nxw; xws[k] = s - ys[k] * P.nxw; }
        }
    } else {
        const size_t plane = (size_t)P.ny * P.nxw;
        const uint64_t* f0 = P.bits + (size_t)z * plane;
        u64x2u wa[4], wb[4], wc[4], wd[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {      // all loads first
            const int s = s0 + 256 * k;
            wa[k] = wb[k] = wc[k] = wd[k] = u64x2u{0, 0};
            if (look[k] && s < nseg) {
                const uint64_t* q = f0 + s;
                wa[k] = *reinterpret_cast<const u64x2u*>(q);
                wb[k] = *reinterpret_cast<const u64x2u*>(q + P.nxw);
                wc[k] = *reinterpret_cast<const u64x2u*>(q + plane);
                wd[k] = *reinterpret_cast<const u64x2u*>(q + plane + P.nxw);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s = s0 + 256 * k;
            if (s < nseg) {
                ys[k] = s / P.nxw;
                xws[k] = s - ys[k] * P.nxw;
                if (look[k]) {
                    uint64_t m13;
                    m[k] = segment_mask(wa[k].x, wa[k].y, wb[k].x, wb[k].y, wc[k].x, wc[k].y, wd[k].x, wd[k].y, P.ncx - xws[k] * 64, m13);
                }
            }
        }
    }
    // segment order within the block = (quarter k, lane): exclusive prefix of the packed per-quarter counts over the lanes, then the
    // totals of the quarters before
    __shared__ uint64_t s_wave[4];
    const uint64_t mine = (uint64_t)__popcll(m[0]) | ((uint64_t)__popcll(m[1]) << 16) | ((uint64_t)__popcll(m[2]) << 32) | ((uint64_t)__popcll(m[3]) << 48);
    uint64_t total;
    const uint64_t pre = block_excl_scan_u64(mine, s_wave, &total);
    __shared__ uint32_t s_before[4];
    {
        const uint32_t r = wave_sum_u32(before);
        if ((threadIdx.x & 63) == 0) s_before[threadIdx.x >> 6] = r;
    }
    __syncthreads();
    const uint32_t base = s_before[0] + s_before[1] + s_before[2] + s_before[3];
    if (b == 0 && !P.blockpre) {
        // the first block publishes the totals and the layer marks (everything follows from the count pass's blockcnt[])
        publish_compaction(P, nlay, nlog);
    }
    uint32_t* rowstart = P.rowstart + (size_t)lay * P.ncy;
    uint32_t qbase = base;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = s0 + 256 * k;
        if (s < nseg) {
            uint32_t pos = qbase + (uint32_t)((pre >> (16 * k)) & 0xffffull);
            if (xws[k] == 0) rowstart[ys[k]] = pos;   // first record of cell row (z, y)
            const uint32_t yz = (uint32_t)ys[k] << P.xbits;
            uint64_t mk = m[k];
            while (mk) {
                const int bit = __builtin_ctzll(mk);
                mk &= mk - 1;
                if (pos < P.cap_active) {
                    P.rec_xy[pos] = (uint32_t)(xws[k] * 64 + bit) | yz;
                    P.rec_z[pos] = (uint32_t)z;
                }
                pos++;
            }
        }
        qbase += (uint32_t)((total >> (16 * k)) & 0xffffull);
    }
}

// Grids of more than MC_SCAN_BLOCKS logical blocks (1024^3: 16 K): the exclusive prefix of the count pass's cell counts, ONE workgroup of
// 1024 lanes between the two passes (each lane a run of consecutive blocks), and the totals the write pass's first block publishes
// otherwise.  Below that size a write-pass workgroup sums its few predecessors itself, in flight with its sign words, and a launch
// would only add its latency.
// (both scans: a lane takes SCAN_PER consecutive entries, all loaded before the first is used -- 16 K entries are ONE trip of loads, one
// workgroup scan and one trip of stores)
constexpr int SCAN_PER = 16;
template <class T>
__device__ __forceinline__ T scan_1024(T sum, T* s_w /*[16]*/, T* tile_total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    __syncthreads();   // (s_w of the previous tile has been read)
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    T pre = incl - sum, tile = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const T v = s_w[w];
        if (w < wave) pre += v;
        tile += v;
    }
    *tile_total = tile;
    return pre;
}

__global__ __launch_bounds__(1024) void k_blockscan(McParams P)
{
    const int nlay = P.lay_list_end - P.lay_count_begin, nlog = nlay * P.bpl;
    __shared__ uint32_t s_w[16];
    uint32_t carry = 0;
    for (int t0 = 0; t0 < nlog; t0 += 1024 * SCAN_PER) {
        const int i = t0 + SCAN_PER * (int)threadIdx.x;
        uint32_t c[SCAN_PER], sum = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; k++) c[k] = i + k < nlog ? (uint32_t)P.blockcnt[i + k] : 0u;
#pragma unroll
        for (int k = 0; k < SCAN_PER; k++) sum += c[k];
        uint32_t tile;
        uint32_t pre = carry + scan_1024(sum, s_w, &tile);
#pragma unroll
        for (int k = 0; k < SCAN_PER; k++) {
            if (i + k < nlog) P.blockpre[i + k] = pre;
            pre += c[k];
        }
        carry += tile;
    }
    publish_compaction(P, nlay, nlog);
}

// Sum of the per-chunk (vertices << 31 | triangles) totals of chunks [from, to) over the
// workgroup (a few thousand words at most, L2 resident): every workgroup derives the prefix of
// its own chunk this way instead of waiting for a separate scan launch.
// (in two halves, so that a caller can issue these loads together with its other loads: the lane's share, four
// independent loads per trip -- a plain "load; add" loop is not pipelined by the compiler and costs one L2 round trip per
// 256 chunks, eight in a row for the last workgroups of a 512^3 sphere --, then the workgroup's sum)
__device__ __forceinline__ uint64_t chunk_totals_lane(const uint64_t* chunktot, uint32_t from, uint32_t to)
{
    uint64_t v = 0;
    for (uint32_t i = from + threadIdx.x; i < to; i += 1024u) {
        const uint64_t a = chunktot[i];
        const uint64_t b = i + 256u < to ? chunktot[i + 256u] : 0ull;
        const uint64_t c = i + 512u < to ? chunktot[i + 512u] : 0ull;
        const uint64_t d = i + 768u < to ? chunktot[i + 768u] : 0ull;
        v += (a + b) + (c + d);
    }
    return v;
}
__device__ __forceinline__ uint64_t chunk_totals_block(uint64_t v, uint64_t* s_part /*[4]*/)
{
    v = wave_sum_u64(v);
    __syncthreads();   // s_part may still be read from a previous call
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}
__device__ __forceinline__ uint64_t chunk_totals_sum(const uint64_t* chunktot, uint32_t from, uint32_t to, uint64_t* s_part /*[4]*/)
{
    return chunk_totals_block(chunk_totals_lane(chunktot, from, to), s_part);
}
// This lane's share of the totals of chunks [0, to): summed from chunktot, or -- long lists, P.chunkscan -- the one word k_chunkscan left
__device__ __forceinline__ uint64_t chunk_prefix_lane(const McParams& P, uint32_t to)
{
    if (P.chunkscan) return threadIdx.x == 0 ? P.chunkpre[to] : 0ull;
    return chunk_totals_lane(P.chunktot, 0, to);
}

// Grand totals, the vertex count of the ghost layer, dead cells: published to the device
// counters and to the host mirror by ONE workgroup after k_resolve has completed (workgroup 0
// of k_vertices, or k_publish when the host wants the counts before emitting).
__device__ __forceinline__ void publish_totals(const McParams& P)
{
    const uint32_t nrec = min(P.counters->n_active, P.cap_active);
    const uint32_t nch = (nrec + MC_CHUNK - 1u) / MC_CHUNK;
    // vertices numbered before the first emitted cell: chunk prefix + in-chunk prefix
    const uint32_t i0 = min(P.counters->n_ghost_cells, nrec);
    const uint32_t c0 = i0 / MC_CHUNK;
    uint64_t all = chunk_prefix_lane(P, nch), upto = chunk_prefix_lane(P, c0), dead = 0;
    if (P.counters->n_case13 != 0)   // (k_resolve counts dead cells only in volumes that have such sign words at all)
        for (uint32_t i = threadIdx.x; i < nch; i += 1024u) {
            const uint32_t a = P.chunkdead[i];
            const uint32_t b = i + 256u < nch ? P.chunkdead[i + 256u] : 0u;
            const uint32_t c = i + 512u < nch ? P.chunkdead[i + 512u] : 0u;
            const uint32_t d = i + 768u < nch ? P.chunkdead[i + 768u] : 0u;
            dead += (uint64_t)(a + b) + (uint64_t)(c + d);
        }
    all = wave_sum_u64(all); upto = wave_sum_u64(upto); dead = wave_sum_u64(dead);
    __shared__ uint64_t s_tot[3][4];
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; s_tot[0][w] = all; s_tot[1][w] = upto; s_tot[2][w] = dead; }
    __syncthreads();
    if (threadIdx.x == 0) {
        all = s_tot[0][0] + s_tot[0][1] + s_tot[0][2] + s_tot[0][3];
        upto = s_tot[1][0] + s_tot[1][1] + s_tot[1][2] + s_tot[1][3];
        dead = s_tot[2][0] + s_tot[2][1] + s_tot[2][2] + s_tot[2][3];
        McCounters c = *P.counters;
        c.total_v = (uint32_t)(all >> 31);
        c.total_t = (uint32_t)(all & 0x7fffffffull);
        c.nghost = i0 < nrec ? (uint32_t)(upto >> 31) + (P.rec_pre[i0] & 0xffffu) : c.total_v;
        c.n_dead = (uint32_t)dead;
        c.overflow = 0;
        *P.counters = c;
        // mirror for the host (read after the stream has drained); `overflow` is only ever
        // raised afterwards, directly in the mirror, by the emit kernels
        McCounters* h = P.host_counters;
        h->n_active = c.n_active; h->n_case13 = c.n_case13; h->n_dead = c.n_dead; h->total_v = c.total_v;
        h->total_t = c.total_t; h->nghost = c.nghost; h->n_ghost_cells = c.n_ghost_cells; h->n_emit_cells = c.n_emit_cells;
    }
}

__global__ __launch_bounds__(256) void k_publish(McParams P) { publish_totals(P); }

// Long record lists (capacities of more than MC_SCAN_CHUNKS chunks: 1024^3): the exclusive prefix of k_resolve's per-chunk totals, ONE
// workgroup of 1024 lanes between k_resolve and k_vertices; chunkpre[number of chunks] = the grand total.
__global__ __launch_bounds__(1024) void k_chunkscan(McParams P)
{
    const uint32_t nrec = min(P.counters->n_active, P.cap_active);
    const uint32_t nch = (nrec + MC_CHUNK - 1u) / MC_CHUNK;
    __shared__ uint64_t s_w[16];
    uint64_t carry = 0;
    for (uint32_t t0 = 0; t0 <= nch; t0 += 1024u * SCAN_PER) {   // (<=: the entry after the last chunk gets the total)
        const uint32_t i = t0 + (uint32_t)SCAN_PER * threadIdx.x;
        uint64_t c[SCAN_PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)SCAN_PER; k++) c[k] = i + k < nch ? P.chunktot[i + k] : 0ull;
#pragma unroll
        for (int k = 0; k < SCAN_PER; k++) sum += c[k];
        uint64_t tile;
        uint64_t pre = carry + scan_1024(sum, s_w, &tile);
#pragma unroll
        for (uint32_t k = 0; k < (uint32_t)SCAN_PER; k++) {
            if (i + k <= nch) P.chunkpre[i + k] = pre;
            pre += c[k];
        }
        carry += tile;
    }
}

// ---------------------------------------------------------------------------
// shared helpers
// ---------------------------------------------------------------------------
// 4-byte-aligned vector types: gfx950 (unaligned access mode) loads them with one instruction
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// the 8 corner voxels of cell (x,y,z) into this thread's LDS column ([corner][thread]); nz = the volume's ROW PITCH
__device__ __forceinline__ void stage_corners(const float* values, int ny, int nz, int x, int y, int z, float* col, int stride)
{
    // the two z-neighbours of a corner pair are adjacent in memory: four 8-byte loads
    const size_t sx = (size_t)ny * nz, sy = (size_t)nz;
    const float* p = values + (size_t)x * sx + (size_t)y * sy + z;
    const f2u q0 = *reinterpret_cast<const f2u*>(p), q1 = *reinterpret_cast<const f2u*>(p + sx);
    const f2u q3 = *reinterpret_cast<const f2u*>(p + sy), q2 = *reinterpret_cast<const f2u*>(p + sx + sy);
    col[0] = q0.x; col[stride] = q1.x; col[2 * stride] = q2.x; col[3 * stride] = q3.x;
    col[4 * stride] = q0.y; col[5 * stride] = q1.y; col[6 * stride] = q2.y; col[7 * stride] = q3.y;
}

__device__ __forceinline__ void corners_to_column(const float4& lo, const float4& hi, float* col)
{
    col[0] = lo.x; col[256] = lo.y; col[2 * 256] = lo.z; col[3 * 256] = lo.w;
    col[4 * 256] = hi.x; col[5 * 256] = hi.y; col[6 * 256] = hi.z; col[7 * 256] = hi.w;
}

__device__ __forceinline__ bool cell_in_range(const McParams& P, int x, int y, int z)
{
    // z is a LOCAL layer index; the slab holds every layer that exists globally around the
    // layers it emits, so local range == global range for all cells this is asked about.
    return x >= 0 && y >= 0 && z >= 0 && x < P.ncx && y < P.ncy && z < P.ncz;
}

// "Impossible case 13" cells emit nothing and therefore never create or reference a vertex.
// `col` is the calling thread's LDS column ([corner][256 threads]); its content is replaced.
// (takes the few fields it needs by value: a reference to the kernel-argument block would
// force a scratch copy of it at this out-of-line call)
__device__ __noinline__ bool cell_is_dead(const int8_t* lut, const float* values, int ny, int nz, float iso,
                                          int x, int y, int z, float* col)
{
    stage_corners(values, ny, nz, x, y, z, col, 256);
    const CornersLds v{col, 256, (double)iso};
    int index = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) index |= (v[k] > 0.0) ? (1 << k) : 0;
    if (index != 0xA5 && index != 0x5A) return false;
    const Tiling t = mc_resolve(lut, v);
    return t.nt == 0;
}

// Does an earlier LIVE cell of the sweep share edge e of cell (x,y,z)?  Predecessor sets
// derived from Cell.cs:371-441 (which cells map to the same face-layer slot) and the sweep
// order of MarchingCubes.cs:53-80.  Up to three predecessors per edge, offsets (dx,dy,dz); 9 = none.
__constant__ int8_t c_pred[12][3][3] = {
    {{0, -1, -1}, {0, 0, -1}, {0, -1, 0}},   // e0
    {{0, 0, -1}, {1, 0, -1}, {9, 9, 9}},     // e1
    {{0, 0, -1}, {0, 1, -1}, {9, 9, 9}},     // e2
    {{-1, 0, -1}, {0, 0, -1}, {-1, 0, 0}},   // e3
    {{0, -1, 0}, {9, 9, 9}, {9, 9, 9}},      // e4
    {{9, 9, 9}, {9, 9, 9}, {9, 9, 9}},       // e5
    {{9, 9, 9}, {9, 9, 9}, {9, 9, 9}},       // e6
    {{-1, 0, 0}, {9, 9, 9}, {9, 9, 9}},      // e7
    {{-1, -1, 0}, {0, -1, 0}, {-1, 0, 0}},   // e8
    {{0, -1, 0}, {1, -1, 0}, {9, 9, 9}},     // e9
    {{9, 9, 9}, {9, 9, 9}, {9, 9, 9}},       // e10
    {{-1, 0, 0}, {9, 9, 9}, {9, 9, 9}}};     // e11

__device__ __forceinline__ bool edge_has_live_predecessor(const McParams& P, const int8_t* lut, int e, int x, int y,
                                                          int z, bool check_dead, float* col)
{
    for (int k = 0; k < 3; k++) {
        const int dx = c_pred[e][k][0];
        if (dx == 9) break;
        const int px = x + dx, py = y + c_pred[e][k][1], pz = z + c_pred[e][k][2];
        if (px < 0 || py < 0 || px >= P.ncx || py >= P.ncy) continue;
        if (pz < 0) {
            // below the slab: exists globally iff this is not global layer 0; assumed alive
            // (only reachable for context layers whose counts are never used)
            if (P.z0 + pz >= 0) return true;
            continue;
        }
        if (pz >= P.ncz) continue;
        if (!check_dead || !cell_is_dead(lut, P.values, P.ny, P.nzp, P.iso, px, py, pz, col)) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------
// K3: resolve tilings + vertex creation; per-chunk totals of (vertices, triangles)
// ---------------------------------------------------------------------------
// Which of its 12 edges does cell (x,y,z) CREATE when no cell of the volume is "dead"?
// Exactly those no earlier in-range cell of the sweep shares (table c_pred reduced to the
// three facts x>0, y>0, z>0): bit e set = created here.
__device__ __forceinline__ unsigned positional_own_mask(bool X, bool Y, bool Z)
{
    unsigned m = (1u << 5) | (1u << 6) | (1u << 10);
    if (!Y) m |= (1u << 4) | (1u << 9);
    if (!X) m |= (1u << 7) | (1u << 11);
    if (!Z) m |= (1u << 1) | (1u << 2);
    if (!Y && !Z) m |= 1u << 0;
    if (!X && !Z) m |= 1u << 3;
    if (!X && !Y) m |= 1u << 8;
    return m;
}

// K3a: corner values of every active cell -> its record.  The only sparse access to the
// volume in the whole pipeline; one lane per record, four 8-byte loads in flight per lane,
// full occupancy; the stores are 32 contiguous bytes per lane.
__global__ __launch_bounds__(256) void k_gather_corners(McParams P)
{
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const size_t sx = (size_t)P.ny * P.nzp, sy = (size_t)P.nzp;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t xy = P.rec_xy[i];
        const float* p = P.values + (size_t)(xy & P.xmask) * sx + (size_t)(xy >> P.xbits) * sy + P.rec_z[i];
        const f2u q0 = *reinterpret_cast<const f2u*>(p), q1 = *reinterpret_cast<const f2u*>(p + sx);
        const f2u q3 = *reinterpret_cast<const f2u*>(p + sy), q2 = *reinterpret_cast<const f2u*>(p + sx + sy);
        *reinterpret_cast<float4*>(P.rec_corners + (size_t)i * 8) = make_float4(q0.x, q1.x, q2.x, q3.x);
        *reinterpret_cast<float4*>(P.rec_corners + (size_t)i * 8 + 4) = make_float4(q0.y, q1.y, q2.y, q3.y);
    }
}

// (five workgroups per CU -- what the 27.5 KB of LDS allow: the register allocator then stays at 96 VGPRs without a
// spill, against 153 VGPRs / three workgroups per CU unconstrained; 17.2 instead of 20.3 us at 512^3)
__global__ __launch_bounds__(256, 5) void k_resolve(McParams P)
{
    __shared__ float s_v[8 * 256];   // [corner][thread]: run-time corner indexing without scratch
    __shared__ __attribute__((aligned(16))) int8_t s_lut[MCDEC_PADDED];   // the decision tables only (mc_device.h): 1.1 KB
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_ord[MCLUT_NROWS];
    __shared__ uint32_t s_rows[2][2];   // [chunk parity][first, last]: row of the chunk's first / last record
    mc_load_dec_to_lds(s_lut);
    {   // per-row creation order -> LDS (independent loads per lane)
        const int t = (int)threadIdx.x;
        static_assert(MCLUT_NROWS <= 768, "row table copy assumes <= 3 rounds");
        const uint64_t a0 = c_roword[min(t, MCLUT_NROWS - 1)], a1 = c_roword[min(t + 256, MCLUT_NROWS - 1)];
        const uint64_t a2 = c_roword[min(t + 512, MCLUT_NROWS - 1)];
        if (t < MCLUT_NROWS) s_ord[t] = a0;
        if (t + 256 < MCLUT_NROWS) s_ord[t + 256] = a1;
        if (t + 512 < MCLUT_NROWS) s_ord[t + 512] = a2;
    }
    __syncthreads();
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    // (a volume without storage -- SDFK_OPT_ELIDE_VOLUME -- has no voxels for the dead-cell test to read: the host sees the
    // same counter, discards this job's result and redoes it on a volume that has them)
    const bool any13 = P.counters->n_case13 != 0;
    const bool check_dead = any13 && P.values != nullptr;
    const int nchunks = (int)((n + MC_CHUNK - 1u) / MC_CHUNK);
    float* col = s_v + threadIdx.x;
    int parity = 0;
    for (int c = (int)first_chunk_of_block<SDFK_KR_XCD_GROUP>(); c < nchunks; c += gridDim.x, parity ^= 1) {
        const uint32_t i = (uint32_t)c * MC_CHUNK + threadIdx.x;
        const bool mine_rec = threadIdx.x < MC_CHUNK && i < n;   // (lanes MC_CHUNK..255 idle here: see MC_CHUNK)
        uint32_t nown = 0, nt_emit = 0, nslots = 0, info = 0, dead = 0;
        uint64_t own = 0;
        int ry = 0, rz = 0;
        if (mine_rec) {
            const uint32_t xy = P.rec_xy[i];
            const int x = (int)(xy & P.xmask), y = (int)(xy >> P.xbits), z = (int)P.rec_z[i];
            ry = y; rz = z;
            corners_to_column(*reinterpret_cast<const float4*>(P.rec_corners + (size_t)i * 8),
                              *reinterpret_cast<const float4*>(P.rec_corners + (size_t)i * 8 + 4), col);
            const CornersLds v{col, 256, (double)P.iso};
            const Tiling t = mc_resolve(s_lut, v);
            const bool counted = z < P.lay_emit_end;     // the layer above is context only
            const bool emit = counted && z >= P.lay_emit_begin;
            if (t.nt > 0) {
                const uint64_t ord = s_ord[t.row];   // the row's vertex ids in the order of their first reference = creation order
                const int nd = (int)(ord >> 60);     // ... and their number
                if (counted) {
                    const unsigned pmask = positional_own_mask(x > 0, y > 0, P.z0 + z > 0) | (1u << 12);
                    for (int k = 0; k < nd; k++) {
                        const int e = (int)((ord >> (4 * k)) & 15ull);
                        bool mine = (pmask >> e) & 1u;
                        if (check_dead && !mine)   // an earlier sharer exists: is any of them alive?
                            mine = !edge_has_live_predecessor(P, s_lut, e, x, y, z, true, col);
                        if (mine) {
                            own |= (uint64_t)e << (4 * nown);
                            nown++;
                        }
                    }
                }
                nt_emit = emit ? (uint32_t)t.nt : 0u;
                nslots = emit ? (uint32_t)nd : 0u;   // one vertex-id slot per distinct id of the row (mc_device.h): only k_triangles reads them
                info = (1u << 13) | (nt_emit << 14) | (nown << 18) | ((uint32_t)t.row << 22);   // (+ the slot prefix, below)
            } else if (emit && (t.index == 0xA5 || t.index == 0x5A)) {
                dead = 1;
            }
        }
        // rows spanned by the chunk (lets K4 fetch its windows without first reading records): its first and last record are
        // lanes 0 and cnt - 1, which hold their coordinates already
        const uint32_t first = (uint32_t)c * MC_CHUNK, last = min(first + MC_CHUNK - 1u, n - 1u);
        if (mine_rec && (i == first || i == last)) {
            const uint32_t row = (uint32_t)(rz - P.lay_count_begin) * (uint32_t)P.ncy + (uint32_t)ry;
            // (two buffers: the next chunk's rows are written before a barrier separates them from this chunk's reads)
            if (i == first) s_rows[parity][0] = row;
            if (i == last) s_rows[parity][1] = row;
        }
        // in-chunk prefix and chunk total of (vertex-id slots, created vertices, triangles): 21 bits each (at most 13 * 256 per chunk)
        uint64_t total;
        const uint64_t pre = block_excl_scan_u64(((uint64_t)nslots << 42) | ((uint64_t)nown << 21) | nt_emit, s_wave, &total);   // (syncs: s_rows is complete)
        // ... and the record ranges of its two neighbour windows (see k_vertices): three rowstart loads, one per lane, issued
        // HERE so that they are in flight during the stores and the second scan below (thread 0 loading them one after the
        // other at the very end left every workgroup waiting for two more L2 round trips)
        const uint32_t rf = s_rows[parity][0], rl = s_rows[parity][1];
        const uint32_t nrows = (uint32_t)((P.lay_list_end - P.lay_count_begin) * P.ncy);
        const uint32_t nrs = min(rl - rf + 3u, (uint32_t)K4_RMAX);
        uint32_t wrow = 0;
        if (threadIdx.x < 3) {
            const uint32_t r = threadIdx.x == 0 ? rf + nrs - 1u : (threadIdx.x == 1 ? rf + (uint32_t)P.ncy : rf + (uint32_t)P.ncy + nrs - 1u);
            wrow = min(P.rowstart[min(r, nrows)], n);
        }
        if (mine_rec) {
            P.rec_info[i] = info ? info | (uint32_t)(pre >> 42) : 0u;
            P.rec_own[i] = own;
            P.rec_pre[i] = (uint32_t)((pre >> 21) & 0x1fffffull) | ((uint32_t)(pre & 0x1fffffull) << 16);
        }
        // "impossible case 13" cells of the chunk: only volumes that have such sign words at all need the count
        uint64_t ndead = 0;
        if (any13) (void)block_excl_scan_u64(dead, s_wave, &ndead);
        const uint32_t w1e = __shfl(wrow, 0), w2s = __shfl(wrow, 1), w2e = __shfl(wrow, 2);
        if (threadIdx.x == 0) {
            P.chunktot[c] = (((total >> 21) & 0x1fffffull) << 31) | (total & 0x1fffffull);   // v << 31 | t, as every consumer sums them
            P.chunkslots[c] = (uint32_t)(total >> 42);
            P.chunkdead[c] = (uint32_t)ndead;
            P.chunkwin[c] = make_uint4(rf, rl, w1e, w2s);
            P.chunkwin2[c] = w2e;
        }
    }
}

// ---------------------------------------------------------------------------
// K4 / K5 helpers
// ---------------------------------------------------------------------------
// Workgroup-level re-balancing: a chunk of MC_CHUNK cells owns a variable number of output items
// each (created vertices / triangle indices).  An exclusive prefix in LDS plus a small
// item -> cell table (filled by the cells' own lanes) turns "one lane per cell" into "one lane
// per output item".
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_pre /*[257]*/, uint32_t* s_wave /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    __syncthreads();   // previous chunk is done with s_pre / s_wave
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - v;
    for (int w = 0; w < wave; w++) pre += s_wave[w];
    s_pre[threadIdx.x] = pre;
    if (threadIdx.x == 255) s_pre[256] = pre + v;
    __syncthreads();
    return s_pre[256];
}

// Record index of active cell (cx,cy,cz), or -1.  Records are sorted by (z, y, x) and
// rowstart[] gives the span of each cell row, so this is a search over the handful of
// active cells of one row.  Cells sharing a sign-changing edge are always active, hence
// always found when their layer is listed.
// `n` = number of records actually stored (the list may have been cut at its capacity when a
// speculative launch under-estimated it; row starts beyond that must never be dereferenced).
__device__ __forceinline__ int find_record(const McParams& P, uint32_t n, int cx, int cy, int cz)
{
    if (cz < P.lay_count_begin || cz >= P.lay_list_end) return -1;
    const uint32_t* rs = P.rowstart + (size_t)(cz - P.lay_count_begin) * P.ncy + cy;
    uint32_t a = min(rs[0], n), b = min(rs[1], n);
    while (b - a > 4u) {   // long rows: bisect
        const uint32_t mid = (a + b) >> 1;
        if ((int)(P.rec_xy[mid] & P.xmask) <= cx) a = mid; else b = mid;
    }
    for (uint32_t k = a; k < b; k++)
        if ((int)(P.rec_xy[k] & P.xmask) == cx) return (int)k;
    return -1;
}

__device__ __forceinline__ float v3len(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

__device__ __forceinline__ void load_corner_color(const McParams& P, int x, int y, int z, int corner, float* c)
{
    const size_t o = ((size_t)(x + mc_corner_dx(corner)) * P.ny + (y + mc_corner_dy(corner))) * P.nzp + (z + mc_corner_dz(corner));
    const f3u q = *reinterpret_cast<const f3u*>(P.colors + o * 3);
    c[0] = q.x; c[1] = q.y; c[2] = q.z;
}

// the two end corners of cube edge e (Luts.cs:30-52), in the order of Luts.edgesrel{x,y,z}
__device__ __forceinline__ int mc_edge_corner_a(int e) { return e < 8 ? e : e - 8; }
__device__ __forceinline__ int mc_edge_corner_b(int e) { return e < 8 ? ((e & 4) | ((e + 1) & 3)) : e - 4; }

// ---------------------------------------------------------------------------
// K4: vertices
// ---------------------------------------------------------------------------
// Records are sorted by (z, y, x), so everything a chunk of MC_CHUNK consecutive records needs
// from its neighbours lives in two CONTIGUOUS windows of the record list:
//   W1 = the chunk itself up to the end of the row after its last row   (+x, +y sharers)
//   W2 = the same rows (+1) of the next layer                           (+z, +y+z sharers)
// (a vertex is created by the FIRST cell of the sweep around its edge, so the other sharers
// are always at +x/+y/+z).  The windows -- coordinates, tiling info, the 8 corner values --
// and the matching slices of rowstart[] are staged in LDS with coalesced loads; the
// per-vertex work then runs on LDS only.  Sharers that fall outside a truncated window take
// a slow path through global memory.

struct CornersGlobal {          // 8 corner values of a record, read from global memory (slow path)
    const float* p;
    double iso;
    __device__ __forceinline__ double operator[](int k) const { return (double)p[k] - iso; }
};

// contribution of one sharer cell to the normal of the vertex on its edge `es`, in the order
// of Cell.cs:332-333/355-356: once per occurrence in the LUT row, corner 1 then corner 2
template <class V>
__device__ __forceinline__ void add_sharer_gradients(const V& vs, int es, int occ, int dir, double w_lo, double w_hi, float* nrm)
{
    const int a = mc_edge_corner_a(es), b = mc_edge_corner_b(es);
    const int da = dir == 0 ? mc_corner_dx(a) : (dir == 1 ? mc_corner_dy(a) : mc_corner_dz(a));
    const double wa = da ? w_hi : w_lo, wb = da ? w_lo : w_hi;
    // NB: Cell.cs:157-158 indexes the corner-ordered gradient table with the BIT-order index
    // (inherited quirk); reproduced.
    const int i1 = mc_bit_to_corner(a), i2 = mc_bit_to_corner(b);   // involution: corner -> bit order
    float g1[3], g2[3];
#pragma unroll
    for (int jj = 0; jj < 3; jj++) {
        g1[jj] = (float)(mc_corner_gradient(vs, i1, jj) * wa);
        g2[jj] = (float)(mc_corner_gradient(vs, i2, jj) * wb);
    }
    for (int o = 0; o < occ; o++) {
        nrm[0] = nrm[0] + g1[0]; nrm[1] = nrm[1] + g1[1]; nrm[2] = nrm[2] + g1[2];
        nrm[0] = nrm[0] + g2[0]; nrm[1] = nrm[1] + g2[1]; nrm[2] = nrm[2] + g2[2];
    }
}

#ifndef SDFK_KV_MINWAVES
#define SDFK_KV_MINWAVES 4   // (wavefronts per SIMD the register allocation leaves room for: four workgroups per CU, see K4_WMAX)
#endif
template <bool ISO0>   // ISO0: P.iso == +0.0 (CornersLdsT, mc_device.h)
__global__ __launch_bounds__(256, SDFK_KV_MINWAVES) void k_vertices(McParams P, McMeshOut M)
{
    __shared__ float s_wc[8 * K4_WMAX];          // window corner values: [corner][slot]
    __shared__ uint32_t s_wxy[K4_WMAX], s_winfo[K4_WMAX];
    __shared__ uint16_t s_rs[2][K4_RMAX];   // row starts RELATIVE to their window: 0 = before it, 1 + i = its record i, count + 2 = beyond
    __shared__ uint64_t s_occ[MCLUT_NROWS];
    __shared__ uint32_t s_pre[257];
    __shared__ uint64_t s_own[256];
    __shared__ uint32_t s_z[256];
    __shared__ uint8_t s_creator[256 * 13];   // creator record (slot in the chunk) of each vertex of the chunk
    __shared__ float s_red[6][4];
    {   // per-row reference counts -> LDS (independent loads per lane)
        const int t = (int)threadIdx.x;
        static_assert(MCLUT_NROWS <= 768, "row table copy assumes <= 3 rounds");
        const uint64_t a0 = c_rowocc[min(t, MCLUT_NROWS - 1)], a1 = c_rowocc[min(t + 256, MCLUT_NROWS - 1)];
        const uint64_t a2 = c_rowocc[min(t + 512, MCLUT_NROWS - 1)];
        if (t < MCLUT_NROWS) s_occ[t] = a0;
        if (t + 256 < MCLUT_NROWS) s_occ[t + 256] = a1;
        if (t + 512 < MCLUT_NROWS) s_occ[t + 512] = a2;
    }
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const int nrows_total = (P.lay_list_end - P.lay_count_begin) * P.ncy;
    const double iso = (double)P.iso;
    const double stp = (double)P.step;
    // corner colours: 12-byte gathers from the colour volume (1 MiB apart along x) -- unless the program that sampled the
    // volume re-evaluates them afterwards (M.vdesc)
    const bool gather_colors = P.colors != nullptr && M.vdesc == nullptr;
    float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    __shared__ uint64_t s_part[4];
    // vertices numbered below the first emitted layer (slab runs): prefix of record n_ghost_cells
    uint32_t nghost = 0;
    {
        const uint32_t i0 = min(P.counters->n_ghost_cells, n);
        if (i0 > 0)   // (all records ghost: every chunk counts)
            nghost = (uint32_t)(chunk_totals_block(chunk_prefix_lane(P, i0 < n ? i0 / MC_CHUNK : (n + MC_CHUNK - 1u) / MC_CHUNK), s_part) >> 31) +
                     (i0 < n ? (P.rec_pre[i0] & 0xffffu) : 0u);
    }
    if (blockIdx.x == 0) publish_totals(P);   // k_resolve has completed (stream order)
    uint64_t chunk_prefix = 0;
    uint32_t prefix_upto = 0;
    for (uint32_t base = first_chunk_of_block<SDFK_KV_XCD_GROUP>() * MC_CHUNK; base < n; base += gridDim.x * MC_CHUNK) {
        const uint32_t cnt = min(MC_CHUNK, n - base);
        const uint32_t ci = base / MC_CHUNK;   // chunk index
        const uint32_t irec = base + threadIdx.x;
        uint32_t my_nown = 0;
        __syncthreads();   // previous chunk is done with all s_* arrays
        // ---- one batch of loads: own chunk fields, rowstart slices of the two windows (rows
        // [r_f, r_l+2] and the same + ncy) and the windows themselves.  Rows and window
        // ranges were left per chunk by k_resolve, so nothing here waits on another load.
        const uint4 cw = P.chunkwin[ci];
        const int r_f = (int)cw.x, r_l = (int)cw.y;
        const int nrs = min(r_l - r_f + 3, K4_RMAX);
        const uint32_t w1_start = base;                              // +x / +y sharers come after the chunk start
        // (at least the chunk itself: the row slice may have been cut at K4_RMAX rows)
        const uint32_t w1_cnt = min(max(cw.z, base + cnt) - w1_start, (uint32_t)K4_WMAX);
        const uint32_t w2_start = cw.w;
        const uint32_t w2_cnt = min(P.chunkwin2[ci] - w2_start, (uint32_t)K4_WMAX - w1_cnt);
        // (the in-chunk prefix of the created vertices and the chunk's total were left by k_resolve -- rec_pre, chunktot --: two
        // more loads in this batch instead of a workgroup scan, three barriers, after it)
        uint32_t my_pre = 0;
        const uint32_t total = (uint32_t)(P.chunktot[ci] >> 31);
        if (threadIdx.x < cnt) {
            my_nown = (P.rec_info[irec] >> 18) & 15u;
            my_pre = P.rec_pre[irec] & 0xffffu;
            s_own[threadIdx.x] = P.rec_own[irec];
            s_z[threadIdx.x] = P.rec_z[irec];
        }
        for (int i = threadIdx.x; i < 2 * nrs; i += 256) {
            const int w = i >= nrs, k = w ? i - nrs : i;
            const uint32_t rs = min(P.rowstart[min(r_f + (w ? P.ncy : 0) + k, nrows_total)], n);   // never past the stored records
            const uint32_t ws = w ? w2_start : w1_start, wc = w ? w2_cnt : w1_cnt;
            // 0 = before the window, 1 + i = its record i, count + 2 = beyond.  (In 64-bit signed arithmetic on purpose: hipcc 7.0 -O3
            // compiles "rs < ws ? 0u : min(rs - ws, wc + 1u) + 1u" for gfx950 WITHOUT the comparison -- the wrapped difference, clamped:
            // "beyond" for a row that starts before the window.  tools/ubench/ub_sel.hip reproduces it in twenty lines.)
            const long long d = (long long)rs + 1 - (long long)ws;
            s_rs[w][k] = (uint16_t)max(0ll, min(d, (long long)wc + 2));
        }
        uint64_t prefix_lane = 0;
        {
            const uint32_t wtot = w1_cnt + w2_cnt;
            uint32_t rxy[3], rin[3];
            float4 rlo[3], rhi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t slot = threadIdx.x + 256u * k;
                if (slot < wtot) {
                    const uint32_t g = slot < w1_cnt ? w1_start + slot : w2_start + (slot - w1_cnt);
                    rxy[k] = P.rec_xy[g];
                    rin[k] = P.rec_info[g];
                    rlo[k] = *reinterpret_cast<const float4*>(P.rec_corners + (size_t)g * 8);
                    rhi[k] = *reinterpret_cast<const float4*>(P.rec_corners + (size_t)g * 8 + 4);
                }
            }
            // (the totals of the chunks before this one ride in the same batch of loads)
            prefix_lane = P.chunkscan ? (threadIdx.x == 0 ? P.chunkpre[ci] : 0ull) : chunk_totals_lane(P.chunktot, prefix_upto, ci);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t slot = threadIdx.x + 256u * k;
                if (slot < wtot) {
                    s_wxy[slot] = rxy[k];
                    s_winfo[slot] = rin[k];
                    float* c = s_wc + slot;
                    c[0] = rlo[k].x; c[K4_WMAX] = rlo[k].y; c[2 * K4_WMAX] = rlo[k].z; c[3 * K4_WMAX] = rlo[k].w;
                    c[4 * K4_WMAX] = rhi[k].x; c[5 * K4_WMAX] = rhi[k].y; c[6 * K4_WMAX] = rhi[k].z; c[7 * K4_WMAX] = rhi[k].w;
                }
            }
        }
        // chunk prefix = totals of all earlier chunks (advanced incrementally when a workgroup takes more than one chunk); left in
        // chunkpre[] for k_triangles (long lists: it comes from k_chunkscan, whole)
        if (P.chunkscan) chunk_prefix = chunk_totals_block(prefix_lane, s_part);
        else {
            chunk_prefix += chunk_totals_block(prefix_lane, s_part);
            prefix_upto = ci;
            if (threadIdx.x == 0) P.chunkpre[ci] = chunk_prefix;
        }
        s_pre[threadIdx.x] = threadIdx.x < cnt ? my_pre : total;
        if (threadIdx.x == 255) s_pre[256] = total;
        // vertex -> creator table (a cell creates at most 13): one LDS read per vertex instead of a search
        for (uint32_t k = 0; k < my_nown; k++) s_creator[my_pre + k] = (uint8_t)threadIdx.x;
        __syncthreads();   // (LDS is complete)
        const uint32_t chunk_vbase = (uint32_t)(chunk_prefix >> 31);
        // ---- per created vertex
        for (uint32_t j = threadIdx.x; j < total; j += 256u) {
            const int rr = (int)s_creator[j];          // = window slot of the creator (W1 starts at the chunk)
            const int r = (int)(j - s_pre[rr]);
            const uint32_t info = s_winfo[rr];
            const int x = (int)(s_wxy[rr] & P.xmask), y = (int)(s_wxy[rr] >> P.xbits);
            const int e = (int)((s_own[rr] >> (4 * r)) & 15u);
            const int dir = mc_edge_dir(e);
            const int z = (int)s_z[rr];
            const bool emit = z >= P.lay_emit_begin;   // the layer below a slab is numbered, not emitted
            const uint32_t vi = chunk_vbase + j;       // chunk prefix + in-chunk prefix: serial vertex id
            const uint32_t out = vi - nghost;
            if (emit && out >= M.cap_vertices) { P.host_counters->overflow = 1u; continue; }
            const int gx = x + mc_edge_ox(e), gy = y + mc_edge_oy(e), gz = z + mc_edge_oz(e);
            const int own_row = (int)(info >> 22);
            float pos[3] = {0.0f, 0.0f, 0.0f}, colr[3] = {0.0f, 0.0f, 0.0f}, nrm[3] = {0.0f, 0.0f, 0.0f};
            const int xs = x * P.step, ys = y * P.step, zs = (z + P.z0) * P.step;
            const CornersLdsT<ISO0> v{s_wc + rr, K4_WMAX, iso};   // the creator cell
            if (e == 12) {
                // (only the cell itself references its centre vertex: the last of its slots -- 12 is the highest id)
                if ((info >> 14) & 15u)
                    P.rec_vid[(size_t)ci * MC_VSTRIDE + (info & 0xfffu) + mc_slot_of(s_occ[own_row], 12)] = vi;
                if (!emit) continue;
                // Cell.CalculateCenterVertex, Cell.cs:501-549
                double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0, gsum[3] = {0.0, 0.0, 0.0};
                float fc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll 1
                for (int k = 0; k < 8; k++) {
                    const double wk = 1.0 / (MC_EPS + fabs(v[k]));
                    fx += (double)mc_corner_dx(k) * wk;
                    fy += (double)mc_corner_dy(k) * wk;
                    fz += (double)mc_corner_dz(k) * wk;
                    ff += wk;
                    if (gather_colors) {
                        float ck[3];
                        load_corner_color(P, x, y, z, k, ck);
                        const float wf = (float)wk;
                        if (k == 0) { fc[0] = ck[0] * wf; fc[1] = ck[1] * wf; fc[2] = ck[2] * wf; }
                        else { fc[0] = fc[0] + ck[0] * wf; fc[1] = fc[1] + ck[1] * wf; fc[2] = fc[2] + ck[2] * wf; }
                    }
#pragma unroll
                    for (int jj = 0; jj < 3; jj++) {
                        const double term = wk * mc_corner_gradient(v, k, jj);
                        gsum[jj] = (k == 0) ? term : gsum[jj] + term;
                    }
                }
                pos[0] = (float)((double)xs + stp * fx / ff);
                pos[1] = (float)((double)ys + stp * fy / ff);
                pos[2] = (float)((double)zs + stp * fz / ff);
                if (gather_colors) {
#pragma unroll
                    for (int jj = 0; jj < 3; jj++) colr[jj] = (float)((double)fc[jj] / ff);
                }
                const float g0 = (float)gsum[0], g1 = (float)gsum[1], g2 = (float)gsum[2];
                const int occ = (int)((s_occ[own_row] >> 48) & 15ull);
                for (int o = 0; o < occ; o++) { nrm[0] = nrm[0] + g0; nrm[1] = nrm[1] + g1; nrm[2] = nrm[2] + g2; }
            } else {
                double w_lo = 0.0, w_hi = 0.0;
                if (emit) {
                    // Cell.AddFaceFromEdgeIndex, Cell.cs:314-350, in the creator cell's frame.
                    // (index1,index2) of Cell.cs:318-319 are the bit-order ids of the edge's end corners.
                    const int c1 = mc_edge_corner_a(e), c2 = mc_edge_corner_b(e);
                    const double w1 = 1.0 / (MC_EPS + fabs(v[c1]));
                    const double w2 = 1.0 / (MC_EPS + fabs(v[c2]));
                    const double ff = w1 + w2;                       // (0 + w1) + w2
                    const int d1 = dir == 0 ? mc_corner_dx(c1) : (dir == 1 ? mc_corner_dy(c1) : mc_corner_dz(c1));
                    // the two weights by position along the edge: low = base voxel, high = base + 1
                    w_lo = d1 ? w2 : w1; w_hi = d1 ? w1 : w2;
                    if (P.step == 1 && ff > 0.0 && isfinite(ff)) {
                        // stp = 1: off the edge axis both corners share the offset o, so
                        // (o*w1 + o*w2)/ff is exactly o; on the axis it is w_hi/ff (one division).
                        const double t = w_hi / ff;
                        pos[0] = dir == 0 ? (float)((double)gx + t) : (float)(xs + mc_corner_dx(c1));
                        pos[1] = dir == 1 ? (float)((double)gy + t) : (float)(ys + mc_corner_dy(c1));
                        pos[2] = dir == 2 ? (float)((double)(gz + P.z0) + t) : (float)(zs + mc_corner_dz(c1));
                    } else {
                        double fx = 0.0, fy = 0.0, fz = 0.0;
                        fx += (double)mc_corner_dx(c1) * w1; fy += (double)mc_corner_dy(c1) * w1; fz += (double)mc_corner_dz(c1) * w1;
                        fx += (double)mc_corner_dx(c2) * w2; fy += (double)mc_corner_dy(c2) * w2; fz += (double)mc_corner_dz(c2) * w2;
                        pos[0] = (float)((double)xs + stp * fx / ff);
                        pos[1] = (float)((double)ys + stp * fy / ff);
                        pos[2] = (float)((double)zs + stp * fz / ff);
                    }
                    if (gather_colors) {
                        float ca[3], cb[3];
                        load_corner_color(P, x, y, z, c1, ca);
                        load_corner_color(P, x, y, z, c2, cb);
                        const float w1f = (float)w1, w2f = (float)w2;
    #pragma unroll
                        for (int jj = 0; jj < 3; jj++) {
                            const float cj = ca[jj] * w1f + cb[jj] * w2f;
                            colr[jj] = (float)((double)cj / ff);
                        }
                    }
                }
                // The cells around the edge in sweep order, ONE at a time -- a loop, not four copies of the search and of the
                // gradient arithmetic (a quarter of the code, fewer live registers): find it, leave it this vertex's id, add its
                // share of the normal.
#pragma unroll 1
                for (int s = 0; s < 4; s++) {
                    const int cx = gx + mc_share_dx(dir, s), cy = gy + mc_share_dy(dir, s), cz = gz + mc_share_dz(dir, s);
                    int sls = -1;   // window slot, or -1 = none, or -2-k = record k via the slow path
                    if (cx == x && cy == y && cz == z) sls = rr;
                    else if (!(cx < 0 || cy < 0 || cz < P.lay_count_begin || cx >= P.ncx || cy >= P.ncy || cz >= P.lay_list_end)) {
                        const int row = (cz - P.lay_count_begin) * P.ncy + cy;
                        int w = -1, k = 0;
                        if (row >= r_f && row - r_f + 1 < nrs) { w = 0; k = row - r_f; }
                        else if (row >= r_f + P.ncy && row - r_f - P.ncy + 1 < nrs) { w = 1; k = row - r_f - P.ncy; }
                        bool in_window = false;
                        if (w >= 0) {
                            const uint32_t ra = s_rs[w][k], rb = s_rs[w][k + 1];   // (relative: see s_rs)
                            const uint32_t wc = w ? w2_cnt : w1_cnt, off = w ? w1_cnt : 0u;
                            // (W1 begins at the chunk's first record, usually in the MIDDLE of row r_f: the only sharers looked
                            // up in that row are the +x neighbours of the chunk's own cells, which lie after them -- the part of
                            // the row before the window cannot hold the cell.  Sending these to the global-memory path cost two
                            // slow wave-iterations per chunk: 6 % of all sharer iterations of the 512^3 sphere.)
                            if ((ra >= 1u || w == 0) && rb >= 1u && rb <= wc + 1u) {
                                in_window = true;
                                uint32_t lo = (ra >= 1u ? ra - 1u : 0u) + off, hi = rb - 1u + off;
                                while (hi - lo > 4u) {
                                    const uint32_t mid = (lo + hi) >> 1;
                                    if ((int)(s_wxy[mid] & P.xmask) <= cx) lo = mid; else hi = mid;
                                }
                                // (four independent probes: one LDS round trip instead of a dependent scan)
#pragma unroll
                                for (uint32_t q = 0; q < 4u; q++) {
                                    const uint32_t idx = min(lo + q, (uint32_t)K4_WMAX - 1u);
                                    const int wx = (int)(s_wxy[idx] & P.xmask);
                                    if (lo + q < hi && wx == cx) sls = (int)idx;
                                }
                            }
                        }
                        if (!in_window) {
                            const int g = find_record(P, n, cx, cy, cz);
                            if (g >= 0) sls = -2 - g;
                        }
                    }
                    if (sls == -1) continue;
                    const int es = mc_share_edge(dir, s);
                    const uint32_t g = sls >= 0 ? ((uint32_t)sls < w1_cnt ? w1_start + (uint32_t)sls : w2_start + ((uint32_t)sls - w1_cnt)) : (uint32_t)(-2 - sls);
                    const uint32_t ti = sls >= 0 ? s_winfo[sls] : P.rec_info[g];   // (sls < 0: outside the staged windows, through global memory)
                    const uint64_t tocc = ti ? s_occ[ti >> 22] : 0ull;
                    // the sharer's slot for this vertex id (only cells that emit triangles have slots: K5 reads only its own chunk's block)
                    if ((ti >> 14) & 15u) P.rec_vid[(size_t)(g / MC_CHUNK) * MC_VSTRIDE + (ti & 0xfffu) + mc_slot_of(tocc, es)] = vi;
                    if (!emit) continue;
                    const int occ = (int)((tocc >> (4 * es)) & 15ull);
                    if (occ) {
                        if (sls >= 0) add_sharer_gradients(CornersLdsT<ISO0>{s_wc + sls, K4_WMAX, iso}, es, occ, dir, w_lo, w_hi, nrm);
                        else add_sharer_gradients(CornersGlobal{P.rec_corners + (size_t)g * 8, iso}, es, occ, dir, w_lo, w_hi, nrm);
                    }
                }
                if (!emit) continue;
            }
            // Cell.NegativeNormals (Cell.cs:97-109), then Mesh.Transform (Mesh.cs:47-64)
            const float len = v3len(nrm[0], nrm[1], nrm[2]);
            const float q0 = -(nrm[0] / len), q1 = -(nrm[1] / len), q2 = -(nrm[2] / len);
            const float t0 = q0 * M.inv[0], t1 = q1 * M.inv[1], t2 = q2 * M.inv[2];
            const float tl = v3len(t0, t1, t2);
            const float px = pos[0] * M.sc[0] + M.tr[0];
            const float py = pos[1] * M.sc[1] + M.tr[1];
            const float pz = pos[2] * M.sc[2] + M.tr[2];
            float* ov = M.vertices + (size_t)out * 3;
            float* on = M.normals + (size_t)out * 3;
            // (one 12-byte store per array: a third of the store instructions of three dword stores)
            *reinterpret_cast<f3u*>(ov) = f3u{px, py, pz};
            if (M.vdesc)    // colours by re-evaluation: sdfk_vertex_colors needs the creator cell and the edge
                M.vdesc[out] = make_uint2(base + (uint32_t)rr, (uint32_t)e);
            else if (M.colors)   // (null: a slab payload without a colour section -- the volume has no colours, they are all zero)
                *reinterpret_cast<f3u*>(M.colors + (size_t)out * 3) = f3u{colr[0], colr[1], colr[2]};
            *reinterpret_cast<f3u*>(on) = f3u{t0 / tl, t1 / tl, t2 / tl};
            if (M.grid_vertices) {
                float* og = M.grid_vertices + (size_t)out * 3;
                *reinterpret_cast<f3u*>(og) = f3u{pos[0], pos[1], pos[2]};
            }
            bmin[0] = fminf(bmin[0], px); bmin[1] = fminf(bmin[1], py); bmin[2] = fminf(bmin[2], pz);
            bmax[0] = fmaxf(bmax[0], px); bmax[1] = fmaxf(bmax[1], py); bmax[2] = fmaxf(bmax[2], pz);
        }
    }
    // per-workgroup AABB partials (Mesh.Measure, Mesh.cs:30-45), reduced by K5's first workgroup
    float r[6] = {bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2]};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], __shfl_down(r[j], o));
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], __shfl_down(r[j], o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_red[j][wave] = r[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        float a = s_red[j][0];
        for (int w = 1; w < 4; w++) a = (j < 3) ? fminf(a, s_red[j][w]) : fmaxf(a, s_red[j][w]);
        M.bounds_partial[(size_t)blockIdx.x * 6 + j] = a;
    }
}

// A kernel that does nothing for `ticks` of the 100 MHz wall clock: the probe of the stream-placement measurement
// (lib_context.hip, "stream placement").
__global__ __launch_bounds__(64) void k_spin(int ticks, int* sink)
{
    const long long t0 = wall_clock64();
    int n = 0;
    while (wall_clock64() - t0 < ticks) n++;
    if (sink && n < 0) *sink = n;
}

// Mesh.Measure (Mesh.cs:30-45): reduce the per-workgroup AABB partials of K4.  Run by one
// workgroup of K5 (K4 is complete by then), so it costs no launch of its own.
__device__ __forceinline__ void reduce_bounds(const McMeshOut& M, float* s_bounds /*[6], LDS*/)
{
    __shared__ float s_red[6][4];
    float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < M.bounds_blocks; b += 256) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], M.bounds_partial[(size_t)b * 6 + j]);
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], M.bounds_partial[(size_t)b * 6 + j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], __shfl_down(r[j], o));
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], __shfl_down(r[j], o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_red[j][wave] = r[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        float a = s_red[j][0];
        for (int w = 1; w < 4; w++) a = (j < 3) ? fminf(a, s_red[j][w]) : fmaxf(a, s_red[j][w]);
        M.bounds[j] = a;
        M.host_bounds[j] = a;
        s_bounds[j] = a;
    }
    __syncthreads();
}

// ---- Z-slab payload header (sdfkit_amd/dist.py; SDFK_SLAB_HEADER_BYTES) -------------------------------------
// cap_v: vertex slots each of the V / (C) / N sections is laid out for -- the sections start at 64, 64 + 12 cap_v, ...,
// the indices at 64 + vbytes * cap_v.  0 (or nv) = dense.  A step that emits straight into the send buffer lays the
// sections out for the CAPACITIES it guessed, before it knows the counts.
// idx_bits: 0 (or 32) = Triangles as int32; 16 = the compact form k_payload_compact writes (uint16 offsets against one int32
// base per block of 1024 indices, the bases after the offsets).  flags bit 0: the 16-bit form did not fit (some block of
// indices spans more than 65535 ids): the step is redone, the session goes back to int32 indices.

// ---------------------------------------------------------------------------
// K5: triangles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_triangles(McParams P, McMeshOut M)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_lut[MCSLOT_BYTES];   // the SLOT form of the triangle rows, 4 bits an entry (mc_device.h)
    __shared__ uint16_t s_rowoff[MCLUT_NROWS];
    __shared__ uint32_t s_vid[K5_VMAX];    // the chunk's block of vertex ids (what lies beyond K5_VMAX slots is read from global memory)
    __shared__ uint32_t s_pre[257];
    __shared__ uint16_t s_lo[256], s_sp[256];   // per record: start of its triangle row in the blob, first of its slots in the block
    __shared__ uint8_t s_cell[256 * 12];   // record (slot in the chunk) of each triangle of the chunk (<= 12 triangles a cell)
    mc_load_slotlut_to_lds(s_lut, s_rowoff);
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const uint32_t nghost = P.counters->nghost;
    if (blockIdx.x == 0) {   // K4 has completed (stream order): finish Mesh.Measure
        __shared__ float s_bounds[6];
        reduce_bounds(M, s_bounds);
        // A sharded step that emits straight into its all-gather send buffer (the mesh arrays ARE sections of the
        // payload, laid out for the capacities): the payload header, from the job's own counters -- counts, or
        // -1/-1 when a speculative capacity was too small (every rank then redoes the step on the exact path).
        if (M.slab_header && threadIdx.x == 0) {
            const McCounters c = *P.counters;
            const bool ok = c.n_active <= P.cap_active && (uint64_t)(c.total_v - c.nghost) <= (uint64_t)M.cap_vertices &&
                            (uint64_t)c.total_t * 3u <= (uint64_t)M.cap_indices;
            SlabHeader h;
            h.nv = ok ? (int64_t)(c.total_v - c.nghost) : -1;
            h.ni = ok ? (int64_t)c.total_t * 3 : -1;
            h.vbytes = M.slab_vbytes;
            h.cap_v = (int32_t)M.cap_vertices;
            for (int k = 0; k < 3; k++) { h.bmin[k] = h.nv > 0 ? s_bounds[k] : 0.0f; h.bmax[k] = h.nv > 0 ? s_bounds[3 + k] : 0.0f; }
            h.idx_bits = 0; h.flags = 0; h.pad[0] = h.pad[1] = 0.0f;
            *reinterpret_cast<SlabHeader*>(M.slab_header) = h;
        }
    }
    __syncthreads();   // (s_rowoff is read below)
    for (uint32_t base = first_chunk_of_block<SDFK_KT_XCD_GROUP>() * MC_CHUNK; base < n; base += gridDim.x * MC_CHUNK) {
        const uint32_t irec = base + threadIdx.x;
        const uint32_t ci = base / MC_CHUNK;
        uint32_t my_ni = 0;
        __syncthreads();
        const uint64_t chunk_pre = P.chunkpre[ci];   // left by k_vertices (loaded with the records: not a round trip of its own after the scan)
        // (in-chunk prefix and chunk total of the triangles: left by k_resolve, as in k_vertices -- no workgroup scan)
        const uint32_t total = 3u * (uint32_t)(P.chunktot[ci] & 0x7fffffffull);
        // the chunk's block of vertex ids, pushed by the creators in k_vertices: contiguous, loaded with coalesced loads (all of a
        // lane's loads are issued before the first LDS store)
        const uint32_t nvid = min(P.chunkslots[ci], (uint32_t)K5_VMAX);
        const uint32_t* vblock = P.rec_vid + (size_t)ci * MC_VSTRIDE;
        uint32_t vv[K5_VMAX / 256];
#pragma unroll
        for (int q = 0; q < K5_VMAX / 256; q++) {
            const uint32_t k = threadIdx.x + 256u * q;
            vv[q] = k < nvid ? vblock[k] : 0u;
        }
        uint32_t t0 = 0, info = 0;
        if (threadIdx.x < MC_CHUNK && irec < n) {
            info = P.rec_info[irec];
            my_ni = 3u * ((info >> 14) & 15u);
            t0 = P.rec_pre[irec] >> 16;
        }
#pragma unroll
        for (int q = 0; q < K5_VMAX / 256; q++) s_vid[threadIdx.x + 256u * q] = vv[q];
        s_lo[threadIdx.x] = s_rowoff[min(info >> 22, (uint32_t)MCLUT_NROWS - 1u)];
        s_sp[threadIdx.x] = (uint16_t)(info & 0xfffu);
        s_pre[threadIdx.x] = (threadIdx.x < MC_CHUNK && irec < n) ? 3u * t0 : total;   // (the prefix counts indices)
        if (threadIdx.x == 255) s_pre[256] = total;
        // triangle -> cell table: one LDS read per index instead of a search over the prefix
        for (uint32_t k = 0; 3u * k < my_ni; k++) s_cell[t0 + k] = (uint8_t)threadIdx.x;
        __syncthreads();
        const size_t chunk_ibase = (size_t)(chunk_pre & 0x7fffffffull) * 3;
        for (uint32_t j = threadIdx.x; j < total; j += 256u) {   // one lane per triangle index
            const size_t o = chunk_ibase + j;   // serial position of this triangle index
            if (o >= M.cap_indices) { P.host_counters->overflow = 1u; continue; }
            const int rr = (int)s_cell[j / 3u];
            const uint32_t k = j - s_pre[rr];
            const uint32_t slot = (uint32_t)s_sp[rr] + mc_slot_entry(s_lut, (uint32_t)s_lo[rr] + k);   // the record's slot of this triangle corner's vertex id
            const uint32_t vi = slot < (uint32_t)K5_VMAX ? s_vid[slot] : vblock[slot];             // pushed by the creator (K4)
            M.triangles[o] = (int32_t)((int64_t)vi - (int64_t)nghost + M.vertex_base);
        }
    }
}

// ---------------------------------------------------------------------------
// auxiliary volume kernels
// ---------------------------------------------------------------------------
// Voxels.ClipToBounds, Voxels.cs:133-167 (all six faces get Size.X/NX): one lane per element of
// the largest face, two opposite faces per axis (the z faces only where the slab holds them).  Rows are `pitch` long.
__global__ __launch_bounds__(256) void k_clip(float* __restrict__ values, int nx, int ny, int nz, int pitch, int z0, int nz_global, float outside)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t sx = (size_t)ny * pitch;
    if (i < (size_t)ny * nz) {                      // x = 0 and x = nx-1
        const size_t y = i / nz, z = i % nz;
        values[y * pitch + z] = outside;
        values[(size_t)(nx - 1) * sx + y * pitch + z] = outside;
    }
    if (i < (size_t)nx * nz) {                      // y = 0 and y = ny-1
        const size_t x = i / nz, z = i % nz;
        values[x * sx + z] = outside;
        values[x * sx + (size_t)(ny - 1) * pitch + z] = outside;
    }
    if (i < (size_t)nx * ny) {                      // global z = 0 and z = nz_global-1
        if (z0 == 0) values[i * pitch] = outside;
        if (z0 + nz == nz_global) values[i * pitch + (nz - 1)] = outside;
    }
}

// The same on the cached sign bits (bits[z][y][xw]): boundary voxels all become `bit`
// (= outside > iso); one lane per (z, y) row.  Bits of x >= nx stay 0.
__global__ __launch_bounds__(256) void k_clip_bits(uint64_t* __restrict__ bits, int nx, int ny, int nz, int z0, int nz_global, int nxw, int bit)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)nz * ny) return;
    const int z = (int)(i / ny), y = (int)(i % ny), zg = z + z0;
    uint64_t* row = bits + i * nxw;
    const int last = nxw - 1, lastbit = (nx - 1) & 63;
    const uint64_t lastvalid = lastbit == 63 ? ~0ull : ((1ull << (lastbit + 1)) - 1ull);
    if (y == 0 || y == ny - 1 || zg == 0 || zg == nz_global - 1) {
        for (int w = 0; w < nxw; w++) row[w] = bit ? (w == last ? lastvalid : ~0ull) : 0ull;
    } else {
        uint64_t w0 = row[0];
        w0 = bit ? (w0 | 1ull) : (w0 & ~1ull);
        if (last == 0) w0 = bit ? (w0 | (1ull << lastbit)) : (w0 & ~(1ull << lastbit));
        row[0] = w0;
        if (last > 0) {
            const uint64_t wl = row[last];
            row[last] = bit ? (wl | (1ull << lastbit)) : (wl & ~(1ull << lastbit));
        }
    }
}

// ---- Z-slab exchange helpers (sdfkit_amd/dist.py) ------------------------------------------
// vbytes = bytes per vertex in the payload: 36 (V, C, N) or 24 (V, N: colours are all zero and left out)
// (struct SlabHeader: above k_triangles, whose first workgroup writes it when a step emits into its send buffer)


__global__ void k_slab_header(SlabHeader* dst, int64_t nv, int64_t ni, const float* __restrict__ bounds, int vbytes)
{
    if (threadIdx.x == 0) {
        SlabHeader h;
        h.nv = nv; h.ni = ni; h.vbytes = vbytes; h.cap_v = 0;
        for (int k = 0; k < 3; k++) { h.bmin[k] = nv ? bounds[k] : 0.0f; h.bmax[k] = nv ? bounds[3 + k] : 0.0f; }
        h.idx_bits = 0; h.flags = 0; h.pad[0] = h.pad[1] = 0.0f;
        *dst = h;
    }
}

// Packing of a mesh whose job is still queued (no host knowledge of the counts): header and
// arrays are written by the device from the job's counters.  A speculative job whose buffers
// turned out too small leaves nv = ni = -1 in the header (the host redoes that step); a
// payload that does not fit `capacity` leaves the header only.

__global__ __launch_bounds__(256) void k_pack_pending(PackArgs A)
{
    const McCounters c = *A.counters;
    const bool ok = c.n_active <= A.cap_active && c.overflow == 0 && (uint64_t)(c.total_v - c.nghost) <= (uint64_t)A.cap_v &&
                    (uint64_t)c.total_t * 3u <= A.cap_i;
    const int64_t nv = ok ? (int64_t)(c.total_v - c.nghost) : -1, ni = ok ? (int64_t)c.total_t * 3 : -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SlabHeader h;
        h.nv = nv; h.ni = ni; h.vbytes = A.vbytes; h.cap_v = 0;
        for (int k = 0; k < 3; k++) { h.bmin[k] = nv > 0 ? A.bounds[k] : 0.0f; h.bmax[k] = nv > 0 ? A.bounds[3 + k] : 0.0f; }
        h.idx_bits = 0; h.flags = 0; h.pad[0] = h.pad[1] = 0.0f;
        *reinterpret_cast<SlabHeader*>(A.dst) = h;
    }
    if (!ok || (int64_t)sizeof(SlabHeader) + A.vbytes * nv + 4 * ni > A.capacity) return;
    // [V | (C) | N | T] as one run of 4-byte words
    const bool wc = A.vbytes == 36;
    const int64_t nf = nv * 3, c0 = nf, n0 = wc ? 2 * nf : nf, t0 = n0 + nf, total = t0 + ni;
    uint32_t* out = reinterpret_cast<uint32_t*>(A.dst + sizeof(SlabHeader));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        uint32_t w;
        if (i < nf) w = __float_as_uint(A.vertices[i]);
        else if (i < n0) w = __float_as_uint(A.colors[i - c0]);
        else if (i < t0) w = __float_as_uint(A.normals[i - n0]);
        else w = (uint32_t)A.triangles[i - t0];
        out[i] = w;
    }
}

// The slab payload in its compact form: V / (C) / N dense, the indices as uint16 offsets against one int32 base per block of
// SLAB_IDX_BLOCK indices (a triangle references vertices its own cell, the previous row or the previous layer created:
// within 1024 consecutive indices the ids span about one layer's worth of vertices), the bases after the offsets:
//   [ header | V 12 nv | (C 12 nv) | N 12 nv | T16 2 ni (padded to 4) | bases 4 ceil(ni / 1024) ]
// 48 -> 36 bytes per vertex of a colourless mesh: what every rank has to RECEIVE from every other rank per step.
// src = a plain payload (header + sections laid out for cap_v, indices int32), dst = the rank's section of the gather buffer.
// The header goes last, by the block that finishes last (one 64-bit atomic carries both the arrival count and "some block
// did not fit 16 bits"): counts, idx_bits = 16 -- or nv = ni = -1 with flags bit 0 when a block's ids span more than 65535.
__device__ __forceinline__ int64_t slab_compact_bytes(int64_t nv, int64_t ni, int vbytes)
{
    return (int64_t)sizeof(SlabHeader) + (int64_t)vbytes * nv + ((2 * ni + 3) & ~int64_t(3)) + 4 * ((ni + SLAB_IDX_BLOCK - 1) / SLAB_IDX_BLOCK);
}

__global__ __launch_bounds__(256) void k_payload_compact(const char* __restrict__ src, char* __restrict__ dst, int64_t dst_capacity,
                                                         unsigned long long* __restrict__ ticket)
{
    __shared__ int s_min[4], s_max[4];
    __shared__ unsigned long long s_old;
    const SlabHeader hs = *reinterpret_cast<const SlabHeader*>(src);
    const int64_t nv = hs.nv, ni = hs.ni;
    const bool have = nv >= 0 && ni >= 0 && slab_compact_bytes(nv, ni, hs.vbytes) <= dst_capacity;
    bool overflow = false;
    if (have) {
        const int64_t capv = hs.cap_v > 0 ? (int64_t)hs.cap_v : nv;
        const int nsec = hs.vbytes / 12;                       // V, (C), N
        const int64_t nf = 3 * nv;
        const uint32_t* sv = reinterpret_cast<const uint32_t*>(src + sizeof(SlabHeader));
        uint32_t* dv = reinterpret_cast<uint32_t*>(dst + sizeof(SlabHeader));
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nf * nsec; i += (int64_t)gridDim.x * 256) {
            const int64_t sec = i / nf, o = i - sec * nf;
            dv[i] = sv[sec * 3 * capv + o];
        }
        const int32_t* st = reinterpret_cast<const int32_t*>(src + sizeof(SlabHeader) + (int64_t)hs.vbytes * capv);
        uint16_t* t16 = reinterpret_cast<uint16_t*>(dst + sizeof(SlabHeader) + (int64_t)hs.vbytes * nv);
        int32_t* bases = reinterpret_cast<int32_t*>(dst + sizeof(SlabHeader) + (int64_t)hs.vbytes * nv + ((2 * ni + 3) & ~int64_t(3)));
        const int64_t nblk = (ni + SLAB_IDX_BLOCK - 1) / SLAB_IDX_BLOCK;
        for (int64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
            int32_t v[4];
            int lo = INT_MAX, hi = INT_MIN;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t i = b * SLAB_IDX_BLOCK + threadIdx.x + 256 * q;
                v[q] = i < ni ? st[i] : 0;
                if (i < ni) { lo = min(lo, v[q]); hi = max(hi, v[q]); }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
            __syncthreads();   // (s_min / s_max of the previous block are no longer read)
            if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lo; s_max[threadIdx.x >> 6] = hi; }
            __syncthreads();
            lo = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));
            hi = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
            if ((int64_t)hi - (int64_t)lo > 65535) overflow = true;
            if (threadIdx.x == 0) bases[b] = lo;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int64_t i = b * SLAB_IDX_BLOCK + threadIdx.x + 256 * q;
                if (i < ni) t16[i] = (uint16_t)(uint32_t)(v[q] - lo);
            }
        }
    }
    // the header, by the block that arrives last
    if (threadIdx.x == 0) s_old = atomicAdd(ticket, 1ull + (overflow ? (1ull << 32) : 0ull));
    __syncthreads();
    if ((uint32_t)s_old != gridDim.x - 1u) return;
    if (threadIdx.x == 0) {
        const bool any_overflow = overflow || (s_old >> 32) != 0;
        SlabHeader h = hs;
        h.cap_v = 0;
        h.pad[0] = h.pad[1] = 0.0f;
        if (nv >= 0 && ni >= 0 && !any_overflow) { h.idx_bits = 16; h.flags = 0; }
        else if (any_overflow) { h.nv = -1; h.ni = -1; h.idx_bits = 16; h.flags = 1; }
        else { h.idx_bits = 16; h.flags = 0; }   // (the step's own -1 / -1: passed on)
        *reinterpret_cast<SlabHeader*>(dst) = h;
        *ticket = 0ull;                            // ready for the next launch (stream order)
    }
}

// slab r of the gathered buffer: indices += sum of the vertex counts of slabs 0..r-1
// (nothing is touched when a header is marked invalid: that step is redone by the host)
// (mirror_only: a rank that received headers but no foreign payloads -- gather-to-root exchange -- has nothing to rebase)
__global__ __launch_bounds__(256) void k_slabs_rebase(char* __restrict__ gathered, int world, int64_t stride, SlabHeader* mirror, int mirror_only)
{
    const int r = blockIdx.y;
    if (mirror && blockIdx.x == 0 && threadIdx.x < sizeof(SlabHeader) / 4)   // header r -> the host's (mapped) copy
        reinterpret_cast<uint32_t*>(mirror + r)[threadIdx.x] =
            reinterpret_cast<const uint32_t*>(gathered + (size_t)r * stride)[threadIdx.x];
    if (mirror_only) return;
    int64_t base = 0;
    for (int q = 0; q < world; q++) {
        const int64_t nvq = reinterpret_cast<const SlabHeader*>(gathered + (size_t)q * stride)->nv;
        if (nvq < 0) return;
        if (q < r) base += nvq;
    }
    if (r == 0 || base == 0) return;
    const SlabHeader* h = reinterpret_cast<const SlabHeader*>(gathered + (size_t)r * stride);
    if (h->idx_bits == 16) return;   // (16-bit offsets are decoded -- and rebased -- by whoever extracts the mesh: k_slabs_concat)
    const int64_t capv = h->cap_v > 0 ? (int64_t)h->cap_v : h->nv;
    if ((int64_t)sizeof(SlabHeader) + h->vbytes * capv + 4 * h->ni > stride) return;   // header-only payload
    int32_t* t = reinterpret_cast<int32_t*>(gathered + (size_t)r * stride + sizeof(SlabHeader) + (size_t)capv * h->vbytes);
    const int64_t n = h->ni;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) t[i] += (int32_t)base;
}

// Mesh.Transform(Matrix4x4), Mesh.cs:47-64, on the device arrays: Vector3.Transform for the positions, Vector3.TransformNormal
// with the normal matrix the HOST derived (Matrix4x4.Invert / Transpose are BCL calls of the reference: the shim makes them
// with the BCL itself) + Vector3.Normalize for the normals; per-workgroup AABB partials for Mesh.Measure.  Row-vector
// convention, products summed left to right, no contraction.

__global__ __launch_bounds__(256) void k_mesh_transform(XformArgs A)
{
    __shared__ float s_red[6][4];
    float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * 256) {
        const f3u v = *reinterpret_cast<const f3u*>(A.vertices + i * 3);
        const f3u q = *reinterpret_cast<const f3u*>(A.normals + i * 3);
        const float px = ((v.x * A.m[0] + v.y * A.m[4]) + v.z * A.m[8]) + A.m[12];
        const float py = ((v.x * A.m[1] + v.y * A.m[5]) + v.z * A.m[9]) + A.m[13];
        const float pz = ((v.x * A.m[2] + v.y * A.m[6]) + v.z * A.m[10]) + A.m[14];
        const float tx = (q.x * A.nm[0] + q.y * A.nm[4]) + q.z * A.nm[8];
        const float ty = (q.x * A.nm[1] + q.y * A.nm[5]) + q.z * A.nm[9];
        const float tz = (q.x * A.nm[2] + q.y * A.nm[6]) + q.z * A.nm[10];
        const float len = v3len(tx, ty, tz);
        *reinterpret_cast<f3u*>(A.vertices + i * 3) = f3u{px, py, pz};
        *reinterpret_cast<f3u*>(A.normals + i * 3) = f3u{tx / len, ty / len, tz / len};
        r[0] = fminf(r[0], px); r[1] = fminf(r[1], py); r[2] = fminf(r[2], pz);
        r[3] = fmaxf(r[3], px); r[4] = fmaxf(r[4], py); r[5] = fmaxf(r[5], pz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], __shfl_down(r[j], o));
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], __shfl_down(r[j], o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_red[j][wave] = r[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        float a = s_red[j][0];
        for (int w = 1; w < 4; w++) a = (j < 3) ? fminf(a, s_red[j][w]) : fmaxf(a, s_red[j][w]);
        A.partial[(size_t)blockIdx.x * 6 + j] = a;
    }
}

__global__ __launch_bounds__(256) void k_bounds_reduce(const float* __restrict__ partial, int blocks, float* __restrict__ bounds)
{
    __shared__ float s_red[6][4];
    float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < blocks; b += 256) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], partial[(size_t)b * 6 + j]);
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], partial[(size_t)b * 6 + j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], __shfl_down(r[j], o));
#pragma unroll
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], __shfl_down(r[j], o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_red[j][wave] = r[j];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        float a = s_red[j][0];
        for (int w = 1; w < 4; w++) a = (j < 3) ? fminf(a, s_red[j][w]) : fmaxf(a, s_red[j][w]);
        bounds[j] = a;
    }
}

// Gathered COMPACT slabs (idx_bits = 16) -> the whole mesh's int32 index array: out[sum of ni of slabs 0..r-1 + k] = base of k's
// block + offset + sum of nv of slabs 0..r-1.  The counterpart of k_slabs_rebase for the compact form -- after it every rank
// holds directly usable global indices (V / N stay where the exchange left them) --, and like it the kernel mirrors the
// `world` headers into pinned host memory.  Nothing is decoded when a header is marked invalid (that step is redone).
__global__ __launch_bounds__(256) void k_slabs_decode16(const char* __restrict__ gathered, int world, int64_t stride, SlabHeader* mirror,
                                                        int32_t* __restrict__ out, int64_t out_capacity)
{
    const int r = blockIdx.y;
    if (mirror && blockIdx.x == 0 && threadIdx.x < sizeof(SlabHeader) / 4)
        reinterpret_cast<uint32_t*>(mirror + r)[threadIdx.x] =
            reinterpret_cast<const uint32_t*>(gathered + (size_t)r * stride)[threadIdx.x];
    int64_t vbase = 0, ibase = 0;
    for (int q = 0; q < world; q++) {
        const SlabHeader* hq = reinterpret_cast<const SlabHeader*>(gathered + (size_t)q * stride);
        if (hq->nv < 0 || hq->ni < 0) return;
        if (q < r) { vbase += hq->nv; ibase += hq->ni; }
    }
    const SlabHeader* h = reinterpret_cast<const SlabHeader*>(gathered + (size_t)r * stride);
    if (h->idx_bits != 16 || slab_compact_bytes(h->nv, h->ni, h->vbytes) > stride) return;
    const int64_t ni = h->ni;
    if (ibase + ni > out_capacity) return;
    const char* sec = gathered + (size_t)r * stride + sizeof(SlabHeader) + (int64_t)h->vbytes * h->nv;
    const uint16_t* t16 = reinterpret_cast<const uint16_t*>(sec);
    const int32_t* bases = reinterpret_cast<const int32_t*>(sec + ((2 * ni + 3) & ~int64_t(3)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ni; i += (int64_t)gridDim.x * 256)
        out[ibase + i] = bases[i / SLAB_IDX_BLOCK] + (int32_t)t16[i] + (int32_t)vbase;
}

// step > 1 (MarchingCubes.cs:49-80): the sweep only ever touches voxels whose indices are
// multiples of `step`; gather them into a dense volume and mesh that with unit cells.  sp / dp = row pitches.
__global__ __launch_bounds__(256) void k_subsample(const float* __restrict__ src, const float* __restrict__ srcc,
                                                   float* __restrict__ dst, float* __restrict__ dstc, int nx, int ny,
                                                   int sp, int mx, int my, int mz, int dp, int step)
{
    const size_t n = (size_t)mx * my * mz;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const int z = (int)(i % mz);
        const size_t t = i / mz;
        const int y = (int)(t % my);
        const int x = (int)(t / my);
        const size_t o = ((size_t)(x * step) * ny + (size_t)(y * step)) * sp + (size_t)(z * step);
        const size_t d = ((size_t)x * my + y) * dp + z;
        dst[d] = src[o];
        if (srcc) { dstc[d * 3] = srcc[o * 3]; dstc[d * 3 + 1] = srcc[o * 3 + 1]; dstc[d * 3 + 2] = srcc[o * 3 + 2]; }
    }
}

// Rows of `w` floats between a dense array (row stride w) and a pitched one (row stride pw >= w): the host side of
// Voxels.Values / Voxels.Colors is dense ([nx][ny][nz], Voxels.cs:8-9), the device volume pads its rows to a
// multiple of 4 voxels.  TO_PITCHED: dense -> pitched; else pitched -> dense.
template <bool TO_PITCHED>
__global__ __launch_bounds__(256) void k_repitch(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw)
{
    const size_t n = rows * (size_t)w;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const size_t r = i / w, c = i - r * w;
        if (TO_PITCHED) dst[r * pw + c] = src[i];
        else dst[i] = src[r * pw + c];
    }
}

// ---- explicit instantiations of the template kernels (declared in mc_kernels.h; launched from lib_*.hip) ---------------------------
template __global__ void k_signbits8<true>(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny, int nz, int nx8, int pitch, float iso);
template __global__ void k_signbits8<false>(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny, int nz, int nx8, int pitch, float iso);
template __global__ void k_compact<true>(McParams P);
template __global__ void k_compact<false>(McParams P);
template __global__ void k_compact_write<true>(McParams P);
template __global__ void k_compact_write<false>(McParams P);
template __global__ void k_vertices<true>(McParams P, McMeshOut M);
template __global__ void k_vertices<false>(McParams P, McMeshOut M);
template __global__ void k_repitch<true>(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw);
template __global__ void k_repitch<false>(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw);

}  // namespace sdfk
