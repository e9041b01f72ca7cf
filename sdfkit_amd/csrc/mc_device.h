// mc_device.h -- device-side Lewiner marching-cubes decisions and vertex math for gfx950.
//
// Everything here is per-cell arithmetic on the eight corner values of one cube, in
// double precision exactly like the reference (Cell.cs:74,191-208: float voxels are
// widened, the iso value is subtracted in double).  The translation unit is compiled
// with -ffp-contract=off: `A*C - B*D` must round as two multiplies and a subtract.
//
// Corner order v0..v7 (Luts.cs:30-52): v0=(x,y,z) v1=(x+1,y,z) v2=(x+1,y+1,z) v3=(x,y+1,z),
// v4..v7 the same at z+1.  "Bit order" index = dz*4+dy*2+dx (Cell.cs:318-319).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_decide.h"

namespace sdfk {

// ---- the table blobs into LDS -------------------------------------------------------------------------------------------
// 256-thread workgroups only.  All loads are issued before the first LDS store: a plain
// "load; store" loop is not unrolled by the compiler and costs one L2 round trip per trip.
__device__ __forceinline__ void mc_load_lut_to_lds(int8_t* s_lut /* MCLUT_PADDED bytes, 16-byte aligned */)
{
    constexpr int N = MCLUT_PADDED / 16;
    static_assert(N > 768 && N <= 1024, "copy below assumes 3 full rounds + 1 partial of 256 lanes");
    const uint4* src = reinterpret_cast<const uint4*>(c_lut);
    uint4* dst = reinterpret_cast<uint4*>(s_lut);
    const int t = (int)threadIdx.x;
    const uint4 r0 = src[t], r1 = src[t + 256], r2 = src[t + 512];
    const uint4 r3 = src[min(t + 768, N - 1)];
    dst[t] = r0; dst[t + 256] = r1; dst[t + 512] = r2;
    if (t + 768 < N) dst[t + 768] = r3;
}

// 256-thread workgroups only
__device__ __forceinline__ void mc_load_dec_to_lds(int8_t* s_dec /* MCDEC_PADDED bytes, 16-byte aligned */)
{
    constexpr int N = MCDEC_PADDED / 16;
    static_assert(N <= 256, "one round of 256 lanes");
    if ((int)threadIdx.x < N) reinterpret_cast<uint4*>(s_dec)[threadIdx.x] = reinterpret_cast<const uint4*>(c_dec.v)[threadIdx.x];
}

// ---- vertex-id SLOTS ------------------------------------------------------------------------------------------------------
// The ids of the vertices a cell references travel from k_vertices (the creators push them) to k_triangles through ONE compact
// block per chunk of MC_CHUNK records: a record that emits triangles owns one slot per DISTINCT vertex id of its tiling row, in
// increasing id order (slot of id e = number of used ids below e), at [chunk * MC_VSTRIDE + in-chunk slot prefix + slot].  A
// smooth surface uses 3-4 ids per cell: ~14 bytes per record, contiguous per chunk -- k_triangles loads its chunk's block with
// coalesced loads -- instead of thirteen sparse [edge][record] planes (4-byte pushes into 32-byte sectors, 4-byte gathers back).
// c_slotlut = the triangle rows of the lookup blob with every entry replaced by its slot: what k_triangles indexes with.
// Built at compile time from the same values as c_lut; the used-id sets must be those of c_rowocc (checked below).
// (4 bits an entry -- a slot is at most 12 --, entry i in nibble i & 1 of byte i >> 1: 6.6 KB of LDS per k_triangles workgroup)
constexpr int MCSLOT_BYTES = ((MCLUT_BLOB_SIZE + 1) / 2 + 15) & ~15;   // copied in 16-byte pieces
struct alignas(16) McSlotBlob { uint8_t v[MCSLOT_BYTES]; };
constexpr McSlotBlob mc_make_slots()
{
    constexpr int8_t full[] = {MCLUT_BLOB_VALUES};
    constexpr uint16_t off[] = {MCLUT_ROWOFF_VALUES};
    constexpr uint8_t nt[] = {MCLUT_ROWNT_VALUES};
    McSlotBlob d{};
    for (int r = 0; r < MCLUT_NROWS; r++) {
        unsigned used = 0;
        for (int k = 0; k < 3 * nt[r]; k++) used |= 1u << full[off[r] + k];
        for (int k = 0; k < 3 * nt[r]; k++) {
            const int i = off[r] + k;
            const unsigned slot = (unsigned)__builtin_popcount(used & ((1u << full[i]) - 1u));
            d.v[i >> 1] = (uint8_t)(d.v[i >> 1] | (slot << (4 * (i & 1))));
        }
    }
    return d;
}
__constant__ McSlotBlob c_slotlut = mc_make_slots();
__device__ __forceinline__ uint32_t mc_slot_entry(const uint8_t* s_slots, uint32_t i) { return ((uint32_t)s_slots[i >> 1] >> (4u * (i & 1u))) & 15u; }
__constant__ uint16_t c_rowoff[MCLUT_NROWS] = {MCLUT_ROWOFF_VALUES};   // start of each triangle row in the blob
constexpr bool mc_rowocc_matches_rows()
{
    constexpr int8_t full[] = {MCLUT_BLOB_VALUES};
    constexpr uint16_t off[] = {MCLUT_ROWOFF_VALUES};
    constexpr uint8_t nt[] = {MCLUT_ROWNT_VALUES};
    constexpr uint64_t occ[] = {MCLUT_ROWOCC_VALUES};
    for (int r = 0; r < MCLUT_NROWS; r++) {
        unsigned used = 0, from_occ = 0;
        for (int k = 0; k < 3 * nt[r]; k++) used |= 1u << full[off[r] + k];
        for (int e = 0; e < 13; e++) from_occ |= ((occ[r] >> (4 * e)) & 15ull) ? 1u << e : 0u;
        if (used != from_occ) return false;
    }
    return true;
}
static_assert(mc_rowocc_matches_rows(), "c_rowocc (k_vertices' view of a row's vertex ids) and the triangle rows (k_triangles') disagree");
// slot of vertex id e in a row whose reference counts are `occ` (c_rowocc: 4 bits per id): the number of used ids below e
__device__ __forceinline__ uint32_t mc_slot_of(uint64_t occ, int e)
{
    const uint64_t nz = (occ | (occ >> 1) | (occ >> 2) | (occ >> 3)) & 0x1111111111111ull;
    return (uint32_t)__popcll(nz & ((1ull << (4 * e)) - 1ull));
}
// 256-thread workgroups only: the slot form of the blob and the row offsets into LDS
__device__ __forceinline__ void mc_load_slotlut_to_lds(uint8_t* s_slots /* MCSLOT_BYTES, 16-byte aligned */, uint16_t* s_rowoff /* MCLUT_NROWS */)
{
    constexpr int N = MCSLOT_BYTES / 16;
    static_assert(N > 256 && N <= 512, "copy below assumes 1 full round + 1 partial of 256 lanes");
    static_assert(MCLUT_NROWS <= 768 && MCLUT_NROWS > 512, "row-offset copy assumes 2 full rounds + 1 partial");
    const uint4* src = reinterpret_cast<const uint4*>(c_slotlut.v);
    uint4* dst = reinterpret_cast<uint4*>(s_slots);
    const int t = (int)threadIdx.x;
    const uint4 r0 = src[t], r1 = src[min(t + 256, N - 1)];
    const uint16_t o0 = c_rowoff[t], o1 = c_rowoff[t + 256], o2 = c_rowoff[min(t + 512, MCLUT_NROWS - 1)];
    dst[t] = r0;
    if (t + 256 < N) dst[t + 256] = r1;
    s_rowoff[t] = o0; s_rowoff[t + 256] = o1;
    if (t + 512 < MCLUT_NROWS) s_rowoff[t + 512] = o2;
}

// ---- edge geometry (pure ALU: tiny tables indexed per lane would each be a dependent
// vector-memory round trip) -------------------------------------------------------------
// Edge e of a cell lies on a grid edge with direction dir (0=X,1=Y,2=Z; 3 = the cell's
// centre vertex) based at voxel (x+ox, y+oy, z+oz) -- the face-layer slot j of Cell.cs:371-441.
__device__ __forceinline__ int mc_edge_dir(int e) { return e < 8 ? (e & 1) : (e < 12 ? 2 : 3); }
__device__ __forceinline__ int mc_edge_ox(int e) { return (0x622 >> e) & 1; }   // e = 1,5,9,10
__device__ __forceinline__ int mc_edge_oy(int e) { return (0xC44 >> e) & 1; }   // e = 2,6,10,11
__device__ __forceinline__ int mc_edge_oz(int e) { return (0x0F0 >> e) & 1; }   // e = 4..7
// The (up to) four cells around a grid edge in sweep order (z outer, y, x inner), s = 0..3,
// as offsets from the base voxel, and the id the edge has inside each of them:
//   X: (0,-1,-1) (0,0,-1) (0,-1,0) (0,0,0)  ids 6,4,2,0
//   Y: (-1,0,-1) (0,0,-1) (-1,0,0) (0,0,0)  ids 5,7,1,3
//   Z: (-1,-1,0) (0,-1,0) (-1,0,0) (0,0,0)  ids 10,11,9,8
__device__ __forceinline__ int mc_share_dx(int dir, int s) { return (dir != 0 && !(s & 1)) ? -1 : 0; }
__device__ __forceinline__ int mc_share_dy(int dir, int s) { return (dir == 0 ? !(s & 1) : (dir == 2 && s < 2)) ? -1 : 0; }
__device__ __forceinline__ int mc_share_dz(int dir, int s) { return (dir < 2 && s < 2) ? -1 : 0; }
__device__ __forceinline__ int mc_share_edge(int dir, int s) { return (int)((0x89ba31750246ull >> (4 * (dir * 4 + s))) & 15ull); }
// bit-order index -> corner number (Cell.cs:453-460: vv[2]=v3, vv[3]=v2, vv[6]=v7, vv[7]=v6)
__device__ __forceinline__ int mc_bit_to_corner(int i) { return i ^ ((i >> 1) & 1); }
// corner k (v0..v7 order) -> voxel offsets
__device__ __forceinline__ int mc_corner_dx(int k) { return ((k + 1) >> 1) & 1; }
__device__ __forceinline__ int mc_corner_dy(int k) { return (k >> 1) & 1; }
__device__ __forceinline__ int mc_corner_dz(int k) { return k >> 2; }

// Corner gradients of Cell.cs:491-498, component j of corner k (corner order):
// vg[k][j] = v[a] - v[b], (a,b) packed 3 bits per corner.
template <class V>
__device__ __forceinline__ double mc_corner_gradient(const V& v, int k, int j)
{
    const unsigned pa = j == 0 ? 0xfe46c0u : (j == 1 ? 0x96c048u : 0x688688u);
    const unsigned pb = j == 0 ? 0xdad489u : (j == 1 ? 0xfb7693u : 0xfacfacu);
    return v[(pa >> (3 * k)) & 7u] - v[(pb >> (3 * k)) & 7u];
}

}  // namespace sdfk
