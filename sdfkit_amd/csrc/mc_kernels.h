// mc_kernels.h -- what the HOST side of the library needs of mc_kernels.hip (the marching-cubes pipeline's kernels, compiled as a translation
// unit of its own): the kernels' declarations, the argument structs they take by value, and the compile-time constants the launch
// geometry depends on.  The template kernels are instantiated in mc_kernels.hip (explicit instantiations at its end).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mc_params.h"

namespace sdfk {

// layers of sign planes a k_compact workgroup walks (mc_kernels.hip, "K2": 1 is the product, measured)
#ifndef SDFK_COMPACT_LPB
#define SDFK_COMPACT_LPB 1
#endif
constexpr int K2_LPB = SDFK_COMPACT_LPB;
static_assert(K2_LPB >= 1 && K2_LPB <= 8, "sign planes per workgroup are held in registers");

// ---- Z-slab payload header (sdfkit_amd/dist.py; SDFK_SLAB_HEADER_BYTES) -------------------------------------
// cap_v: vertex slots each of the V / (C) / N sections is laid out for -- the sections start at 64, 64 + 12 cap_v, ...,
// the indices at 64 + vbytes * cap_v.  0 (or nv) = dense.  A step that emits straight into the send buffer lays the
// sections out for the CAPACITIES it guessed, before it knows the counts.
// idx_bits: 0 (or 32) = Triangles as int32; 16 = the compact form k_payload_compact writes (uint16 offsets against one int32
// base per block of 1024 indices, the bases after the offsets).  flags bit 0: the 16-bit form did not fit (some block of
// indices spans more than 65535 ids): the step is redone, the session goes back to int32 indices.
struct SlabHeader { int64_t nv, ni; float bmin[3], bmax[3]; int32_t vbytes; int32_t cap_v; int32_t idx_bits; int32_t flags; float pad[2]; };
constexpr int SLAB_IDX_BLOCK = 1024;
static_assert(sizeof(SlabHeader) == 64, "SDFK_SLAB_HEADER_BYTES");

// Packing of a mesh whose job is still queued (k_pack_pending): header and arrays are written by the device from the job's counters.
struct PackArgs {
    const McCounters* counters;   // of the queued job
    uint32_t cap_active, cap_v;
    uint64_t cap_i;
    const float* vertices;
    const float* colors;
    const float* normals;
    const int32_t* triangles;
    const float* bounds;          // device float[6], written by k_triangles
    char* dst;
    int64_t capacity;
    int vbytes;                   // 36, or 24 = colours left out
};

// Mesh.Transform on the device arrays (k_mesh_transform)
struct XformArgs {
    float* vertices;
    float* normals;
    int64_t n;
    float m[16], nm[16];
    float* partial;   // [grid][6]
};

// ---- the kernels --------------------------------------------------------------------------------------------------------------------
template <bool FLAT> __global__ void k_signbits8(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny, int nz, int nx8, int pitch, float iso);
extern template __global__ void k_signbits8<true>(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny, int nz, int nx8, int pitch, float iso);
extern template __global__ void k_signbits8<false>(const float* __restrict__ values, uint8_t* __restrict__ bits8, int nx, int ny, int nz, int nx8, int pitch, float iso);
__global__ void k_bits_transpose(const uint8_t* __restrict__ bits8, uint64_t* __restrict__ bits, int nx8, int ny, int nz, int nxw, int pitch8);
template <bool WRITE> __global__ void k_compact(McParams P);
extern template __global__ void k_compact<true>(McParams P);
extern template __global__ void k_compact<false>(McParams P);
template <bool MASKS> __global__ void k_compact_write(McParams P);
extern template __global__ void k_compact_write<true>(McParams P);
extern template __global__ void k_compact_write<false>(McParams P);
__global__ void k_blockscan(McParams P);
__global__ void k_publish(McParams P);
__global__ void k_chunkscan(McParams P);
__global__ void k_gather_corners(McParams P);
__global__ void k_resolve(McParams P);
template <bool ISO0> __global__ void k_vertices(McParams P, McMeshOut M);
extern template __global__ void k_vertices<true>(McParams P, McMeshOut M);
extern template __global__ void k_vertices<false>(McParams P, McMeshOut M);
__global__ void k_spin(int ticks, int* sink);
__global__ void k_triangles(McParams P, McMeshOut M);
__global__ void k_clip(float* __restrict__ values, int nx, int ny, int nz, int pitch, int z0, int nz_global, float outside);
__global__ void k_clip_bits(uint64_t* __restrict__ bits, int nx, int ny, int nz, int z0, int nz_global, int nxw, int bit);
__global__ void k_slab_header(SlabHeader* dst, int64_t nv, int64_t ni, const float* __restrict__ bounds, int vbytes);
__global__ void k_pack_pending(PackArgs A);
__global__ void k_payload_compact(const char* __restrict__ src, char* __restrict__ dst, int64_t dst_capacity, unsigned long long* __restrict__ ticket);
__global__ void k_slabs_rebase(char* __restrict__ gathered, int world, int64_t stride, SlabHeader* mirror, int mirror_only);
__global__ void k_mesh_transform(XformArgs A);
__global__ void k_bounds_reduce(const float* __restrict__ partial, int blocks, float* __restrict__ bounds);
__global__ void k_slabs_decode16(const char* __restrict__ gathered, int world, int64_t stride, SlabHeader* mirror, int32_t* __restrict__ out, int64_t out_capacity);
__global__ void k_subsample(const float* __restrict__ src, const float* __restrict__ srcc, float* __restrict__ dst, float* __restrict__ dstc, int nx, int ny, int sp, int mx, int my, int mz, int dp, int step);
template <bool TO_PITCHED> __global__ void k_repitch(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw);
extern template __global__ void k_repitch<true>(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw);
extern template __global__ void k_repitch<false>(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int w, int pw);

}  // namespace sdfk
