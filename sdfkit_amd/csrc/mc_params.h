// mc_params.h -- kernel argument blocks shared by mc_kernels.hip and the host driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdfk {

struct McCounters {
    uint32_t n_active;   // active cells found by K2a (may exceed the list capacity)
    uint32_t n_case13;   // cells whose sign word is 0xA5/0x5A (candidates for "impossible 13")
    uint32_t n_dead;     // case-13 cells that resolved to no tiling
    uint32_t total_v;    // vertices created in [lay_count_begin, lay_emit_end)
    uint32_t total_t;    // triangles emitted in [lay_emit_begin, lay_emit_end)
    uint32_t nghost;     // vertices created in [lay_count_begin, lay_emit_begin)
    uint32_t overflow;   // an output capacity was too small
    uint32_t pad;
};

struct McParams {
    const float* values;   // [nx][ny][nz], z fastest (Voxels.cs:8)
    const float* colors;   // [nx][ny][nz][3] or nullptr (= zeros)
    int nx, ny, nz;        // voxel dims of this (slab) volume
    int ncx, ncy, ncz;     // cell dims = n-1
    int nxw;               // 64-bit X words per (z,y) row
    uint32_t nseg;         // ncz*ncy*nxw segments, serial-sweep order
    int z0;                // global z of local plane 0
    int lay_count_begin;   // first local cell layer whose created vertices are counted
    int lay_emit_begin;    // first local cell layer that is emitted
    int lay_emit_end;      // one past the last emitted layer
    float iso;
    int step;              // scale of cell coordinates in vertex positions (Cell.cs:345-347)
    // workspace
    uint64_t* bits;
    uint32_t* segpack;     // per segment: created vertices | triangles << 16
    uint2* segprefix;      // exclusive scan of segpack, (vertices, triangles)
    uint2* blocksum;
    uint32_t nscanblk;
    uint32_t* act;         // active cells: segment << 6 | bit
    uint32_t* rec_info;    // lut_off | nt << 14 | n_created << 18
    uint64_t* rec_own;     // created edge ids, 4 bits each, creation order
    uint32_t* rec_pre;     // within-segment prefix: vertices | triangles << 16
    uint32_t cap_active;
    uint32_t* emap;        // [4][nz][ny][nx] vertex id per grid edge (X,Y,Z) / cell centre
    McCounters* counters;
};

struct McMeshOut {
    float* vertices;
    float* colors;
    float* normals;
    float* grid_vertices;  // optional (debug): positions in voxel units before Mesh.Transform
    int32_t* triangles;
    uint32_t cap_vertices;
    size_t cap_indices;
    int64_t vertex_base;   // added to every triangle index (slab sharding)
    float sc[3], tr[3];    // Mesh.Transform: v*sc + tr (MarchingCubes.cs:85-90)
    float inv[3];          // diagonal of transpose(inverse(scale)) (Mesh.cs:49-55)
    float* bounds_partial; // [grid][6]
};

}  // namespace sdfk
