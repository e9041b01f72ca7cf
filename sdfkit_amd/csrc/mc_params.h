// mc_params.h -- kernel argument blocks shared by mc_kernels.hip and the host driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdfk {

// Records per chunk of k_resolve / k_vertices / k_triangles (one 256-thread workgroup per chunk).  Not 256: a smooth surface
// creates ~1.00 vertices per active cell, so a 256-record chunk holds 256 +- 15 vertices and every other chunk needed a
// SECOND, almost empty pass of the per-vertex loop of k_vertices (x1.36 passes per chunk on the 256^3 sphere, x1.45 on the
// README scene); with 240 records (16 idle lanes in the per-record phases) it is x1.02 / x1.09: a fifth fewer passes of the
// loop that is 80 % of the kernel.
#ifndef MC_CHUNK_RECORDS
#define MC_CHUNK_RECORDS 240   // (A/B builds: tools/variants.sh build c256 "-DMC_CHUNK_RECORDS=256")
#endif
constexpr uint32_t MC_CHUNK = MC_CHUNK_RECORDS;
static_assert(MC_CHUNK >= 64 && MC_CHUNK <= 256, "one 256-thread workgroup per chunk");
// vertex-id slots per chunk (rec_vid): a record references at most 13 distinct vertex ids (12 edges + the centre vertex)
// logical compaction blocks / chunks of records above which ONE workgroup computes their prefix between two kernels of the chain
// (k_blockscan, k_chunkscan) instead of every consumer workgroup summing its predecessors itself
// (8192: a one-workgroup launch costs a serial chain ~6 us; what it saves grows with the square of the count -- ~2 us at 5 000 chunks,
// 14 us at 13 000, 30 us at 16 000 blocks: profiles/r05_ab_prefix_scans_1024.txt)
#ifndef SDFK_SCAN_BLOCKS
#define SDFK_SCAN_BLOCKS 8192
#endif
#ifndef SDFK_SCAN_CHUNKS
#define SDFK_SCAN_CHUNKS 8192
#endif
constexpr int MC_SCAN_BLOCKS = SDFK_SCAN_BLOCKS;
constexpr uint32_t MC_SCAN_CHUNKS = SDFK_SCAN_CHUNKS;
constexpr uint32_t MC_VSTRIDE = MC_CHUNK * 13u;
static_assert(MC_VSTRIDE <= 4096, "the in-chunk slot prefix has 12 bits in rec_info");

struct McCounters {
    uint32_t n_active;     // active cells listed (may exceed the list capacity)
    uint32_t n_case13;     // cells whose sign word is 0xA5/0x5A (candidates for "impossible 13")
    uint32_t n_dead;       // case-13 cells that resolved to no tiling, in emitted layers
    uint32_t total_v;      // vertices created in [lay_count_begin, lay_emit_end)
    uint32_t total_t;      // triangles emitted in [lay_emit_begin, lay_emit_end)
    uint32_t nghost;       // vertices created below lay_emit_begin
    uint32_t n_ghost_cells;// active cells listed below lay_emit_begin
    uint32_t overflow;     // 1: an output capacity was too small; 2: a bounded spin gave up
    uint32_t n_emit_cells; // active cells inside emitted layers
    uint32_t pad[7];
};

struct McParams {
    const float* values;   // [nx][ny][nz], z fastest (Voxels.cs:8)
    const float* colors;   // [nx][ny][nz][3] or nullptr (= zeros)
    int nx, ny, nz;        // voxel dims of this (slab) volume
    int nzp;               // row pitch of values / colors in voxels: nz rounded up to a multiple of 4 (16-byte aligned rows)
    int ncx, ncy, ncz;     // cell dims = n-1
    int nxw;               // 64-bit X words per (z,y) row of the sign-bit array
    int z0;                // global z of local plane 0
    int lay_count_begin;   // first local cell layer whose created vertices are numbered
    int lay_emit_begin;    // first local cell layer that is emitted
    int lay_emit_end;      // one past the last emitted layer
    int lay_list_end;      // one past the last layer that is classified (emit_end or +1)
    float iso;
    int step;              // scale of cell coordinates in vertex positions (Cell.cs:345-347)
    // workspace
    const uint64_t* bits;  // [nz][ny][nxw]: bit b of word xw = (value(64*xw+b, y, z) > iso)
    uint32_t* zero_cull;   // non-null (volume-less jobs): the 64 counters (32 words apart) of the culling kernel that made `bits`; the count
                           // pass clears them, so that the next job of the lane needs no memset
    int bpl;               // logical blocks of k_compact per layer: ceil(ncy * nxw / 1024)
    uint64_t* blockcnt;    // per logical block of k_compact: active cells | case-13 sign words << 32
    uint32_t* wavecnt;     // active cells per wavefront of the count pass ([block][4])
    uint64_t* segmask;     // non-null: the count pass leaves the activity masks of the wavefronts that found something ([block][1024 segments],
                           // a bit per cell); the write pass loads ONE word per segment instead of sixteen sign words and recomputes nothing
    uint32_t* blockpre;    // non-null (grids of more than MC_SCAN_BLOCKS logical blocks): exclusive prefix of blockcnt's cell counts, by
                           // k_blockscan between the two passes -- a write-pass workgroup that sums its predecessors itself reads
                           // O(blocks) words, O(blocks^2) per launch: 16 K blocks at 1024^3
    uint64_t* chunktot;    // (vertices << 31 | triangles) per MC_CHUNK-record chunk (k_resolve)
    uint64_t* chunkpre;    // exclusive prefix of chunktot, per chunk (+ the total after the last): by k_vertices, read by k_triangles
    int chunkscan;         // 1 (record capacities of more than MC_SCAN_CHUNKS chunks): chunkpre comes from k_chunkscan, one workgroup
                           // between k_resolve and k_vertices -- a k_vertices workgroup that sums its predecessors' totals itself reads
                           // O(chunks) words: 13 K chunks at 1024^3, 13 dependent round trips per workgroup
    // Active cells ("records") in serial-sweep order.  Everything the emit kernels read is
    // compact (tens of MB, L2/MALL resident): no per-voxel maps.
    uint32_t* rec_xy;      // x | y << xbits
    int xbits;             // bits of x in rec_xy: 16, or more when nx > 65536 (then ny needs fewer: nx * ny * nz < 2^31, Voxels.cs:82)
    uint32_t xmask;        // (1 << xbits) - 1
    uint32_t* rec_z;       // z (local layer)
    uint32_t* rowstart;    // [(z - lay_count_begin) * ncy + y] = first record of cell row (z,y); +1 sentinel
    uint32_t* rec_info;    // in-chunk slot prefix (12 bits) | 1 << 13 | nt_emitted << 14 | n_created << 18 | row_id << 22   (0 = emits nothing)
    uint64_t* rec_own;     // created edge ids, 4 bits each, creation order
    uint32_t* rec_pre;     // in-chunk exclusive prefix: created vertices | triangles << 16
    float* rec_corners;    // 8 corner voxel values (v0..v7), 32 bytes per record
    uint32_t* rec_vid;     // [chunk][MC_VSTRIDE]: the vertex ids the chunk's emitting records reference, one slot per distinct id of a
                           // record's tiling row in increasing id order, records back to back (rec_info: slot prefix); pushed by the
                           // creators (k_vertices), loaded as one contiguous block per chunk by k_triangles (mc_device.h, "vertex-id SLOTS")
    uint32_t* chunkslots;  // slots used per chunk
    uint32_t* chunkdead;   // "impossible case 13" cells per MC_CHUNK-record chunk
    uint4* chunkwin;       // per chunk: (first row, last row, end of window 1, start of window 2) (K4 set-up)
    uint32_t* chunkwin2;   // per chunk: end of window 2
    uint32_t cap_active;
    McCounters* counters;       // device copy; every field is written by a kernel (no memset)
    McCounters* host_counters;  // pinned, device-mapped mirror the host reads after ONE sync
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
    int bounds_blocks;     // number of k_vertices workgroups
    float* bounds;         // device float[6] (Mesh.Min, Mesh.Max)
    float* host_bounds;    // pinned, device-mapped mirror
    void* slab_header;     // non-null: the mesh arrays are sections of a slab payload; k_triangles writes its 64-byte header here
    int32_t slab_vbytes;   // 36 or 24 (no colour section)
    uint2* vdesc;          // non-null: k_vertices leaves (creator record, edge) per emitted vertex here instead of gathering corner
                           // colours from the volume; the program's own sdfk_vertex_colors kernel re-evaluates them (sample_codegen.h)
};

}  // namespace sdfk
