// mc_kernels.hip -- marching-cubes pipeline for MI355X (gfx950, wave64).
//
// The reference (MarchingCubes.cs:39-92 + Cell.cs) is one serial z->y->x sweep whose
// output order (vertex numbering, triangle order, float32 normal accumulation order)
// depends on that sweep.  This pipeline reproduces the same output in parallel:
//
//   K1 signbits   volume[x][y][z] (z fastest)  ->  1 bit per voxel, packed along X:
//                 bits[z][y][xw], bit b = (value(64*xw+b, y, z) > iso).  The only dense
//                 pass over the volume: 4 B/voxel read, 1/8 B/voxel written.
//   K2a segments  one thread per 64-cell X-run ("segment", = 64 consecutive cells of the
//                 serial sweep): bit-parallel test "8 corners not all equal", wave
//                 ballot/prefix-sum compaction of the active cells into a list.
//   K2b resolve   one thread per active cell: gathers the 8 corners, runs the 33-case
//                 dispatcher, decides which edge vertices the cell CREATES in the serial
//                 sweep (it is the first cell of the sweep that touches that grid edge).
//   K2c segsum    per-segment ordered prefix of created-vertex / triangle counts.
//   K3  scan      exclusive scan of the per-segment counts in serial-sweep order.
//   K4  vertices  creator cells write position / colour / normal at the vertex's serial
//                 index; the normal is a gather over the <=4 cells around the edge in
//                 sweep order (bit-exact float32 accumulation order, no atomics).
//   K5  triangles every active cell writes its triangle indices at its serial offset,
//                 reading vertex ids from a sparse per-grid-edge map written by K4.
//
// LUTs live in __constant__ memory (mc_device.h).  No MFMA: nothing here is a
// contraction.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_device.h"
#include "mc_params.h"

namespace sdfk {

// ---------------------------------------------------------------------------
// K1: sign bits
// ---------------------------------------------------------------------------
// Fast path (nz % 4 == 0): a workgroup stages a tile of 64*WX rows (X) x 64 voxels (Z)
// of one Y.  Each lane loads a float4 (4 consecutive z), reduces it to a 4-bit nibble in
// LDS; then, with lane = x, four __ballot()s per nibble column give the four 64-bit
// X-words of z..z+3.  Loads are 256-B contiguous per row, 1 KiB per wave instruction.
template <int WX>
__global__ __launch_bounds__(256) void k_signbits_tile(const float* __restrict__ values,
                                                       uint64_t* __restrict__ bits, int nx, int ny,
                                                       int nz, int nxw, float iso)
{
    constexpr int ROWS = 64 * WX;
    constexpr int PITCH = 20;  // bytes per LDS row: 16 nibbles + 4 pad (5 dwords: odd bank stride)
    __shared__ uint8_t nib[ROWS * PITCH];
    const int z0 = blockIdx.x * 64;
    const int y = blockIdx.y;
    const int x0 = blockIdx.z * ROWS;
    const size_t row_stride = (size_t)ny * nz;
    const float* base = values + (size_t)y * nz;
#pragma unroll 8
    for (int s = threadIdx.x; s < ROWS * 16; s += 256) {
        const int row = s >> 4, q = s & 15;
        const int x = x0 + row, z = z0 + 4 * q;
        unsigned n = 0;
        if (x < nx && z < nz) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)x * row_stride + z);
            n = (v.x > iso ? 1u : 0u) | (v.y > iso ? 2u : 0u) | (v.z > iso ? 4u : 0u) | (v.w > iso ? 8u : 0u);
        }
        nib[row * PITCH + q] = (uint8_t)n;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int pr = wave; pr < WX * 16; pr += 4) {
        const int xw = pr >> 4, q = pr & 15;
        const unsigned n = nib[(xw * 64 + lane) * PITCH + q];
        const uint64_t b0 = __ballot(n & 1u), b1 = __ballot(n & 2u), b2 = __ballot(n & 4u), b3 = __ballot(n & 8u);
        const int z = z0 + 4 * q;
        const int gxw = blockIdx.z * WX + xw;
        if (lane < 4 && z < nz && gxw < nxw) {
            const uint64_t w = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
            bits[((size_t)(z + lane) * ny + y) * nxw + gxw] = w;
        }
    }
}

// Generic path (any nz): lane = x, one ballot per (z, y, xw).  Each lane walks 16
// consecutive z of its own row so its cache line is reused; only used for odd sizes.
__global__ __launch_bounds__(256) void k_signbits_generic(const float* __restrict__ values,
                                                          uint64_t* __restrict__ bits, int nx, int ny,
                                                          int nz, int nxw, float iso)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int zrun = (blockIdx.x * 4 + wave) * 16;
    const int y = blockIdx.y;
    const int xw = blockIdx.z;
    const int x = xw * 64 + lane;
    if (zrun >= nz) return;
    const float* p = values + ((size_t)(x < nx ? x : 0) * ny + y) * nz;
    for (int z = zrun; z < zrun + 16 && z < nz; z++) {
        const bool s = (x < nx) && (p[z] > iso);
        const uint64_t w = __ballot(s);
        if (lane == 0) bits[((size_t)z * ny + y) * nxw + xw] = w;
    }
}

// ---------------------------------------------------------------------------
// K2a: per-segment activity, compaction of active cells
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t shr1_in(uint64_t w, uint64_t next) { return (w >> 1) | (next << 63); }

constexpr int SEG_PER_THREAD = 8;                    // segments handled by one lane
constexpr int SEG_PER_BLOCK = 256 * SEG_PER_THREAD;  // one returning atomic per 2048 segments

__device__ __forceinline__ uint64_t segment_active_mask(const McParams& P, uint32_t s, uint64_t& m13)
{
    const uint32_t xw = s % P.nxw;
    const uint32_t t = s / P.nxw;
    const uint32_t y = t % P.ncy;
    const uint32_t z = t / P.ncy;
    m13 = 0;
    if ((int)z < P.lay_count_begin || (int)z >= P.lay_emit_end) return 0;
    const uint64_t* r00 = P.bits + ((size_t)z * P.ny + y) * P.nxw + xw;  // (y  , z  )
    const uint64_t* r01 = r00 + P.nxw;                                     // (y+1, z  )
    const uint64_t* r10 = r00 + (size_t)P.ny * P.nxw;                      // (y  , z+1)
    const uint64_t* r11 = r10 + P.nxw;                                     // (y+1, z+1)
    const bool more = (xw + 1 < (uint32_t)P.nxw);
    const uint64_t a = r00[0], b = r01[0], c = r10[0], d = r11[0];
    const uint64_t as = shr1_in(a, more ? r00[1] : 0), bs = shr1_in(b, more ? r01[1] : 0);
    const uint64_t cs = shr1_in(c, more ? r10[1] : 0), ds = shr1_in(d, more ? r11[1] : 0);
    // cells of this word: x = 64*xw + bit, valid while x < ncx
    const int rem = P.ncx - (int)xw * 64;
    const uint64_t valid = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    const uint64_t all1 = a & as & b & bs & c & cs & d & ds;
    const uint64_t any1 = a | as | b | bs | c | cs | d | ds;
    // corner sign words 0xA5 / 0x5A (case 13): v0=a v1=as v2=bs v3=b v4=c v5=cs v6=ds v7=d
    const uint64_t pa5 = a & bs & cs & d & ~as & ~b & ~c & ~ds;
    const uint64_t p5a = ~a & ~bs & ~cs & ~d & as & b & c & ds;
    m13 = (pa5 | p5a) & valid;
    return (any1 & ~all1) & valid;
}

__global__ __launch_bounds__(256) void k_segments(McParams P)
{
    __shared__ uint32_t s_wave_tot[4];
    __shared__ uint32_t s_base;
    const uint32_t nseg = P.nseg;
    const uint32_t chunk = blockIdx.x * (uint32_t)SEG_PER_BLOCK;
    uint64_t active[SEG_PER_THREAD];
    uint32_t cnt = 0, n13 = 0;
#pragma unroll
    for (int i = 0; i < SEG_PER_THREAD; i++) {
        const uint32_t s = chunk + (uint32_t)i * 256u + threadIdx.x;   // coalesced across lanes
        active[i] = 0;
        if (s < nseg) {
            uint64_t m13;
            active[i] = segment_active_mask(P, s, m13);
            n13 += (uint32_t)__popcll(m13);
            P.segpack[s] = 0;
        }
        cnt += (uint32_t)__popcll(active[i]);
    }
    if (n13) atomicAdd(&P.counters->n_case13, n13);
    // exclusive prefix of the per-lane counts; ONE returning atomic per workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_wave_tot[0] + s_wave_tot[1] + s_wave_tot[2] + s_wave_tot[3];
        s_base = tot ? atomicAdd(&P.counters->n_active, tot) : 0u;
    }
    __syncthreads();
    if (cnt) {
        uint32_t pos = s_base + (incl - cnt);
        for (int w = 0; w < wave; w++) pos += s_wave_tot[w];
#pragma unroll
        for (int i = 0; i < SEG_PER_THREAD; i++) {
            const uint32_t s = chunk + (uint32_t)i * 256u + threadIdx.x;
            uint64_t m = active[i];
            while (m) {   // the cells of one segment stay contiguous and in x order
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                if (pos < P.cap_active) P.act[pos] = (s << 6) | (uint32_t)bit;
                pos++;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// shared helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ void decode_cell(const McParams& P, uint32_t id, int& x, int& y, int& z, uint32_t& seg)
{
    seg = id >> 6;
    const uint32_t xw = seg % P.nxw;
    const uint32_t t = seg / P.nxw;
    y = (int)(t % P.ncy);
    z = (int)(t / P.ncy);
    x = (int)(xw * 64 + (id & 63));
}

// the 8 corner voxels of cell (x,y,z) into this thread's LDS column ([corner][thread])
__device__ __forceinline__ void stage_corners(const McParams& P, int x, int y, int z, float* col, int stride)
{
    const size_t sx = (size_t)P.ny * P.nz, sy = (size_t)P.nz;
    const float* p = P.values + (size_t)x * sx + (size_t)y * sy + z;
    const float a0 = p[0], a4 = p[1], a1 = p[sx], a5 = p[sx + 1];
    const float a3 = p[sy], a7 = p[sy + 1], a2 = p[sx + sy], a6 = p[sx + sy + 1];
    col[0] = a0; col[stride] = a1; col[2 * stride] = a2; col[3 * stride] = a3;
    col[4 * stride] = a4; col[5 * stride] = a5; col[6 * stride] = a6; col[7 * stride] = a7;
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
__device__ __noinline__ bool cell_is_dead(const float* values, int ny, int nz, float iso, int x, int y, int z, float* col)
{
    {
        const size_t sx = (size_t)ny * nz, sy = (size_t)nz;
        const float* p = values + (size_t)x * sx + (size_t)y * sy + z;
        col[0] = p[0]; col[256] = p[sx]; col[2 * 256] = p[sx + sy]; col[3 * 256] = p[sy];
        col[4 * 256] = p[1]; col[5 * 256] = p[sx + 1]; col[6 * 256] = p[sx + sy + 1]; col[7 * 256] = p[sy + 1];
    }
    const CornersLds v{col, 256, (double)iso};
    int index = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) index |= (v[k] > 0.0) ? (1 << k) : 0;
    if (index != 0xA5 && index != 0x5A) return false;
    const Tiling t = mc_resolve(v);
    return t.nt == 0;
}

// Does an earlier cell of the sweep (one that is alive) share edge e of cell (x,y,z)?
// Predecessor sets derived from Cell.cs:371-441 (which cells map to the same face-layer
// slot) and the sweep order of MarchingCubes.cs:53-80.
// up to three predecessors per edge, offsets (dx,dy,dz); 9 = none
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

__device__ __forceinline__ bool edge_has_live_predecessor(const McParams& P, int e, int x, int y, int z, bool check_dead, float* col)
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
        if (!check_dead || !cell_is_dead(P.values, P.ny, P.nz, P.iso, px, py, pz, col)) return true;
    }
    return false;
}

// ---------------------------------------------------------------------------
// K2b: resolve tilings and vertex creation per active cell
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(McParams P)
{
    __shared__ float s_v[8 * 256];   // [corner][thread]: run-time corner indexing without scratch
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const bool check_dead = P.counters->n_case13 != 0;
    float* col = s_v + threadIdx.x;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        int x, y, z;
        uint32_t seg;
        decode_cell(P, P.act[i], x, y, z, seg);
        stage_corners(P, x, y, z, col, 256);
        const CornersLds v{col, 256, (double)P.iso};
        const Tiling t = mc_resolve(v);
        uint32_t info = 0;
        uint64_t own = 0;
        if (t.nt > 0) {
            uint32_t seen = 0;
            int nown = 0;
            for (int k = 0; k < 3 * t.nt; k++) {
                const int e = c_lut[t.lut_off + k];
                if (seen & (1u << e)) continue;
                seen |= 1u << e;
                const bool mine = (e == 12) || !edge_has_live_predecessor(P, e, x, y, z, check_dead, col);
                if (mine) {
                    own |= (uint64_t)e << (4 * nown);
                    nown++;
                }
            }
            const bool emit = (z >= P.lay_emit_begin);
            info = (uint32_t)t.lut_off | ((uint32_t)(emit ? t.nt : 0) << 14) | ((uint32_t)nown << 18);
        } else if (t.index == 0xA5 || t.index == 0x5A) {
            atomicAdd(&P.counters->n_dead, 1u);
        }
        P.rec_info[i] = info;
        P.rec_own[i] = own;
    }
}

// ---------------------------------------------------------------------------
// K2c: ordered within-segment prefix + segment totals
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segsum(McParams P)
{
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t seg = P.act[i] >> 6;
        if (i > 0 && (P.act[i - 1] >> 6) == seg) continue;  // not the first cell of its segment
        uint32_t vs = 0, ts = 0;
        for (uint32_t j = i; j < n && (P.act[j] >> 6) == seg; j++) {
            const uint32_t info = P.rec_info[j];
            P.rec_pre[j] = vs | (ts << 16);
            vs += (info >> 18) & 15u;
            ts += (info >> 14) & 15u;
        }
        P.segpack[seg] = vs | (ts << 16);
    }
}

// ---------------------------------------------------------------------------
// K3: exclusive scan of segpack (serial-sweep order) -> segprefix
// ---------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__device__ __forceinline__ uint2 block_reduce_add(uint2 v, uint2* smem)
{
    // wave reduce
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v.x += __shfl_down(v.x, o);
        v.y += __shfl_down(v.y, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    uint2 r = make_uint2(0, 0);
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { r.x += smem[w].x; r.y += smem[w].y; }
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_scan_reduce(McParams P)
{
    __shared__ uint2 sm[4];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint2 acc = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + k * 256 + threadIdx.x;
        if (i < P.nseg) {
            const uint32_t p = P.segpack[i];
            acc.x += p & 0xffffu;
            acc.y += p >> 16;
        }
    }
    const uint2 tot = block_reduce_add(acc, sm);
    if (threadIdx.x == 0) P.blocksum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_scan_blocks(McParams P)
{
    // single workgroup: exclusive scan of blocksum[0..nblk) in place, totals to counters
    __shared__ uint2 sm[16];
    __shared__ uint2 s_carry;
    if (threadIdx.x == 0) s_carry = make_uint2(0, 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < P.nscanblk; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        uint2 v = (i < P.nscanblk) ? P.blocksum[i] : make_uint2(0, 0);
        uint2 incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t ax = __shfl_up(incl.x, o), ay = __shfl_up(incl.y, o);
            if (lane >= o) { incl.x += ax; incl.y += ay; }
        }
        if (lane == 63) sm[wave] = incl;
        __syncthreads();
        uint2 wpre = make_uint2(0, 0), all = make_uint2(0, 0);
        for (int w = 0; w < 16; w++) {
            if (w < wave) { wpre.x += sm[w].x; wpre.y += sm[w].y; }
            all.x += sm[w].x; all.y += sm[w].y;
        }
        const uint2 carry = s_carry;
        if (i < P.nscanblk)
            P.blocksum[i] = make_uint2(carry.x + wpre.x + incl.x - v.x, carry.y + wpre.y + incl.y - v.y);
        __syncthreads();
        if (threadIdx.x == 0) s_carry = make_uint2(carry.x + all.x, carry.y + all.y);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        P.counters->total_v = s_carry.x;
        P.counters->total_t = s_carry.y;
    }
}

__global__ __launch_bounds__(256) void k_scan_final(McParams P)
{
    __shared__ uint2 sm[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint2 item[SCAN_ITEMS];
    uint2 acc = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + k;
        uint32_t p = (i < P.nseg) ? P.segpack[i] : 0u;
        item[k] = make_uint2(p & 0xffffu, p >> 16);
        acc.x += item[k].x;
        acc.y += item[k].y;
    }
    // exclusive scan of per-thread sums across the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 incl = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ax = __shfl_up(incl.x, o), ay = __shfl_up(incl.y, o);
        if (lane >= o) { incl.x += ax; incl.y += ay; }
    }
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    uint2 pre = P.blocksum[blockIdx.x];
    for (int w = 0; w < wave; w++) { pre.x += sm[w].x; pre.y += sm[w].y; }
    pre.x += incl.x - acc.x;
    pre.y += incl.y - acc.y;
    const uint32_t seg_emit0 = (uint32_t)P.lay_emit_begin * (uint32_t)P.ncy * (uint32_t)P.nxw;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + k;
        if (i < P.nseg) {
            P.segprefix[i] = pre;
            if (i == seg_emit0) P.counters->nghost = pre.x;
        }
        pre.x += item[k].x;
        pre.y += item[k].y;
    }
}

// ---------------------------------------------------------------------------
// K4: vertices
// ---------------------------------------------------------------------------
__device__ __forceinline__ float v3len(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

__device__ __forceinline__ void load_corner_color(const McParams& P, int x, int y, int z, int corner, float* c)
{
    if (!P.colors) { c[0] = c[1] = c[2] = 0.0f; return; }
    const size_t o = ((size_t)(x + c_corner_dx[corner]) * P.ny + (y + c_corner_dy[corner])) * P.nz + (z + c_corner_dz[corner]);
    c[0] = P.colors[o * 3]; c[1] = P.colors[o * 3 + 1]; c[2] = P.colors[o * 3 + 2];
}

// Accumulate into n[] what a cell with corners v adds for its edge `es`, in the order of
// Cell.cs:332-333/355-356: once per occurrence in the LUT row, corner 1 then corner 2.
template <class V>
__device__ __forceinline__ void add_cell_edge_gradients(const V& v, int lut_off, int nt, int es, float* n)
{
    int occ = 0;
    for (int k = 0; k < 3 * nt; k++) occ += (c_lut[lut_off + k] == es) ? 1 : 0;
    if (!occ) return;
    const int i1 = MC_L2(edgesrelz, es, 0) * 4 + MC_L2(edgesrely, es, 0) * 2 + MC_L2(edgesrelx, es, 0);
    const int i2 = MC_L2(edgesrelz, es, 1) * 4 + MC_L2(edgesrely, es, 1) * 2 + MC_L2(edgesrelx, es, 1);
    const double w1 = 1.0 / (MC_EPS + fabs(v[c_bit_to_corner[i1]]));
    const double w2 = 1.0 / (MC_EPS + fabs(v[c_bit_to_corner[i2]]));
    float g1[3], g2[3];
    // NB: Cell.cs:157-158 indexes the corner-ordered gradient table with the BIT-order
    // index (inherited quirk); reproduced: gradient of "corner i1", not of corner bit_to_corner[i1].
#pragma unroll
    for (int j = 0; j < 3; j++) {
        g1[j] = (float)(mc_corner_gradient(v, i1, j) * w1);
        g2[j] = (float)(mc_corner_gradient(v, i2, j) * w2);
    }
    for (int o = 0; o < occ; o++) {
        n[0] = n[0] + g1[0]; n[1] = n[1] + g1[1]; n[2] = n[2] + g1[2];
        n[0] = n[0] + g2[0]; n[1] = n[1] + g2[1]; n[2] = n[2] + g2[2];
    }
}

// Workgroup-level re-balancing: a chunk of 256 records owns a variable number of output
// items each (created vertices / triangle indices).  An exclusive prefix in LDS plus a
// binary search turns "one thread per record" into "one thread per output item".
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* s_pre /*[257]*/, uint32_t* s_wave /*[4]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t pre = incl - v;
    for (int w = 0; w < wave; w++) pre += s_wave[w];
    s_pre[threadIdx.x] = pre;
    if (threadIdx.x == 255) s_pre[256] = pre + v;
    __syncthreads();
    return s_pre[256];
}

__device__ __forceinline__ int find_owner_256(const uint32_t* s_pre, uint32_t j)
{
    int lo = 0, hi = 255;   // largest r with s_pre[r] <= j
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_pre[mid] <= j) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_vertices(McParams P, McMeshOut M)
{
    __shared__ float s_c[4 * 8 * 256];   // corners of the <=4 cells around the edge: [cell][corner][thread]
    __shared__ uint32_t s_pre[257];
    __shared__ uint32_t s_wave[4];
    __shared__ float s_red[6][4];
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const uint32_t nghost = P.counters->nghost;
    const size_t nvox = (size_t)P.nx * P.ny * P.nz;
    const double iso = (double)P.iso;
    const double stp = (double)P.step;
    float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    float* col = s_c + threadIdx.x;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t irec = base + threadIdx.x;
        const uint32_t my_nown = (irec < n) ? ((P.rec_info[irec] >> 18) & 15u) : 0u;
        const uint32_t total = block_exclusive_scan_256(my_nown, s_pre, s_wave);
        for (uint32_t j = threadIdx.x; j < total; j += 256u) {
            const int rr = find_owner_256(s_pre, j);
            const uint32_t i = base + (uint32_t)rr;
            const int r = (int)(j - s_pre[rr]);
            const uint32_t info = P.rec_info[i];
            int x, y, z;
            uint32_t seg;
            decode_cell(P, P.act[i], x, y, z, seg);
            const uint32_t vi = P.segprefix[seg].x + (P.rec_pre[i] & 0xffffu) + (uint32_t)r;
            const int e = (int)((P.rec_own[i] >> (4 * r)) & 15u);
            const int dir = c_edge_dir[e];
            const int gx = x + c_edge_ox[e], gy = y + c_edge_oy[e], gz = z + c_edge_oz[e];
            P.emap[(size_t)dir * nvox + ((size_t)gz * P.ny + gy) * P.nx + gx] = vi;
            if (z < P.lay_emit_begin) continue;   // context layer: only its vertex ids are needed
            const uint32_t out = vi - nghost;
            if (out >= M.cap_vertices) { P.counters->overflow = 1u; continue; }
            const int lut_off = (int)(info & 0x3fffu);
            const int nt_row = (int)((info >> 14) & 15u);
            float pos[3], colr[3], nrm[3] = {0.0f, 0.0f, 0.0f};
            const int xs = x * P.step, ys = y * P.step, zs = (z + P.z0) * P.step;
            if (e == 12) {
                // Cell.CalculateCenterVertex, Cell.cs:501-549
                stage_corners(P, x, y, z, col, 256);
                const CornersLds v{col, 256, iso};
                double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0, gsum[3] = {0.0, 0.0, 0.0};
                float fc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double wk = 1.0 / (MC_EPS + fabs(v[k]));
                    fx += (double)((k == 1 || k == 2 || k == 5 || k == 6) ? 1 : 0) * wk;
                    fy += (double)((k == 2 || k == 3 || k == 6 || k == 7) ? 1 : 0) * wk;
                    fz += (double)(k >= 4 ? 1 : 0) * wk;
                    ff += wk;
                    float ck[3];
                    load_corner_color(P, x, y, z, k, ck);
                    const float wf = (float)wk;
                    if (k == 0) { fc[0] = ck[0] * wf; fc[1] = ck[1] * wf; fc[2] = ck[2] * wf; }
                    else { fc[0] = fc[0] + ck[0] * wf; fc[1] = fc[1] + ck[1] * wf; fc[2] = fc[2] + ck[2] * wf; }
#pragma unroll
                    for (int jj = 0; jj < 3; jj++) {
                        const double term = wk * mc_corner_gradient(v, k, jj);
                        gsum[jj] = (k == 0) ? term : gsum[jj] + term;
                    }
                }
                pos[0] = (float)((double)xs + stp * fx / ff);
                pos[1] = (float)((double)ys + stp * fy / ff);
                pos[2] = (float)((double)zs + stp * fz / ff);
#pragma unroll
                for (int jj = 0; jj < 3; jj++) colr[jj] = (float)((double)fc[jj] / ff);
                const float g0 = (float)gsum[0], g1 = (float)gsum[1], g2 = (float)gsum[2];
                int occ = 0;
                for (int k = 0; k < 3 * nt_row; k++) occ += (c_lut[lut_off + k] == 12) ? 1 : 0;
                for (int o = 0; o < occ; o++) { nrm[0] = nrm[0] + g0; nrm[1] = nrm[1] + g1; nrm[2] = nrm[2] + g2; }
            } else {
                // stage the corners of every in-range cell around this grid edge (sweep order)
                int own_s = 0;
                unsigned okmask = 0;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int cx = gx + c_share_dx[dir][s], cy = gy + c_share_dy[dir][s], cz = gz + c_share_dz[dir][s];
                    const bool ok = cell_in_range(P, cx, cy, cz);
                    okmask |= ok ? (1u << s) : 0u;
                    if (cx == x && cy == y && cz == z) own_s = s;
                    if (ok) stage_corners(P, cx, cy, cz, col + s * (8 * 256), 256);
                }
                // Cell.AddFaceFromEdgeIndex, Cell.cs:314-350 (creator-cell frame)
                const CornersLds v{col + own_s * (8 * 256), 256, iso};
                const int dx1 = MC_L2(edgesrelx, e, 0), dx2 = MC_L2(edgesrelx, e, 1);
                const int dy1 = MC_L2(edgesrely, e, 0), dy2 = MC_L2(edgesrely, e, 1);
                const int dz1 = MC_L2(edgesrelz, e, 0), dz2 = MC_L2(edgesrelz, e, 1);
                const int c1 = c_bit_to_corner[dz1 * 4 + dy1 * 2 + dx1], c2 = c_bit_to_corner[dz2 * 4 + dy2 * 2 + dx2];
                const double w1 = 1.0 / (MC_EPS + fabs(v[c1]));
                const double w2 = 1.0 / (MC_EPS + fabs(v[c2]));
                double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
                fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
                fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
                float ca[3], cb[3];
                load_corner_color(P, x, y, z, c1, ca);
                load_corner_color(P, x, y, z, c2, cb);
                const float w1f = (float)w1, w2f = (float)w2;
                pos[0] = (float)((double)xs + stp * fx / ff);
                pos[1] = (float)((double)ys + stp * fy / ff);
                pos[2] = (float)((double)zs + stp * fz / ff);
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                    const float cj = ca[jj] * w1f + cb[jj] * w2f;
                    colr[jj] = (float)((double)cj / ff);
                }
                // normal: gather over the cells around the edge, in sweep order
#pragma unroll 1
                for (int s = 0; s < 4; s++) {
                    if (!((okmask >> s) & 1u)) continue;
                    const CornersLds vs{col + s * (8 * 256), 256, iso};
                    const int es = c_share_edge[dir][s];
                    int lo = lut_off, nts = nt_row;
                    if (s != own_s) {
                        const Tiling ts = mc_resolve(vs);
                        lo = ts.lut_off;
                        nts = ts.nt;
                    }
                    if (nts > 0) add_cell_edge_gradients(vs, lo, nts, es, nrm);
                }
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
            float* oc = M.colors + (size_t)out * 3;
            float* on = M.normals + (size_t)out * 3;
            ov[0] = px; ov[1] = py; ov[2] = pz;
            oc[0] = colr[0]; oc[1] = colr[1]; oc[2] = colr[2];
            on[0] = t0 / tl; on[1] = t1 / tl; on[2] = t2 / tl;
            if (M.grid_vertices) {
                float* og = M.grid_vertices + (size_t)out * 3;
                og[0] = pos[0]; og[1] = pos[1]; og[2] = pos[2];
            }
            bmin[0] = fminf(bmin[0], px); bmin[1] = fminf(bmin[1], py); bmin[2] = fminf(bmin[2], pz);
            bmax[0] = fmaxf(bmax[0], px); bmax[1] = fmaxf(bmax[1], py); bmax[2] = fmaxf(bmax[2], pz);
        }
        __syncthreads();   // s_pre is rewritten by the next chunk
    }
    // per-workgroup AABB partials (Mesh.Measure, Mesh.cs:30-45), reduced by k_bounds
    float r[6] = {bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2]};
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
        M.bounds_partial[(size_t)blockIdx.x * 6 + j] = a;
    }
}

__global__ __launch_bounds__(256) void k_bounds(const float* __restrict__ partial, int nblk, float* __restrict__ bounds)
{
    __shared__ float s_red[6][4];
    float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int b = threadIdx.x; b < nblk; b += 256) {
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], partial[(size_t)b * 6 + j]);
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], partial[(size_t)b * 6 + j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        for (int j = 0; j < 3; j++) r[j] = fminf(r[j], __shfl_down(r[j], o));
        for (int j = 3; j < 6; j++) r[j] = fmaxf(r[j], __shfl_down(r[j], o));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int j = 0; j < 6; j++) s_red[j][wave] = r[j];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int j = threadIdx.x;
        float a = s_red[j][0];
        for (int w = 1; w < 4; w++) a = (j < 3) ? fminf(a, s_red[j][w]) : fmaxf(a, s_red[j][w]);
        bounds[j] = a;
    }
}

// ---------------------------------------------------------------------------
// K5: triangles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_triangles(McParams P, McMeshOut M)
{
    __shared__ uint32_t s_pre[257];
    __shared__ uint32_t s_wave[4];
    const uint32_t n = min(P.counters->n_active, P.cap_active);
    const uint32_t nghost = P.counters->nghost;
    const size_t nvox = (size_t)P.nx * P.ny * P.nz;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t irec = base + threadIdx.x;
        const uint32_t my_ni = (irec < n) ? 3u * ((P.rec_info[irec] >> 14) & 15u) : 0u;
        const uint32_t total = block_exclusive_scan_256(my_ni, s_pre, s_wave);
        for (uint32_t j = threadIdx.x; j < total; j += 256u) {   // one lane per triangle index
            const int rr = find_owner_256(s_pre, j);
            const uint32_t i = base + (uint32_t)rr;
            const uint32_t k = j - s_pre[rr];
            const uint32_t info = P.rec_info[i];
            int x, y, z;
            uint32_t seg;
            decode_cell(P, P.act[i], x, y, z, seg);
            const size_t o = ((size_t)P.segprefix[seg].y + (P.rec_pre[i] >> 16)) * 3 + k;
            if (o >= M.cap_indices) { P.counters->overflow = 1u; continue; }
            const int e = c_lut[(info & 0x3fffu) + k];
            const int dir = c_edge_dir[e];
            const int gx = x + c_edge_ox[e], gy = y + c_edge_oy[e], gz = z + c_edge_oz[e];
            const uint32_t vi = P.emap[(size_t)dir * nvox + ((size_t)gz * P.ny + gy) * P.nx + gx];
            M.triangles[o] = (int32_t)((int64_t)vi - (int64_t)nghost + M.vertex_base);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// auxiliary volume kernels
// ---------------------------------------------------------------------------
// Voxels.ClipToBounds, Voxels.cs:133-167 (all six faces get Size.X/NX)
__global__ __launch_bounds__(256) void k_clip(float* __restrict__ values, int nx, int ny, int nz, int z0, int nz_global, float outside)
{
    const size_t n = (size_t)nx * ny * nz;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const int z = (int)(i % nz);
        const size_t t = i / nz;
        const int y = (int)(t % ny);
        const int x = (int)(t / ny);
        const int zg = z + z0;
        if (x == 0 || y == 0 || zg == 0 || x == nx - 1 || y == ny - 1 || zg == nz_global - 1) values[i] = outside;
    }
}

// step > 1 (MarchingCubes.cs:49-80): the sweep only ever touches voxels whose indices are
// multiples of `step`; gather them into a dense volume and mesh that with unit cells.
__global__ __launch_bounds__(256) void k_subsample(const float* __restrict__ src, const float* __restrict__ srcc,
                                                   float* __restrict__ dst, float* __restrict__ dstc, int nx, int ny,
                                                   int nz, int mx, int my, int mz, int step)
{
    const size_t n = (size_t)mx * my * mz;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const int z = (int)(i % mz);
        const size_t t = i / mz;
        const int y = (int)(t % my);
        const int x = (int)(t / my);
        const size_t o = ((size_t)(x * step) * ny + (size_t)(y * step)) * nz + (size_t)(z * step);
        dst[i] = src[o];
        if (srcc) { dstc[i * 3] = srcc[o * 3]; dstc[i * 3 + 1] = srcc[o * 3 + 1]; dstc[i * 3 + 2] = srcc[o * 3 + 2]; }
    }
}

}  // namespace sdfk
