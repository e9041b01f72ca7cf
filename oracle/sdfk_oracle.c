/* sdfk_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See sdfk_oracle.h for scope and pinning status ("parity unpinned" beyond the
 * reference's own known-answer tests).  Every function cites the reference
 * file:line (relative to praeclarum/SdfKit) whose behaviour it restates.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include "sdfk_oracle.h"
#include "lewiner_luts.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static const int8_t LUT[OLUT_BLOB_SIZE] = { OLUT_BLOB_VALUES };
#define L2D(name, i, j) (LUT[OLUT_OFF_##name + (i) * OLUT_DIM1_##name + (j)])
#define L1D(name, i) (LUT[OLUT_OFF_##name + (i)])
#define ROW2(name, i) (OLUT_OFF_##name + (i) * OLUT_DIM1_##name)
#define ROW3(name, i, j) (OLUT_OFF_##name + ((i) * OLUT_DIM1_##name + (j)) * OLUT_DIM2_##name)

/* Both MarchingCubes.cs:37 and Cell.cs:63 define FLT_EPSILON as this DOUBLE. */
static const double EPS = 0.0000001;

/* ======================================================================== */
/* BCL arithmetic restated (System.Numerics / MathF)                         */
/* ======================================================================== */
static float v3_length(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
static float sel_min(float a, float b) { return (a < b) ? a : b; } /* Vector3.Min component */
static float sel_max(float a, float b) { return (a > b) ? a : b; } /* Vector3.Max component */
/* Math.Max(float,float): IEEE 754:2019 maximum (NaN-propagating, +0 > -0). */
static float ieee_max(float a, float b)
{
    if (a != b) {
        if (!isnan(a)) return b < a ? a : b;
        return a;
    }
    return signbit(b) ? a : b;
}
/* VectorOps.Mod, VectorData.cs:697-698 */
static float floor_mod(float a, float b) { return a - b * floorf(a / b); }
/* VectorOps.VMax, VectorData.cs:860-861 */
static float vmax3(float x, float y, float z) { return ieee_max(ieee_max(x, y), z); }

/* ======================================================================== */
/* SDF catalogue                                                              */
/* ======================================================================== */
static void color_fn(int id, const float* prm, const float idx[3], float rgb[3])
{
    if (id == OCF_README) {
        /* (i,p,d) => 0.9f*Vector3.One - Vector3.Abs(i)/6f   README.md:24-30 */
        for (int k = 0; k < 3; k++) rgb[k] = 0.9f * 1.0f - fabsf(idx[k]) / 6.0f;
    } else {
        rgb[0] = prm[0]; rgb[1] = prm[1]; rgb[2] = prm[2];
    }
}

static void box_dist(const float p[3], const float b[3], float* w)
{
    /* Sdf.cs:134-136 / Sdf.cs:223-225 / SdfExpr.cs:20-23 */
    float wx = fabsf(p[0]) - b[0], wy = fabsf(p[1]) - b[1], wz = fabsf(p[2]) - b[2];
    *w = v3_length(sel_max(wx, 0.0f), sel_max(wy, 0.0f), sel_max(wz, 0.0f)) +
         vmax3(sel_min(wx, 0.0f), sel_min(wy, 0.0f), sel_min(wz, 0.0f));
}

void orc_eval(const osc_node* nodes, int root, const float p[3], float out[4])
{
    const osc_node* n = &nodes[root];
    switch (n->kind) {
    case OSC_SPHERE_W: /* d[i].W = p[i].Length() - radius; XYZ untouched */
        out[3] = v3_length(p[0], p[1], p[2]) - n->f[0];
        break;
    case OSC_BOX_W:
        box_dist(p, n->f, &out[3]);
        break;
    case OSC_PLANE_W: /* Vector3.Dot(p, normal) + distanceFromOrigin */
        out[3] = ((p[0] * n->f[0] + p[1] * n->f[1]) + p[2] * n->f[2]) + n->f[3];
        break;
    case OSC_SDF_WITHCOLOR:
    case OSC_F_WITHCOLOR:
        orc_eval(nodes, n->a, p, out);
        out[0] = n->f[0]; out[1] = n->f[1]; out[2] = n->f[2];
        break;
    case OSC_F_SPHERE:
        out[0] = n->f[1]; out[1] = n->f[2]; out[2] = n->f[3];
        out[3] = v3_length(p[0], p[1], p[2]) - n->f[0];
        break;
    case OSC_F_BOX:
        out[0] = out[1] = out[2] = 1.0f;
        box_dist(p, n->f, &out[3]);
        break;
    case OSC_F_CYLINDER: /* MathF.Max(MathF.Sqrt(p.X*p.X + p.Z*p.Z) - r, MathF.Abs(p.Y) - h) */
        out[0] = n->f[2]; out[1] = n->f[3]; out[2] = n->f[4];
        out[3] = ieee_max(sqrtf(p[0] * p[0] + p[2] * p[2]) - n->f[0], fabsf(p[1]) - n->f[1]);
        break;
    case OSC_F_UNION: {
        float da[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
        orc_eval(nodes, n->a, p, da);
        orc_eval(nodes, n->b, p, db);
        const float* r = (da[3] < db[3]) ? da : db;
        out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
        break;
    }
    case OSC_F_TRANSLATE: {
        float q[3] = {p[0] - n->f[0], p[1] - n->f[1], p[2] - n->f[2]};
        orc_eval(nodes, n->a, q, out);
        break;
    }
    case OSC_F_REPEAT_X: {
        float s = n->f[0];
        float q[3] = {floor_mod(p[0] + s * 0.5f, s) - s * 0.5f, p[1], p[2]};
        orc_eval(nodes, n->a, q, out);
        break;
    }
    case OSC_F_REPEAT_Y: {
        float s = n->f[0];
        float q[3] = {p[0], floor_mod(p[1] + s * 0.5f, s) - s * 0.5f, p[2]};
        orc_eval(nodes, n->a, q, out);
        break;
    }
    case OSC_F_REPEAT_XY: {
        float sx = n->f[0], sy = n->f[1];
        float q[3] = {floor_mod(p[0] + sx * 0.5f, sx) - sx * 0.5f,
                      floor_mod(p[1] + sy * 0.5f, sy) - sy * 0.5f, p[2]};
        orc_eval(nodes, n->a, q, out);
        break;
    }
    case OSC_F_REPEAT_XY_IDX: { /* ModifyInputAndOutput, SdfExpr.cs:113-141 */
        float sx = n->f[0], sy = n->f[1];
        float q[3] = {floor_mod(p[0] + sx * 0.5f, sx) - sx * 0.5f,
                      floor_mod(p[1] + sy * 0.5f, sy) - sy * 0.5f, p[2]};
        float idx[3] = {floorf((p[0] + sx * 0.5f) / sx), floorf((p[1] + sy * 0.5f) / sy), 0.0f};
        float d[4] = {0, 0, 0, 0};
        orc_eval(nodes, n->a, q, d);
        color_fn(n->b, &n->f[2], idx, out);
        out[3] = d[3];
        break;
    }
    case OSC_F_REPEAT_XZ_IDX: {
        float sx = n->f[0], sz = n->f[1];
        float q[3] = {floor_mod(p[0] + sx * 0.5f, sx) - sx * 0.5f, p[1],
                      floor_mod(p[2] + sz * 0.5f, sz) - sz * 0.5f};
        float idx[3] = {floorf((p[0] + sx * 0.5f) / sx), 0.0f, floorf((p[2] + sz * 0.5f) / sz)};
        float d[4] = {0, 0, 0, 0};
        orc_eval(nodes, n->a, q, d);
        color_fn(n->b, &n->f[2], idx, out);
        out[3] = d[3];
        break;
    }
    case OSC_F_CONST:
        out[0] = n->f[0]; out[1] = n->f[1]; out[2] = n->f[2]; out[3] = n->f[3];
        break;
    default:
        out[3] = NAN;
    }
}

/* ======================================================================== */
/* Voxels                                                                     */
/* ======================================================================== */
void orc_cell_size(const float min[3], const float max[3], int nx, int ny, int nz, float d[3])
{
    /* Voxels.cs:32-34 */
    d[0] = nx >= 1 ? (max[0] - min[0]) / (float)nx : 0.0f;
    d[1] = ny >= 1 ? (max[1] - min[1]) / (float)ny : 0.0f;
    d[2] = nz >= 1 ? (max[2] - min[2]) / (float)nz : 0.0f;
}

void orc_sample_position(const float min[3], const float max[3], int nx, int ny, int nz,
                         int64_t i, float p[3])
{
    float d[3];
    orc_cell_size(min, max, nx, ny, nz, d);
    /* Voxels.cs:81: min += (0.5f*DX, 0.5f*DY, 0.5f*DZ) */
    float m0 = min[0] + 0.5f * d[0], m1 = min[1] + 0.5f * d[1], m2 = min[2] + 0.5f * d[2];
    /* Voxels.cs:101-106: x fastest */
    int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((int64_t)nx * ny));
    p[0] = m0 + (float)ix * d[0];
    p[1] = m1 + (float)iy * d[1];
    p[2] = m2 + (float)iz * d[2];
}

int orc_batch_sizes(int ntotal, int batch_size, int* sizes, int cap)
{
    /* Voxels.cs:83,96-97 */
    int nb = (ntotal + batch_size - 1) / batch_size;
    for (int ib = 0; ib < nb && ib < cap; ib++) {
        int s = ib * batch_size;
        int e = s + batch_size < ntotal ? s + batch_size : ntotal;
        sizes[ib] = e - s;
    }
    return nb;
}

typedef struct {
    const osc_node* nodes; int root;
    float m[3], d[3];
    int nx, ny, nz, batch, nbatches;
    int64_t ntotal;
    float* values; float* colors;
    volatile int next;
} sample_job;

static void* sample_worker(void* arg)
{
    sample_job* j = (sample_job*)arg;
    /* thread-local scratch, zero-initialised ONCE per worker (Voxels.cs:88-92) */
    float* pos = (float*)calloc((size_t)j->batch * 3, sizeof(float));
    float* val = (float*)calloc((size_t)j->batch * 4, sizeof(float));
    for (;;) {
        int ib = __sync_fetch_and_add(&j->next, 1);
        if (ib >= j->nbatches) break;
        int64_t s = (int64_t)ib * j->batch;
        int64_t e = s + j->batch < j->ntotal ? s + j->batch : j->ntotal;
        for (int64_t i = s; i < e; i++) { /* Voxels.cs:99-108 */
            int ix = (int)(i % j->nx), iy = (int)((i / j->nx) % j->ny);
            int iz = (int)(i / ((int64_t)j->nx * j->ny));
            float* p = pos + (i - s) * 3;
            p[0] = j->m[0] + (float)ix * j->d[0];
            p[1] = j->m[1] + (float)iy * j->d[1];
            p[2] = j->m[2] + (float)iz * j->d[2];
        }
        for (int64_t i = s; i < e; i++) /* sdf(pmem, vmem), Voxels.cs:111 */
            orc_eval(j->nodes, j->root, pos + (i - s) * 3, val + (i - s) * 4);
        for (int64_t i = s; i < e; i++) { /* scatter, Voxels.cs:112-120 */
            int ix = (int)(i % j->nx), iy = (int)((i / j->nx) % j->ny);
            int iz = (int)(i / ((int64_t)j->nx * j->ny));
            size_t o = ((size_t)ix * j->ny + iy) * j->nz + iz;
            const float* v = val + (i - s) * 4;
            j->values[o] = v[3];
            if (j->colors) { j->colors[o * 3] = v[0]; j->colors[o * 3 + 1] = v[1]; j->colors[o * 3 + 2] = v[2]; }
        }
    }
    free(pos); free(val);
    return NULL;
}

int orc_hardware_threads(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

int orc_sample(const osc_node* nodes, int root, const float min[3], const float max[3],
               int nx, int ny, int nz, int batch_size, int nthreads,
               float* values, float* colors)
{
    sample_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes; j.root = root;
    orc_cell_size(min, max, nx, ny, nz, j.d);
    for (int k = 0; k < 3; k++) j.m[k] = min[k] + 0.5f * j.d[k];
    j.nx = nx; j.ny = ny; j.nz = nz; j.batch = batch_size;
    j.ntotal = (int64_t)nx * ny * nz;
    j.nbatches = (int)((j.ntotal + batch_size - 1) / batch_size);
    j.values = values; j.colors = colors; j.next = 0;
    if (nthreads <= 0) nthreads = orc_hardware_threads();
    if (nthreads > j.nbatches) nthreads = j.nbatches > 0 ? j.nbatches : 1;
    if (nthreads == 1) {
        sample_worker(&j);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, sample_worker, &j);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    return j.nbatches;
}

/* Voxels.SampleSdf restricted to the planes [z0, z0 + nzw) of the nx*ny*nz grid: the value of a voxel is a pure function of
   its index (Voxels.cs:99-108: p = (min + D/2) + (ix, iy, iz) * D, evaluated once per voxel), so a window is computed
   voxel by voxel with exactly the arithmetic of the batched sampler above (test infrastructure for 1024^3 grids;
   tests/test_oracle_window.py pins it against orc_sample).  values / colors: [nx][ny][nzw]. */
typedef struct { const osc_node* nodes; int root; float m[3], d[3]; int nx, ny, nzw, z0; float* values; float* colors; int next; } window_job;
static void* window_worker(void* arg)
{
    window_job* j = (window_job*)arg;
    for (;;) {
        const int ix = __sync_fetch_and_add(&j->next, 1);
        if (ix >= j->nx) break;
        for (int iy = 0; iy < j->ny; iy++)
            for (int k = 0; k < j->nzw; k++) {
                float p[3], v[4] = {0.0f, 0.0f, 0.0f, 0.0f};   /* zero-initialised scratch, Voxels.cs:88-92 */
                p[0] = j->m[0] + (float)ix * j->d[0];
                p[1] = j->m[1] + (float)iy * j->d[1];
                p[2] = j->m[2] + (float)(j->z0 + k) * j->d[2];
                orc_eval(j->nodes, j->root, p, v);
                const size_t o = ((size_t)ix * j->ny + iy) * j->nzw + k;
                j->values[o] = v[3];
                if (j->colors) { j->colors[o * 3] = v[0]; j->colors[o * 3 + 1] = v[1]; j->colors[o * 3 + 2] = v[2]; }
            }
    }
    return NULL;
}

void orc_sample_window(const osc_node* nodes, int root, const float min[3], const float max[3], int nx, int ny, int nz,
                       int z0, int nzw, int nthreads, float* values, float* colors)
{
    window_job j;
    memset(&j, 0, sizeof j);
    j.nodes = nodes; j.root = root;
    orc_cell_size(min, max, nx, ny, nz, j.d);
    for (int k = 0; k < 3; k++) j.m[k] = min[k] + 0.5f * j.d[k];
    j.nx = nx; j.ny = ny; j.nzw = nzw; j.z0 = z0; j.values = values; j.colors = colors;
    if (nthreads <= 0) nthreads = orc_hardware_threads();
    if (nthreads > nx) nthreads = nx;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, window_worker, &j);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
}

/* Voxels.ClipToBounds (Voxels.cs:133-167) on the planes [z0, z0 + nzw) of the nx*ny*nz grid */
void orc_clip_window(float* values, int nx, int ny, int nz, int z0, int nzw, const float min[3], const float max[3])
{
    float outside = (max[0] - min[0]) / (float)nx;
#define VOXW(ix, iy, k) values[((size_t)(ix) * ny + (iy)) * nzw + (k)]
    for (int iy = 0; iy < ny; iy++)
        for (int k = 0; k < nzw; k++) { VOXW(0, iy, k) = outside; VOXW(nx - 1, iy, k) = outside; }
    for (int ix = 0; ix < nx; ix++)
        for (int k = 0; k < nzw; k++) { VOXW(ix, 0, k) = outside; VOXW(ix, ny - 1, k) = outside; }
    for (int ix = 0; ix < nx; ix++)
        for (int iy = 0; iy < ny; iy++) {
            if (z0 == 0) VOXW(ix, iy, 0) = outside;
            if (z0 + nzw == nz) VOXW(ix, iy, nzw - 1) = outside;
        }
#undef VOXW
}

void orc_clip_to_bounds(float* values, int nx, int ny, int nz, const float min[3], const float max[3])
{
    /* Voxels.cs:133-167; outsideValue = Size.X / NX for all six faces */
    float outside = (max[0] - min[0]) / (float)nx;
#define VOX(ix, iy, iz) values[((size_t)(ix) * ny + (iy)) * nz + (iz)]
    for (int iy = 0; iy < ny; iy++)
        for (int iz = 0; iz < nz; iz++) { VOX(0, iy, iz) = outside; VOX(nx - 1, iy, iz) = outside; }
    for (int ix = 0; ix < nx; ix++)
        for (int iz = 0; iz < nz; iz++) { VOX(ix, 0, iz) = outside; VOX(ix, ny - 1, iz) = outside; }
    for (int ix = 0; ix < nx; ix++)
        for (int iy = 0; iy < ny; iy++) { VOX(ix, iy, 0) = outside; VOX(ix, iy, nz - 1) = outside; }
#undef VOX
}

/* ======================================================================== */
/* Marching cubes (Lewiner) -- MarchingCubes.cs + Cell.cs                    */
/* ======================================================================== */
typedef struct { float* p; size_t n, cap; } fbuf;
typedef struct { int32_t* p; size_t n, cap; } ibuf;
static void fpush3(fbuf* b, float x, float y, float z)
{
    if (b->n + 3 > b->cap) { b->cap = b->cap ? b->cap * 2 : 1024; b->p = (float*)realloc(b->p, b->cap * sizeof(float)); }
    b->p[b->n++] = x; b->p[b->n++] = y; b->p[b->n++] = z;
}
static void ipush(ibuf* b, int32_t v)
{
    if (b->n + 1 > b->cap) { b->cap = b->cap ? b->cap * 2 : 1024; b->p = (int32_t*)realloc(b->p, b->cap * sizeof(int32_t)); }
    b->p[b->n++] = v;
}

struct orc_mesh {
    fbuf grid_verts, grid_norms, verts, cols, norms;
    ibuf tris, cells;
    float bmin[3], bmax[3];
    int64_t impossible13;
};

typedef struct {
    int nx, ny, nz, step;
    int x, y, z;
    double v[8];        /* corner values, iso subtracted; corner order v0..v7 (Cell.cs:74) */
    float c[8][3];      /* corner colours */
    double vv[8];       /* bit-ordered copy (Cell.cs:453-460) */
    float cc[8][3];
    double vg[8][3];    /* corner gradients (Cell.cs:491-498) */
    int v12_done;
    double v12g[3];
    float v12p[3], v12c[3];
    int32_t* layer_lo;  /* faceLayer1 */
    int32_t* layer_hi;  /* faceLayer2 */
    orc_mesh* m;
} cell_t;

/* MarchingCubes.cs:376-407 */
static int test_face(const double* v, int face)
{
    int af = face < 0 ? -face : face;
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
    double acbd = A * C - B * D;
    if (acbd > -EPS && acbd < EPS) return face >= 0;
    return (double)face * A * acbd >= 0;
}

/* MarchingCubes.cs:412-546 */
static int test_internal(const double* v, int cas, int config, int subconfig, int s)
{
    double t, At = 0, Bt = 0, Ct = 0, Dt = 0, a, b;
    int test = 0, edge = -1;
    if (cas == 4 || cas == 10) {
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
        t = -b / (2 * a + EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v[0] + (v[4] - v[0]) * t;
        Bt = v[3] + (v[7] - v[3]) * t;
        Ct = v[2] + (v[6] - v[2]) * t;
        Dt = v[1] + (v[5] - v[1]) * t;
    } else if (cas == 6 || cas == 7 || cas == 12 || cas == 13) {
        if (cas == 6) edge = L2D(test6, config, 2);
        else if (cas == 7) edge = L2D(test7, config, 4);
        else if (cas == 12) edge = L2D(test12, config, 3);
        else edge = LUT[ROW3(tiling13_5_1, config, subconfig) + 0];
        /* per reference edge: t along the edge, then the three lerps on the
           parallel edges (MarchingCubes.cs:440-511).  Table: {a, b, B0,B1, C0,C1, D0,D1}
           meaning t = v[a]/(v[a]-v[b]+eps); Bt = v[B0]+(v[B1]-v[B0])*t; ... */
        static const int8_t E[12][8] = {
            {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
            {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
            {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
            {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
        if (edge >= 0 && edge < 12) {
            const int8_t* e = E[edge];
            t = v[e[0]] / (v[e[0]] - v[e[1]] + EPS);
            At = 0;
            Bt = v[e[2]] + (v[e[3]] - v[e[2]]) * t;
            Ct = v[e[4]] + (v[e[5]] - v[e[4]]) * t;
            Dt = v[e[6]] + (v[e[7]] - v[e[6]]) * t;
        }
        /* else: reference prints "Invalid edge" and proceeds with zeros */
    }
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    switch (test) { /* MarchingCubes.cs:526-545 */
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12: return s > 0;
    case 5: if (At * Ct - Bt * Dt < EPS) return s > 0; break;
    case 10: if (At * Ct - Bt * Dt >= EPS) return s > 0; break;
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    }
    return s < 0;
}

/* MarchingCubes.cs:94-371 ("TheBigSwitch") as a pure decision function: returns
   the case index; *lut_offset = start of the chosen triangle row in the blob,
   *nt = number of triangles (0 when nothing is emitted). */
int orc_resolve_tiling(const double v[8], int* lut_offset, int* nt)
{
    int index = 0;
    for (int k = 0; k < 8; k++) if (v[k] > 0.0) index |= 1 << k; /* Cell.cs:220-229 */
    int cas = L2D(cases, index, 0), cfg = L2D(cases, index, 1);
    int off = -1, n = 0, sub = 0;
    switch (cas) {
    case 1: off = ROW2(tiling1, cfg); n = 1; break;
    case 2: off = ROW2(tiling2, cfg); n = 2; break;
    case 3:
        if (test_face(v, L1D(test3, cfg))) { off = ROW2(tiling3_2, cfg); n = 4; }
        else { off = ROW2(tiling3_1, cfg); n = 2; }
        break;
    case 4:
        if (test_internal(v, cas, cfg, 0, L1D(test4, cfg))) { off = ROW2(tiling4_1, cfg); n = 2; }
        else { off = ROW2(tiling4_2, cfg); n = 6; }
        break;
    case 5: off = ROW2(tiling5, cfg); n = 3; break;
    case 6:
        if (test_face(v, L2D(test6, cfg, 0))) { off = ROW2(tiling6_2, cfg); n = 5; }
        else if (test_internal(v, cas, cfg, 0, L2D(test6, cfg, 1))) { off = ROW2(tiling6_1_1, cfg); n = 3; }
        else { off = ROW2(tiling6_1_2, cfg); n = 9; }
        break;
    case 7:
        if (test_face(v, L2D(test7, cfg, 0))) sub += 1;
        if (test_face(v, L2D(test7, cfg, 1))) sub += 2;
        if (test_face(v, L2D(test7, cfg, 2))) sub += 4;
        switch (sub) {
        case 0: off = ROW2(tiling7_1, cfg); n = 3; break;
        case 1: off = ROW3(tiling7_2, cfg, 0); n = 5; break;
        case 2: off = ROW3(tiling7_2, cfg, 1); n = 5; break;
        case 3: off = ROW3(tiling7_3, cfg, 0); n = 9; break;
        case 4: off = ROW3(tiling7_2, cfg, 2); n = 5; break;
        case 5: off = ROW3(tiling7_3, cfg, 1); n = 9; break;
        case 6: off = ROW3(tiling7_3, cfg, 2); n = 9; break;
        case 7:
            if (test_internal(v, cas, cfg, sub, L2D(test7, cfg, 3))) { off = ROW2(tiling7_4_2, cfg); n = 9; }
            else { off = ROW2(tiling7_4_1, cfg); n = 5; }
            break;
        }
        break;
    case 8: off = ROW2(tiling8, cfg); n = 2; break;
    case 9: off = ROW2(tiling9, cfg); n = 4; break;
    case 10:
        if (test_face(v, L2D(test10, cfg, 0))) {
            if (test_face(v, L2D(test10, cfg, 1))) { off = ROW2(tiling10_1_1_, cfg); n = 4; }
            else { off = ROW2(tiling10_2, cfg); n = 8; }
        } else {
            if (test_face(v, L2D(test10, cfg, 1))) { off = ROW2(tiling10_2_, cfg); n = 8; }
            else if (test_internal(v, cas, cfg, 0, L2D(test10, cfg, 2))) { off = ROW2(tiling10_1_1, cfg); n = 4; }
            else { off = ROW2(tiling10_1_2, cfg); n = 8; }
        }
        break;
    case 11: off = ROW2(tiling11, cfg); n = 4; break;
    case 12:
        if (test_face(v, L2D(test12, cfg, 0))) {
            if (test_face(v, L2D(test12, cfg, 1))) { off = ROW2(tiling12_1_1_, cfg); n = 4; }
            else { off = ROW2(tiling12_2, cfg); n = 8; }
        } else {
            if (test_face(v, L2D(test12, cfg, 1))) { off = ROW2(tiling12_2_, cfg); n = 8; }
            else if (test_internal(v, cas, cfg, 0, L2D(test12, cfg, 2))) { off = ROW2(tiling12_1_1, cfg); n = 4; }
            else { off = ROW2(tiling12_1_2, cfg); n = 8; }
        }
        break;
    case 13: {
        for (int k = 0; k < 6; k++)
            if (test_face(v, L2D(test13, cfg, k))) sub += 1 << k;
        sub = L1D(subconfig13, sub);
        if (sub == 0) { off = ROW2(tiling13_1, cfg); n = 4; }
        else if (sub >= 1 && sub <= 6) { off = ROW3(tiling13_2, cfg, sub - 1); n = 6; }
        else if (sub >= 7 && sub <= 18) { off = ROW3(tiling13_3, cfg, sub - 7); n = 10; }
        else if (sub >= 19 && sub <= 22) { off = ROW3(tiling13_4, cfg, sub - 19); n = 12; }
        else if (sub >= 23 && sub <= 26) {
            int s5 = sub - 23;
            if (test_internal(v, cas, cfg, s5, L2D(test13, cfg, 6))) { off = ROW3(tiling13_5_1, cfg, s5); n = 6; }
            else { off = ROW3(tiling13_5_2, cfg, s5); n = 10; }
        }
        else if (sub >= 27 && sub <= 38) { off = ROW3(tiling13_3_, cfg, sub - 27); n = 10; }
        else if (sub >= 39 && sub <= 44) { off = ROW3(tiling13_2_, cfg, sub - 39); n = 6; }
        else if (sub == 45) { off = ROW2(tiling13_1_, cfg); n = 4; }
        else { off = -1; n = 0; } /* "Impossible case 13?" -- nothing emitted */
        break;
    }
    case 14: off = ROW2(tiling14, cfg); n = 4; break;
    default: break; /* case 0 */
    }
    *lut_offset = off; *nt = n;
    return index;
}

/* Cell.cs:447-499 */
static void prepare(cell_t* c)
{
    static const int perm[8] = {0, 1, 3, 2, 4, 5, 7, 6};
    for (int k = 0; k < 8; k++) {
        c->vv[k] = c->v[perm[k]];
        memcpy(c->cc[k], c->c[perm[k]], sizeof(float) * 3);
    }
    const double* v = c->v;
    double g[8][3] = {
        {v[0] - v[1], v[0] - v[3], v[0] - v[4]}, {v[0] - v[1], v[1] - v[2], v[1] - v[5]},
        {v[3] - v[2], v[1] - v[2], v[2] - v[6]}, {v[3] - v[2], v[0] - v[3], v[3] - v[7]},
        {v[4] - v[5], v[4] - v[7], v[0] - v[4]}, {v[4] - v[5], v[5] - v[6], v[1] - v[5]},
        {v[7] - v[6], v[5] - v[6], v[2] - v[6]}, {v[7] - v[6], v[4] - v[7], v[3] - v[7]}};
    memcpy(c->vg, g, sizeof g);
}

/* Cell.cs:501-549 */
static void center_vertex(cell_t* c)
{
    double w[8], fx = 0, fy = 0, fz = 0, ff = 0;
    static const double ox[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    static const double oy[8] = {0, 0, 1, 1, 0, 0, 1, 1};
    static const double oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
    for (int k = 0; k < 8; k++) w[k] = 1.0 / (EPS + fabs(c->v[k]));
    for (int k = 0; k < 8; k++) { fx += ox[k] * w[k]; fy += oy[k] * w[k]; fz += oz[k] * w[k]; ff += w[k]; }
    float fc[3];
    for (int j = 0; j < 3; j++) { /* Vector3 float mul-adds, left to right (Cell.cs:526) */
        float acc = c->c[0][j] * (float)w[0];
        for (int k = 1; k < 8; k++) acc = acc + c->c[k][j] * (float)w[k];
        fc[j] = acc;
    }
    double stp = (double)c->step;
    c->v12p[0] = (float)(c->x + stp * fx / ff);
    c->v12p[1] = (float)(c->y + stp * fy / ff);
    c->v12p[2] = (float)(c->z + stp * fz / ff);
    for (int j = 0; j < 3; j++) c->v12c[j] = (float)(fc[j] / ff);
    for (int j = 0; j < 3; j++) {
        double s = w[0] * c->vg[0][j];
        for (int k = 1; k < 8; k++) s = s + w[k] * c->vg[k][j];
        c->v12g[j] = s;
    }
    c->v12_done = 1;
}

/* Cell.cs:371-441: slot in the two face layers */
static int32_t* face_slot(cell_t* c, int vi)
{
    int i = c->nx * c->y + c->x, j = 0;
    int32_t* layer = c->layer_lo;
    if (vi < 8) {
        if (vi >= 4) { vi -= 4; layer = c->layer_hi; }
        if (vi == 1) { i += c->step; j = 1; }
        else if (vi == 2) { i += c->nx * c->step; }
        else if (vi == 3) { j = 1; }
    } else if (vi < 12) {
        j = 2;
        if (vi == 9) i += c->step;
        else if (vi == 10) i += c->nx * c->step + c->step;
        else if (vi == 11) i += c->nx * c->step;
    } else {
        j = 3;
    }
    return &layer[4 * i + j];
}

static void add_gradient(cell_t* c, int32_t vi, double gx, double gy, double gz)
{
    /* Cell.cs:154-155: float32 Vector3 accumulate */
    float* n = c->m->norms.p + (size_t)vi * 3;
    n[0] = n[0] + (float)gx; n[1] = n[1] + (float)gy; n[2] = n[2] + (float)gz;
}

/* Cell.cs:272-359 */
static void add_face_from_edge(cell_t* c, int vi)
{
    orc_mesh* m = c->m;
    int32_t* slot = face_slot(c, vi);
    int32_t idx = *slot;
    double stp = (double)c->step;
    if (vi == 12) {
        if (!c->v12_done) center_vertex(c);
        if (idx < 0) {
            idx = (int32_t)(m->grid_verts.n / 3);
            fpush3(&m->grid_verts, c->v12p[0], c->v12p[1], c->v12p[2]);
            fpush3(&m->cols, c->v12c[0], c->v12c[1], c->v12c[2]);
            fpush3(&m->norms, 0, 0, 0);
            *slot = idx;
        }
        ipush(&m->tris, idx);
        add_gradient(c, idx, c->v12g[0], c->v12g[1], c->v12g[2]);
        return;
    }
    int dx1 = L2D(edgesrelx, vi, 0), dx2 = L2D(edgesrelx, vi, 1);
    int dy1 = L2D(edgesrely, vi, 0), dy2 = L2D(edgesrely, vi, 1);
    int dz1 = L2D(edgesrelz, vi, 0), dz2 = L2D(edgesrelz, vi, 1);
    int i1 = dz1 * 4 + dy1 * 2 + dx1, i2 = dz2 * 4 + dy2 * 2 + dx2;
    double w1 = 1.0 / (EPS + fabs(c->vv[i1]));
    double w2 = 1.0 / (EPS + fabs(c->vv[i2]));
    if (idx < 0) {
        double fx = 0, fy = 0, fz = 0, ff = 0;
        fx += dx1 * w1; fy += dy1 * w1; fz += dz1 * w1; ff += w1;
        fx += dx2 * w2; fy += dy2 * w2; fz += dz2 * w2; ff += w2;
        float col[3];
        for (int j = 0; j < 3; j++) col[j] = c->cc[i1][j] * (float)w1 + c->cc[i2][j] * (float)w2;
        idx = (int32_t)(m->grid_verts.n / 3);
        fpush3(&m->grid_verts, (float)(c->x + stp * fx / ff), (float)(c->y + stp * fy / ff),
               (float)(c->z + stp * fz / ff));
        fpush3(&m->cols, (float)(col[0] / ff), (float)(col[1] / ff), (float)(col[2] / ff));
        fpush3(&m->norms, 0, 0, 0);
        *slot = idx;
    }
    ipush(&m->tris, idx);
    /* NB (Cell.cs:157-158,332-333): vg[] is filled in CORNER order but indexed here with
       the BIT-order index -- inherited quirk, reproduced. */
    add_gradient(c, idx, c->vg[i1][0] * w1, c->vg[i1][1] * w1, c->vg[i1][2] * w1);
    add_gradient(c, idx, c->vg[i2][0] * w2, c->vg[i2][1] * w2, c->vg[i2][2] * w2);
}

static void measure(orc_mesh* m)
{
    /* Mesh.cs:30-45 */
    size_t nv = m->verts.n / 3;
    if (nv == 0) return;
    float mn[3], mx[3];
    memcpy(mn, m->verts.p, sizeof mn); memcpy(mx, mn, sizeof mx);
    for (size_t i = 1; i < nv; i++)
        for (int j = 0; j < 3; j++) {
            float v = m->verts.p[i * 3 + j];
            mn[j] = sel_min(mn[j], v); mx[j] = sel_max(mx[j], v);
        }
    memcpy(m->bmin, mn, sizeof mn); memcpy(m->bmax, mx, sizeof mx);
}

/* The sweep on the planes [z0, z0 + nz) of a grid of nz_global planes (z0 = 0, nz_global = nz: the whole volume, which is
   all the reference has).  Test infrastructure for grids whose whole-volume sweep does not fit a test (1024^3): the cells
   of a window are the cells of the whole sweep, vertex z coordinates are the global ones (the window's first plane is
   plane z0: Cell.cs:345-347 adds the cell's z) and the final transform is the whole grid's (MarchingCubes.cs:85-90 uses
   the volume's N - 1).  What differs from the whole sweep is only what the window's first layer sees below it, so a
   caller compares layers from the window's third one on (tests/test_oracle_window.py pins that against orc_march). */
static orc_mesh* march_impl(const float* values, const float* colors, int nx, int ny, int nz, int z0, int nz_global,
                            const float min[3], const float max[3], float iso, int step,
                            orc_progress_fn progress, void* user)
{
    orc_mesh* m = (orc_mesh*)calloc(1, sizeof(orc_mesh));
    cell_t c;
    memset(&c, 0, sizeof c);
    c.nx = nx; c.ny = ny; c.nz = nz; c.step = step; c.m = m;
    size_t ln = (size_t)nx * ny * 4;
    c.layer_lo = (int32_t*)malloc(ln * sizeof(int32_t));
    c.layer_hi = (int32_t*)malloc(ln * sizeof(int32_t));
    for (size_t i = 0; i < ln; i++) { c.layer_lo[i] = -1; c.layer_hi[i] = -1; }
    const double isod = (double)iso;
    const int xb = nx - 2 * step, yb = ny - 2 * step, zb = nz - 2 * step;
#define VAL(ix, iy, iz) values[((size_t)(ix) * ny + (iy)) * nz + (iz)]
    /* MarchingCubes.cs:53-82 */
    for (int z = -step; z < zb;) {
        z += step;
        int zs = z + step;
        { /* Cell.NewZValue, Cell.cs:173-182 */
            int32_t* t = c.layer_lo; c.layer_lo = c.layer_hi; c.layer_hi = t;
            for (size_t i = 0; i < ln; i++) c.layer_hi[i] = -1;
        }
        for (int y = -step; y < yb;) {
            y += step;
            int ys = y + step;
            for (int x = -step; x < xb;) {
                x += step;
                int xs = x + step;
                const int cx[8] = {x, xs, xs, x, x, xs, xs, x};
                const int cy[8] = {y, y, ys, ys, y, y, ys, ys};
                const int cz[8] = {z, z, z, z, zs, zs, zs, zs};
                c.x = x; c.y = y; c.z = z + z0;
                for (int k = 0; k < 8; k++) { /* Cell.SetCube, Cell.cs:191-233 */
                    size_t o = ((size_t)cx[k] * ny + cy[k]) * nz + cz[k];
                    c.v[k] = (double)values[o] - isod;
                    if (colors) memcpy(c.c[k], colors + o * 3, sizeof(float) * 3);
                    else c.c[k][0] = c.c[k][1] = c.c[k][2] = 0.0f;
                }
                c.v12_done = 0;
                int off, nt;
                int index = orc_resolve_tiling(c.v, &off, &nt);
                int cas = L2D(cases, index, 0);
                if (cas > 0) {
                    ipush(&m->cells, (int32_t)(((int64_t)z * ny + y) * nx + x));
                    ipush(&m->cells, index); ipush(&m->cells, off); ipush(&m->cells, nt);
                    if (off < 0) { m->impossible13++; continue; }
                    prepare(&c); /* Cell.AddTriangles[2], Cell.cs:238-265 */
                    for (int k = 0; k < nt * 3; k++) add_face_from_edge(&c, LUT[off + k]);
                }
            }
        }
        if (progress) progress((float)z / (float)zb, user);
    }
#undef VAL
    free(c.layer_lo); free(c.layer_hi);

    /* Cell.NegativeNormals (Cell.cs:97-109) then Mesh.Transform (MarchingCubes.cs:85-90,
       Mesh.cs:47-64). */
    float size[3] = {max[0] - min[0], max[1] - min[1], max[2] - min[2]};
    float center[3] = {(min[0] + max[0]) * 0.5f, (min[1] + max[1]) * 0.5f, (min[2] + max[2]) * 0.5f};
    int nn[3] = {nx, ny, nz_global};
    float sc[3], tr[3];
    for (int j = 0; j < 3; j++) {
        float t1 = (float)(-(nn[j] - 1)) / 2.0f;
        sc[j] = size[j] / (float)(nn[j] - 1);
        tr[j] = t1 * sc[j] + center[j];
    }
    /* Matrix4x4.Invert of diag(sx,sy,sz,1), cofactor form */
    float det = sc[0] * (sc[1] * sc[2]);
    float inv_det = 1.0f / det;
    float in[3] = {(sc[1] * sc[2]) * inv_det, (sc[0] * sc[2]) * inv_det, (sc[0] * sc[1]) * inv_det};
    size_t nv = m->grid_verts.n / 3;
    for (size_t i = 0; i < nv; i++) {
        const float* g = m->grid_verts.p + i * 3;
        fpush3(&m->verts, g[0] * sc[0] + tr[0], g[1] * sc[1] + tr[1], g[2] * sc[2] + tr[2]);
        float* n = m->norms.p + i * 3;
        float len = v3_length(n[0], n[1], n[2]);
        float q[3] = {-(n[0] / len), -(n[1] / len), -(n[2] / len)};
        fpush3(&m->grid_norms, q[0], q[1], q[2]);   /* Cell.NegativeNormals as the Mesh constructor receives them */
        float t[3] = {q[0] * in[0], q[1] * in[1], q[2] * in[2]};
        float tl = v3_length(t[0], t[1], t[2]);
        n[0] = t[0] / tl; n[1] = t[1] / tl; n[2] = t[2] / tl;
    }
    measure(m);
    return m;
}

/* Mesh.Transform(Matrix4x4), Mesh.cs:47-64, on plain arrays: vertices by Vector3.Transform, normals by
   Vector3.TransformNormal with transpose(inverse(M with its translation row zeroed)) and Vector3.Normalize; then
   Mesh.Measure (bmin / bmax; untouched when n == 0).  System.Numerics arithmetic as documented in sdfk_oracle.h:
   row-vector convention, products summed left to right, one rounding per operation. */
int orc_transform_arrays(float* verts, float* norms, int64_t n, const float M[16], float bmin[3], float bmax[3])
{
    float nm[16], inv[16], nt[16];
    memcpy(nm, M, sizeof nm);
    nm[12] = 0.0f; nm[13] = 0.0f; nm[14] = 0.0f; nm[15] = 1.0f;          /* M41 M42 M43 M44 */
    if (!orc_mat_invert(nm, inv))                                         /* Matrix4x4.Invert failed: NaN matrix (BCL) */
        for (int k = 0; k < 16; k++) inv[k] = NAN;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) nt[r * 4 + c] = inv[c * 4 + r];
    for (int64_t i = 0; i < n; i++) {
        float* v = verts + i * 3;
        float* q = norms + i * 3;
        const float x = v[0], y = v[1], z = v[2];
        v[0] = ((x * M[0] + y * M[4]) + z * M[8]) + M[12];
        v[1] = ((x * M[1] + y * M[5]) + z * M[9]) + M[13];
        v[2] = ((x * M[2] + y * M[6]) + z * M[10]) + M[14];
        const float a = q[0], b = q[1], c = q[2];
        const float tx = (a * nt[0] + b * nt[4]) + c * nt[8];
        const float ty = (a * nt[1] + b * nt[5]) + c * nt[9];
        const float tz = (a * nt[2] + b * nt[6]) + c * nt[10];
        const float len = v3_length(tx, ty, tz);
        q[0] = tx / len; q[1] = ty / len; q[2] = tz / len;
    }
    if (n > 0) {
        memcpy(bmin, verts, 3 * sizeof(float)); memcpy(bmax, verts, 3 * sizeof(float));
        for (int64_t i = 1; i < n; i++)
            for (int j = 0; j < 3; j++) {
                const float w = verts[i * 3 + j];
                bmin[j] = sel_min(bmin[j], w); bmax[j] = sel_max(bmax[j], w);
            }
    }
    return 1;
}

orc_mesh* orc_march(const float* values, const float* colors, int nx, int ny, int nz,
                    const float min[3], const float max[3], float iso, int step,
                    orc_progress_fn progress, void* user)
{
    return march_impl(values, colors, nx, ny, nz, 0, nz, min, max, iso, step, progress, user);
}

orc_mesh* orc_march_window(const float* values, const float* colors, int nx, int ny, int nz_window, int z0, int nz_global,
                           const float min[3], const float max[3], float iso)
{
    return march_impl(values, colors, nx, ny, nz_window, z0, nz_global, min, max, iso, 1, NULL, NULL);
}

int64_t orc_mesh_vertex_count(const orc_mesh* m) { return (int64_t)(m->verts.n / 3); }
int64_t orc_mesh_index_count(const orc_mesh* m) { return (int64_t)m->tris.n; }
const float* orc_mesh_vertices(const orc_mesh* m) { return m->verts.p; }
const float* orc_mesh_colors(const orc_mesh* m) { return m->cols.p; }
const float* orc_mesh_normals(const orc_mesh* m) { return m->norms.p; }
const float* orc_mesh_grid_vertices(const orc_mesh* m) { return m->grid_verts.p; }
const float* orc_mesh_grid_normals(const orc_mesh* m) { return m->grid_norms.p; }
const int32_t* orc_mesh_triangles(const orc_mesh* m) { return m->tris.p; }
int64_t orc_mesh_cell_count(const orc_mesh* m) { return (int64_t)(m->cells.n / 4); }
const int32_t* orc_mesh_cells(const orc_mesh* m) { return m->cells.p; }
int64_t orc_mesh_impossible13(const orc_mesh* m) { return m->impossible13; }
void orc_mesh_bounds(const orc_mesh* m, float mn[3], float mx[3])
{
    memcpy(mn, m->bmin, sizeof(float) * 3); memcpy(mx, m->bmax, sizeof(float) * 3);
}
void orc_mesh_free(orc_mesh* m)
{
    if (!m) return;
    free(m->grid_verts.p); free(m->grid_norms.p); free(m->verts.p); free(m->cols.p); free(m->norms.p);
    free(m->tris.p); free(m->cells.p); free(m);
}
