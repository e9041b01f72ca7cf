/* sdfk_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's hot path
 *     Voxels.SampleSdf -> (Voxels.ClipToBounds) -> MarchingCubes.CreateMesh
 * (praeclarum/SdfKit, C#).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path
 * (sdfkit_amd/, libsdfkit_hip.so) never links, imports or calls it.
 *
 * Pinning status: the reference is C#/.NET and cannot be built or run in this
 * environment, so this oracle is pinned against the reference's own known-answer
 * tests only (tests/test_oracle_golden.py): the ten exact vertex counts of
 * Tests/MarchingCubesTests.cs and Tests/SdfTests.cs, their AABB/centre
 * assertions, the colour inequality, the sampling-position and batch-slicing
 * tests of Tests/VolumeTests.cs.  Vertex positions beyond the AABB, triangle
 * index contents, normals, colour values, step>1, iso!=0 and the ambiguous /
 * centre-vertex tilings are NOT pinned by any reference TEST; they are pinned
 * (round 3) by vectors obtained from EXECUTING the reference's source for the
 * marching-cubes stage -- MarchingCubes.cs and Cell.cs run by an interpreter of
 * the C# subset they are written in (tools/cs_subset.py,
 * tools/gen_reference_vectors.py -> tests/golden/reference_meshes.npz,
 * tests/test_reference_vectors.py: bit-exact; likewise the per-point SDF catalogue
 * SdfFuncs / SdfFuncEx of Sdf.cs -> tests/golden/reference_sdf_points.npz against
 * orc_eval, and the whole path Voxels ctor -> SampleSdf -> ClipToBounds -> CreateMesh
 * -> tests/golden/reference_path.npz, where the executed source also yields every
 * vertex count the reference's own tests assert: 104, 54, 312, 0, 384, 384, 7456,
 * 72240, 1248, 1248).  The interpreter's numeric
 * semantics (IEEE float32 / float64, C#'s promotions) and the BCL pieces below
 * are ours: that much of the pin is a restatement, not the .NET runtime.
 *
 * Third-party arithmetic restated here (not under /root/reference): .NET BCL
 * System.Numerics (SDK pin 6.0.101, global.json): Vector3.Length =
 * sqrtf((x*x + y*y) + z*z); Vector3.Dot likewise left-to-right; Vector3.Min/Max =
 * compare-select; Math.Max/MathF.Max = IEEE-754:2019 maximum; Vector3.Normalize
 * = v / Length(v); Vector3.Transform / TransformNormal = row-vector products
 * evaluated left to right in float; Matrix4x4.Invert = cofactor expansion
 * (software path).  No FMA contraction anywhere (build with -ffp-contract=off).
 */
#ifndef SDFK_ORACLE_H
#define SDFK_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene description (mirrors the reference's SDF catalogue) ------------- */
enum {
    /* batched `Sdf` delegates that write only .W (Sdf.cs:120-214): colour stays
       at the zero-initialised scratch value (Voxels.cs:88-92). */
    OSC_SPHERE_W = 1,   /* f[0]=r                      Sdf.cs:202-214 */
    OSC_BOX_W = 2,      /* f[0..2]=bounds              Sdf.cs:125-139 */
    OSC_PLANE_W = 3,    /* f[0..2]=normal f[3]=dist    Sdf.cs:144-156 */
    OSC_SDF_WITHCOLOR = 4, /* a=child f[0..2]=rgb      Sdf.cs:101-110 */
    /* per-point SdfFunc / SdfExpr forms (colour + distance) */
    OSC_F_SPHERE = 10,  /* f[0]=r f[1..3]=rgb          Sdf.cs:232-238, SdfExpr.cs:47-51 */
    OSC_F_BOX = 11,     /* f[0..2]=bounds, rgb=One     Sdf.cs:219-228, SdfExpr.cs:18-24 */
    OSC_F_CYLINDER = 12,/* f[0]=r f[1]=h f[2..4]=rgb   SdfExpr.cs:28-32 */
    OSC_F_UNION = 13,   /* a,b children                Sdf.cs:240-248, SdfExpr.cs:54-68 */
    OSC_F_TRANSLATE = 14,/* a=child f[0..2]=offset     Sdf.cs:315-321 (p - offset) */
    OSC_F_WITHCOLOR = 15,/* a=child f[0..2]=rgb        Sdf.cs:328-335, SdfExpr.cs:143-147 */
    OSC_F_REPEAT_X = 16, /* a=child f[0]=sx            SdfExpr.cs:149-153 */
    OSC_F_REPEAT_Y = 17, /* a=child f[0]=sy            SdfExpr.cs:197-201 */
    OSC_F_REPEAT_XY = 18,/* a=child f[0]=sx f[1]=sy    SdfExpr.cs:155-161 */
    OSC_F_REPEAT_XY_IDX = 19, /* + colorfn in b, params f[2..]  SdfExpr.cs:163-178, Sdf.cs:267-282 */
    OSC_F_REPEAT_XZ_IDX = 20, /* f[0]=sx f[1]=sz       SdfExpr.cs:180-195, Sdf.cs:284-299 */
    OSC_F_CONST = 21    /* f[0..3] = r,g,b,w constant (VolumeTests.cs: const-1 SDF) */
};
/* colour modifier lambdas (i = repeat index, p = modified point, d = output) */
enum {
    OCF_CONST = 0,      /* colour = f[2..4] */
    OCF_README = 1      /* 0.9f*Vector3.One - Vector3.Abs(i)/6f   (README.md:24-30) */
};
typedef struct osc_node {
    int32_t kind;
    int32_t a, b;       /* child indices / colour-fn id */
    float f[8];
} osc_node;

/* evaluate one point (SdfFunc semantics); out = (r,g,b,w).  `scratch` is the prior
   content of the output slot (what a W-only delegate leaves in XYZ). */
void orc_eval(const osc_node* nodes, int root, const float p[3], float out[4]);

/* Voxels ctor cell size (Voxels.cs:32-34) */
void orc_cell_size(const float min[3], const float max[3], int nx, int ny, int nz, float d[3]);

/* Voxels.SampleSdf (Voxels.cs:72-125): values/colors are [nx][ny][nz] z-fastest.
   nthreads<=0 -> all cores.  Returns number of batches. colors may be NULL. */
int orc_sample(const osc_node* nodes, int root, const float min[3], const float max[3],
               int nx, int ny, int nz, int batch_size, int nthreads,
               float* values, float* colors);
/* sizes the delegate would be called with (VolumeTests.cs:109-135) */
int orc_batch_sizes(int ntotal, int batch_size, int* sizes, int cap);
/* sample position of linear index i (Voxels.cs:81,99-106) */
void orc_sample_position(const float min[3], const float max[3], int nx, int ny, int nz,
                         int64_t i, float p[3]);

/* Voxels.ClipToBounds (Voxels.cs:133-167) */
void orc_clip_to_bounds(float* values, int nx, int ny, int nz, const float min[3], const float max[3]);

typedef struct orc_mesh orc_mesh;
typedef void (*orc_progress_fn)(float value, void* user);

/* MarchingCubes.CreateMesh (MarchingCubes.cs:39-92) incl. Mesh ctor + Transform
   (Mesh.cs:21-64).  colors may be NULL (treated as zeros). */
orc_mesh* orc_march(const float* values, const float* colors, int nx, int ny, int nz,
                    const float min[3], const float max[3], float iso, int step,
                    orc_progress_fn progress, void* user);
/* Mesh.Transform(Matrix4x4) (Mesh.cs:47-64) on arrays, then Mesh.Measure. */
int orc_transform_arrays(float* verts, float* norms, int64_t n, const float M[16], float bmin[3], float bmax[3]);

/* Window forms (test infrastructure for grids whose whole-volume sweep does not fit a test, e.g. 1024^3): the planes
   [z0, z0 + nz_window) of an nx*ny*nz grid.  Sampling and clipping are per-voxel functions of the GLOBAL index; the sweep
   uses global z coordinates and the whole grid's final transform.  Pinned against the whole-volume functions by
   tests/test_oracle_window.py. */
void orc_sample_window(const osc_node* nodes, int root, const float min[3], const float max[3], int nx, int ny, int nz,
                       int z0, int nz_window, int nthreads, float* values, float* colors);
void orc_clip_window(float* values, int nx, int ny, int nz, int z0, int nz_window, const float min[3], const float max[3]);
orc_mesh* orc_march_window(const float* values, const float* colors, int nx, int ny, int nz_window, int z0, int nz_global,
                           const float min[3], const float max[3], float iso);
int64_t orc_mesh_vertex_count(const orc_mesh*);
int64_t orc_mesh_index_count(const orc_mesh*);
const float* orc_mesh_vertices(const orc_mesh*);   /* 3 floats per vertex, transformed */
const float* orc_mesh_colors(const orc_mesh*);
const float* orc_mesh_normals(const orc_mesh*);    /* transformed + normalised */
const float* orc_mesh_grid_vertices(const orc_mesh*); /* voxel-index units, before Transform */
const float* orc_mesh_grid_normals(const orc_mesh*);  /* Cell.NegativeNormals (Cell.cs:97-109), before Transform */
const int32_t* orc_mesh_triangles(const orc_mesh*);
void orc_mesh_bounds(const orc_mesh*, float min[3], float max[3]); /* Mesh.Measure */
/* per-cell debug stream: (case index, lut offset, nt) of every active cell in sweep order */
int64_t orc_mesh_cell_count(const orc_mesh*);
const int32_t* orc_mesh_cells(const orc_mesh*);    /* 4 ints per active cell: linear cell id, index, lutoff, nt */
int64_t orc_mesh_impossible13(const orc_mesh*);
void orc_mesh_free(orc_mesh*);

/* direct access to the decision functions, for table/branch tests */
int orc_resolve_tiling(const double v[8], int* lut_offset, int* nt); /* returns case index */

int orc_hardware_threads(void);

/* ---- RayMarcher (sdfk_oracle_ray.c: pinning status in its header) --- */
void orc_mat_look_at(const float pos[3], const float target[3], const float up[3], float m[16]);
void orc_mat_perspective_fov(float fov, float aspect, float nearp, float farp, float m[16]);
void orc_mat_mul(const float a[16], const float b[16], float out[16]);
int orc_mat_invert(const float m[16], float out[16]);
void orc_ray_camera(const float view[16], float fov_degrees, int width, int height, float nearp, float farp,
                    float cam_pos[3], float vp_inverse[16]);
void orc_raymarch(const osc_node* nodes, int root, int width, int height, const float cam_pos[3],
                  const float vp_inverse[16], float nearp, float farp, int iters, float* depth, float* rgb, int threads);
#ifdef __cplusplus
}
#endif
#endif
