/* sdfk_oracle_ray.c -- CPU ORACLE (test infrastructure) for SURVEY.md section 8(f) row 4:
 * RayMarcher.Render / RenderDepth (SdfKit/RayMarcher.cs:45-211), the sphere-tracing
 * renderer over the same Sdf catalogue.  Plain C, one pixel at a time; float32, one
 * rounding per operation, no FMA contraction (built with -ffp-contract=off).
 *
 * Pinning status: the reference's Tests/RayMarcherTests.cs pins RenderDepth at single
 * pixels with tolerances (SphereDepth/BoxDepth: 4.0 +- 1e-2 at the centre, > 9 at the
 * corner; PlaneDepth: 5.0 +- 1e-2, corner < 9; CylinderDepth: 5 - r +- 1e-1) --
 * tests/test_oracle_golden.py checks them.  Everything finer (the exact float of every
 * pixel, the shaded colours) is NOT pinned by any reference test: PARITY IS UNPINNED
 * there (oracle <-> HIP agreement only).
 *
 * Third-party arithmetic restated here (not under /root/reference): .NET 6 BCL
 * System.Numerics Matrix4x4.CreateLookAt / CreatePerspectiveFieldOfView / operator* /
 * Invert (software cofactor path), Vector4.Transform, Vector3.Transform,
 * Vector3.Normalize (= v / Length), MathF.Tan (libm tanf), MathF.Max (IEEE maximum).
 *
 * Known divergences from the reference, on purpose:
 *  - VectorOps.MulAdd(Vec3Data, FloatData, Vec3Data) has AVX/FMA fast paths
 *    (VectorData.cs:746-786) that leave the last n % 8 (n % 4) floats of every vertical
 *    partition unwritten; the scalar path `a*b + c` (VectorData.cs:788-797) is restated.
 *  - Render() partitions by Environment.ProcessorCount (RayMarcher.cs:50-61): the
 *    per-pixel arithmetic does not depend on it and no partitioning is done here.
 *  - W-only delegates (Sdfs.Sphere ...) leave XYZ of a pooled, uninitialised Vec4 buffer
 *    (RayMarcher.cs:213-218); the colour is taken as 0 here, as in Voxels.SampleSdf.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include "sdfk_oracle.h"

/* ---- System.Numerics.Matrix4x4, row-major m[0..15] = M11 M12 M13 M14 M21 ... M44 -------- */
static float v3len(const float v[3]) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static float v3dot(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void v3normalize(const float v[3], float o[3])
{
    const float l = v3len(v);
    o[0] = v[0] / l; o[1] = v[1] / l; o[2] = v[2] / l;
}
static void v3cross(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

void orc_mat_look_at(const float pos[3], const float target[3], const float up[3], float m[16])
{
    float d[3] = {pos[0] - target[0], pos[1] - target[1], pos[2] - target[2]}, z[3], c[3], x[3], y[3];
    v3normalize(d, z);
    v3cross(up, z, c);
    v3normalize(c, x);
    v3cross(z, x, y);
    m[0] = x[0]; m[1] = y[0]; m[2] = z[0]; m[3] = 0.0f;
    m[4] = x[1]; m[5] = y[1]; m[6] = z[1]; m[7] = 0.0f;
    m[8] = x[2]; m[9] = y[2]; m[10] = z[2]; m[11] = 0.0f;
    m[12] = -v3dot(x, pos); m[13] = -v3dot(y, pos); m[14] = -v3dot(z, pos); m[15] = 1.0f;
}

void orc_mat_perspective_fov(float fov, float aspect, float nearp, float farp, float m[16])
{
    const float yscale = 1.0f / tanf(fov * 0.5f);
    const float xscale = yscale / aspect;
    const float neg_far_range = isinf(farp) && farp > 0 ? -1.0f : farp / (nearp - farp);
    memset(m, 0, 16 * sizeof(float));
    m[0] = xscale;
    m[5] = yscale;
    m[10] = neg_far_range;
    m[11] = -1.0f;
    m[14] = nearp * neg_far_range;
}

void orc_mat_mul(const float a[16], const float b[16], float o[16])
{
    float r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
    memcpy(o, r, sizeof r);
}

int orc_mat_invert(const float M[16], float R[16])
{
    const float a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7];
    const float i = M[8], j = M[9], k = M[10], l = M[11], m = M[12], n = M[13], o = M[14], p = M[15];
    const float kp_lo = k * p - l * o, jp_ln = j * p - l * n, jo_kn = j * o - k * n;
    const float ip_lm = i * p - l * m, io_km = i * o - k * m, in_jm = i * n - j * m;
    const float a11 = +(f * kp_lo - g * jp_ln + h * jo_kn);
    const float a12 = -(e * kp_lo - g * ip_lm + h * io_km);
    const float a13 = +(e * jp_ln - f * ip_lm + h * in_jm);
    const float a14 = -(e * jo_kn - f * io_km + g * in_jm);
    const float det = a * a11 + b * a12 + c * a13 + d * a14;
    if (fabsf(det) < 1.1920929e-07f) {
        for (int q = 0; q < 16; q++) R[q] = NAN;
        return 0;
    }
    const float inv = 1.0f / det;
    R[0] = a11 * inv; R[4] = a12 * inv; R[8] = a13 * inv; R[12] = a14 * inv;
    R[1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv;
    R[5] = +(a * kp_lo - c * ip_lm + d * io_km) * inv;
    R[9] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv;
    R[13] = +(a * jo_kn - b * io_km + c * in_jm) * inv;
    const float gp_ho = g * p - h * o, fp_hn = f * p - h * n, fo_gn = f * o - g * n;
    const float ep_hm = e * p - h * m, eo_gm = e * o - g * m, en_fm = e * n - f * m;
    R[2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv;
    R[6] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv;
    R[10] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv;
    R[14] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv;
    const float gl_hk = g * l - h * k, fl_hj = f * l - h * j, fk_gj = f * k - g * j;
    const float el_hi = e * l - h * i, ek_gi = e * k - g * i, ej_fi = e * j - f * i;
    R[3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv;
    R[7] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv;
    R[11] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv;
    R[15] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv;
    return 1;
}

/* host part of RayMarcher.GetCameraRays (RayMarcher.cs:97-112): camera position and the
 * inverse view-projection the per-pixel part needs */
void orc_ray_camera(const float view[16], float fov_degrees, int width, int height, float nearp, float farp,
                    float cam_pos[3], float vp_inverse[16])
{
    float cam[16], proj[16], vp[16];
    orc_mat_invert(view, cam);
    /* Vector3.Transform(Vector3.Zero, cameraTransform) */
    for (int q = 0; q < 3; q++) cam_pos[q] = ((0.0f * cam[q] + 0.0f * cam[4 + q]) + 0.0f * cam[8 + q]) + cam[12 + q];
    orc_mat_perspective_fov(fov_degrees * 3.14159274f / 180.0f, (float)width / (float)height, nearp, farp, proj);
    orc_mat_mul(view, proj, vp);
    orc_mat_invert(vp, vp_inverse);
}

/* ---- per pixel ---------------------------------------------------------------------------- */
static void scene(const osc_node* nodes, int root, float x, float y, float z, float out[4])
{
    const float p[3] = {x, y, z};
    out[0] = out[1] = out[2] = out[3] = 0.0f;   /* W-only delegates leave XYZ = 0 (see header) */
    orc_eval(nodes, root, p, out);
}

static float max_ieee(float a, float b)   /* MathF.Max */
{
    if (a != b) { if (!(a != a)) return b < a ? a : b; return a; }
    return signbit(b) ? a : b;
}

static void normalize_inplace(float v[3])   /* Vec3Data.NormalizeInplace, VectorData.cs:490-508 */
{
    const float x = v[0], y = v[1], z = v[2];
    const float len = sqrtf((x * x + y * y) + z * z);
    if (len > 0) {
        const float r = 1.0f / len;
        v[0] = x * r; v[1] = y * r; v[2] = z * r;
    }
}

static void pixel(const osc_node* nodes, int root, int i, int j, int width, int height, const float cam[3],
                  const float M[16], float nearp, float farp, int iters, float* depth_out, float* rgb_out)
{
    /* GetCameraRays, RayMarcher.cs:113-128 */
    const float y = 1.0f - 2.0f * (float)j / (float)(height - 1);
    const float x = -1.0f + 2.0f * (float)i / (float)(width - 1);
    float v4[4];
    for (int q = 0; q < 4; q++) v4[q] = ((x * M[q] + y * M[4 + q]) + 0.0f * M[8 + q]) + 1.0f * M[12 + q];
    float d[3] = {v4[0] / v4[3] - cam[0], v4[1] / v4[3] - cam[1], v4[2] / v4[3] - cam[2]}, rd[3];
    v3normalize(d, rd);
    /* RenderDepth (RayMarcher.cs:80-95) and the depth loop of Render (:140-148): identical arithmetic */
    float depth = nearp - 0.1f, s[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        scene(nodes, root, rd[0] * depth + cam[0], rd[1] * depth + cam[1], rd[2] * depth + cam[2], s);
        depth = depth + s[3];
    }
    if (depth_out) *depth_out = depth;
    if (!rgb_out) return;
    float diffuse[3] = {0.0f, 0.0f, 0.0f};
    if (iters > 0) { diffuse[0] = 0.0f + s[0]; diffuse[1] = 0.0f + s[1]; diffuse[2] = 0.0f + s[2]; }
    /* Render, RayMarcher.cs:149-168 */
    const float sp[3] = {cam[0] + rd[0] * depth, cam[1] + rd[1] * depth, cam[2] + rd[2] * depth};
    const float go = 1e-5f;
    const float offs[6][3] = {{go * 1.0f, go * 0.0f, go * 0.0f}, {go * 0.0f, go * 1.0f, go * 0.0f}, {go * 0.0f, go * 0.0f, go * 1.0f},
                              {-go * 1.0f, -go * 0.0f, -go * 0.0f}, {-go * 0.0f, -go * 1.0f, -go * 0.0f}, {-go * 0.0f, -go * 0.0f, -go * 1.0f}};
    float w[6];
    for (int q = 0; q < 6; q++) {
        float t[4];
        scene(nodes, root, sp[0] + offs[q][0], sp[1] + offs[q][1], sp[2] + offs[q][2], t);
        w[q] = t[3];
    }
    float nrm[3] = {w[0] - w[3], w[1] - w[4], w[2] - w[5]};
    normalize_inplace(nrm);
    float ld[3] = {5.0f - sp[0], 5.0f - sp[1], 10.0f - sp[2]};
    normalize_inplace(ld);
    const float dv = max_ieee((nrm[0] * ld[0] + nrm[1] * ld[1]) + nrm[2] * ld[2], 0.0f);
    const float bgm = depth > farp ? 1.0f : 0.0f;
    const float bg[3] = {bgm * 0.5f, bgm * 0.75f, bgm * 1.0f};
    const float fgm = bgm == 0.0f ? 1.0f : 0.0f;
    for (int q = 0; q < 3; q++) {
        const float lighting = dv * diffuse[q] + 0.1f;
        rgb_out[q] = 0.0f + (lighting * fgm + bg[q]);
    }
}

typedef struct {
    const osc_node* nodes; int root, width, height, iters, j0, j1;
    const float* cam; const float* M; float nearp, farp; float* depth; float* rgb;
} ray_job;

static void* ray_worker(void* arg)
{
    ray_job* J = (ray_job*)arg;
    for (int j = J->j0; j < J->j1; j++)
        for (int i = 0; i < J->width; i++) {
            const size_t k = (size_t)j * J->width + i;
            pixel(J->nodes, J->root, i, j, J->width, J->height, J->cam, J->M, J->nearp, J->farp, J->iters,
                  J->depth ? J->depth + k : NULL, J->rgb ? J->rgb + 3 * k : NULL);
        }
    return NULL;
}

/* depth: width*height floats (row j, column i at j*width + i) or NULL; rgb: 3 per pixel or NULL */
void orc_raymarch(const osc_node* nodes, int root, int width, int height, const float cam_pos[3],
                  const float vp_inverse[16], float nearp, float farp, int iters, float* depth, float* rgb, int threads)
{
    if (threads <= 0) threads = orc_hardware_threads();
    if (threads > height) threads = height > 0 ? height : 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    ray_job* jobs = (ray_job*)malloc(sizeof(ray_job) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (ray_job){nodes, root, width, height, iters, (int)((long)height * t / threads), (int)((long)height * (t + 1) / threads),
                            cam_pos, vp_inverse, nearp, farp, depth, rgb};
        pthread_create(&th[t], NULL, ray_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}
