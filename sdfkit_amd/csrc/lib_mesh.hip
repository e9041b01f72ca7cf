// lib_mesh.hip -- Mesh (Mesh.cs:10-64): the accessors of a device-resident mesh.
#include "lib_internal.h"

// ---------------------------------------------------------------------------
// meshes
// ---------------------------------------------------------------------------
// How large will the mesh be?  Without waiting: for a mesh whose job is still queued, the sizes of the previous mesh of the
// same grid shape (what its buffers were sized from; exact whenever the scene repeats) -- the host can allocate and
// pre-fault its arrays while the GPU works and only re-allocates if sdfk_mesh_counts says otherwise; for a finished mesh,
// its counts.  *exact = 1 in the second case.
extern "C" int sdfk_mesh_size_hint(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices, int32_t* exact)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    int64_t nv = m->nv, ni = m->ni;
    int ex = 1;
    if (m->pending && !m->status) {
        ex = 0;
        auto it = g.hints.find(m->key);
        if (it != g.hints.end()) { nv = it->second.nv; ni = it->second.ni; }
        else { nv = (int64_t)m->cap_v; ni = (int64_t)m->cap_i; }
    }
    if (n_vertices) *n_vertices = nv;
    if (n_indices) *n_indices = ni;
    if (exact) *exact = ex;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_counts(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    if (n_vertices) *n_vertices = m->nv;
    if (n_indices) *n_indices = m->ni;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_stats(const sdfk_mesh* m, int64_t* n_active_cells, int64_t* n_case13_cells)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    if (n_active_cells) *n_active_cells = m->n_active;
    if (n_case13_cells) *n_case13_cells = m->n_case13;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_bounds(const sdfk_mesh* mc, float min[3], float max[3])
{
    StateScope in_owner_context(mc ? mc->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    sdfk_mesh* m = const_cast<sdfk_mesh*>(mc);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(m)) return r;
    if (!m->bounds_valid) {
        float hb[6];
        HIPCHK(hipMemcpyAsync(hb, m->bounds, 6 * sizeof(float), hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        if (m->nv > 0) { memcpy(m->h_min, hb, 12); memcpy(m->h_max, hb + 3, 12); }
        m->bounds_valid = true;
    }
    if (min) memcpy(min, m->h_min, 12);
    if (max) memcpy(max, m->h_max, 12);
    return SDFK_OK;
}

extern "C" int sdfk_mesh_copy(const sdfk_mesh* m, float* vertices3, float* colors3, float* normals3, int32_t* triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const size_t vb = (size_t)m->nv * 3 * sizeof(float);
    std::vector<CopyPiece> pieces;
    if (vertices3 && vb) pieces.push_back({m->vertices, vertices3, vb});
    if (colors3 && vb) {
        // (a volume without colours has all-zero mesh colours: nothing to move, the pool clears the array)
        if (m->has_colors) pieces.push_back({m->colors, colors3, vb});
    }
    if (normals3 && vb) pieces.push_back({m->normals, normals3, vb});
    if (triangles && m->ni) pieces.push_back({m->triangles, triangles, (size_t)m->ni * sizeof(int32_t)});
    if (m->lane != g.cur_lane) const_cast<sdfk_mesh*>(m)->used_on_main = true;
    std::function<void()> clear_colors;
    if (colors3 && vb && !m->has_colors)   // zero-fill (and first touch) on the pool, beside the transfers of the other arrays
        clear_colors = [=]() {
            std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
            const size_t per = size_t(2) << 20, nt = (vb + per - 1) / per;
            char* c = (char*)colors3;
            g_pool.start((int)nt, [=](int t) { const size_t a = (size_t)t * per; memset(c + a, 0, std::min(per, vb - a)); });
            g_pool.wait();
        };
    return copy_to_host(pieces, clear_colors);
}

extern "C" int sdfk_mesh_copy_device(const sdfk_mesh* m, void* vertices3, void* colors3, void* normals3, void* triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const_cast<sdfk_mesh*>(m)->used_on_main = true;
    const size_t vb = (size_t)m->nv * 3 * sizeof(float);
    if (vertices3 && vb) HIPCHK(hipMemcpyAsync(vertices3, m->vertices, vb, hipMemcpyDeviceToDevice, g.stream));
    if (colors3 && vb) {
        if (m->has_colors || m->colors_valid) HIPCHK(hipMemcpyAsync(colors3, m->colors, vb, hipMemcpyDeviceToDevice, g.stream));
        else HIPCHK(hipMemsetAsync(colors3, 0, vb, g.stream));   // (all zero, never stored: colors_valid)
    }
    if (normals3 && vb) HIPCHK(hipMemcpyAsync(normals3, m->normals, vb, hipMemcpyDeviceToDevice, g.stream));
    if (triangles && m->ni) HIPCHK(hipMemcpyAsync(triangles, m->triangles, (size_t)m->ni * sizeof(int32_t), hipMemcpyDeviceToDevice, g.stream));
    return SDFK_OK;
}

extern "C" int sdfk_mesh_device_ptrs(const sdfk_mesh* m, void** vertices3, void** colors3, void** normals3, void** triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;   // (the buffers may be replaced by an exact re-run)
    const_cast<sdfk_mesh*>(m)->used_on_main = true;
    if (colors3 && !m->has_colors && !m->colors_valid && m->colors && m->nv > 0) {   // the all-zero colours were never stored: now they are asked for
        if (int r = require_init()) return r;
        HIPCHK(hipMemsetAsync(m->colors, 0, (size_t)m->nv * 3 * sizeof(float), g.stream));
        const_cast<sdfk_mesh*>(m)->colors_valid = true;
    }
    if (vertices3) *vertices3 = m->vertices;
    if (colors3) *colors3 = m->colors;
    if (normals3) *normals3 = m->normals;
    if (triangles) *triangles = m->triangles;
    return SDFK_OK;
}

// Mesh.Transform(Matrix4x4) (Mesh.cs:47-64) on the device-resident mesh, in place.
extern "C" int sdfk_mesh_transform(sdfk_mesh* m, const float matrix[16], const float normal_matrix[16])
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m || !matrix || !normal_matrix) return fail(SDFK_ERR_INVALID, "sdfk_mesh_transform: null argument");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(m)) return r;
    // (a mesh whose arrays are a captured job's -- every repeat sdfk_sample_march of a launch-bound grid -- is transformed in
    // place like any other: the job stays busy, its buffers untouched by later launches, until this handle is freed)
    if (m->external) return fail(SDFK_ERR_UNSUPPORTED, "sdfk_mesh_transform: the mesh arrays are sections of a slab payload (extract the mesh with sdfk_dist_mesh first)");
    if (m->nv == 0) return SDFK_OK;   // (Mesh.Measure leaves Min / Max alone, Mesh.cs:32)
    if (m->lane != g.cur_lane) m->used_on_main = true;
    const int grid = grid_for((size_t)m->nv, 256, 1024);
    float* partial = nullptr;
    if (int r = dev_alloc((void**)&partial, (size_t)grid * 6 * sizeof(float))) return r;
    XformArgs A;
    A.vertices = m->vertices; A.normals = m->normals; A.n = m->nv; A.partial = partial;
    memcpy(A.m, matrix, sizeof A.m);
    memcpy(A.nm, normal_matrix, sizeof A.nm);
    hipLaunchKernelGGL(k_mesh_transform, dim3(grid), dim3(256), 0, g.stream, A);
    hipLaunchKernelGGL(k_bounds_reduce, dim3(1), dim3(256), 0, g.stream, (const float*)partial, grid, m->bounds);
    const hipError_t e = hipGetLastError();
    dev_free(partial);   // (stream-ordered)
    if (e != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_mesh_transform: %s", hipGetErrorString(e));
    m->bounds_valid = false;   // sdfk_mesh_bounds reads the new AABB from the device
    return SDFK_OK;
}

extern "C" void sdfk_mesh_free(sdfk_mesh* m)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return;
    if (m->pending) {   // never read: drop the queued job's workspace (stream-ordered, no wait)
        for (auto it = g.pending.begin(); it != g.pending.end(); ++it)
            if (*it == m) { g.pending.erase(it); break; }
        if (!m->graph_job) {
            job_release(m->pending, true);   // its kernels may still be queued (and will still write the result slot)
            delete m->pending;
        }
        m->pending = nullptr;
        if (m->done) (void)hipEventDestroy(m->done);
    }
    if (m->graph_job) { m->src = nullptr; graph_job_retire(m, false); }   // (the GraphJob is free for the next call on its lane: stream order protects its buffers)
    drop_source(m);
    if (m->lane != 0 && m->used_on_main && g.inited) {
        // lane-0 work may still read the buffers: their lane must not reuse them before that
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            (void)hipEventRecord(ev, g.lanes[0].stream);
            (void)hipStreamWaitEvent(lane_stream(m->lane), ev, 0);
            (void)hipEventDestroy(ev);
        }
    }
    free_mesh_buffers(m);
    delete m;
}

