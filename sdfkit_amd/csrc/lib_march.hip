// lib_march.hip -- MarchingCubes.CreateMesh (MarchingCubes.cs:39-92): the job driver, deferred completion, captured launch graphs, the fused
// SdfEx.ToMesh (sdfk_sample_march) and the slab forms the sharded step uses.
#include "lib_internal.h"

// ---------------------------------------------------------------------------
// marching cubes driver
// ---------------------------------------------------------------------------


template <typename T>
int job_alloc(sdfk_march_job* j, T** p, size_t count)
{
    void* q = nullptr;
    if (int r = dev_alloc(&q, count * sizeof(T))) return r;
    j->owned.push_back(q);
    *p = (T*)q;
    return SDFK_OK;
}

int acquire_slot()
{
    for (int pass = 0; pass < 2; pass++) {
        int oldest = -1;
        for (int i = 0; i < Context::NSLOTS; i++) {
            const int s = (g.slot_next + i) % Context::NSLOTS;
            Context::SlotState& st = g.slot_state[s];
            if (st.busy) continue;
            if (st.drop_pending) {   // the dropped job's kernels may still be queued: has its lane passed them?
                if (hipEventQuery(st.dropped) != hipSuccess) { if (oldest < 0) oldest = s; continue; }
                st.drop_pending = false;
            }
            st.busy = true;
            g.slot_next = (s + 1) % Context::NSLOTS;
            memset(&g.slots[s], 0, sizeof(Context::HostSlot));
            return s;
        }
        if (oldest < 0) break;   // every slot belongs to a live job
        // all free slots still wait for dropped jobs: wait for the one dropped first (slots are
        // handed out round-robin, so the first candidate after slot_next is the oldest) -- one
        // event, not a synchronisation of every lane
        (void)hipEventSynchronize(g.slot_state[oldest].dropped);
    }
    return -1;
}

void job_release(sdfk_march_job* j, bool kernels_may_be_queued)
{
    if (j->slot >= 0) {
        Context::SlotState& st = g.slot_state[j->slot];
        st.busy = false;
        st.drop_pending = false;
        if (kernels_may_be_queued) {
            hipError_t e = hipSuccess;
            if (!st.dropped) e = hipEventCreateWithFlags(&st.dropped, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(st.dropped, lane_stream(j->lane));
            if (e == hipSuccess) st.drop_pending = true;
            else (void)hipStreamSynchronize(lane_stream(j->lane));   // no event: wait here instead
        }
        j->slot = -1;
    }
    for (void* p : j->owned) dev_free(p);
    j->owned.clear();
    if (j->eval_prog) program_release(j->eval_prog);
    j->eval_prog = nullptr;
    if (j->sub) {
        dev_free(j->sub->values);
        dev_free(j->sub->colors);
        delete j->sub;
        j->sub = nullptr;
    }
}

#ifndef SDFK_COMPACT_MASKS
#define SDFK_COMPACT_MASKS 1   // the count pass leaves its activity masks for the write pass (McParams::segmask)
#endif
#ifndef SDFK_COMPACT_STRIDED
#define SDFK_COMPACT_STRIDED 1   // the write pass with interleaved segments (k_compact_write, mc_kernels.hip)
#endif

int alloc_records(sdfk_march_job* j, size_t c)
{
    McParams& P = j->P;
    for (size_t k = j->rec_first; k < j->owned.size(); k++) dev_free(j->owned[k]);
    j->owned.resize(j->rec_first);
    j->bounds_partial = nullptr;   // (allocated after the records: freed with them)
    j->vdesc = nullptr;
    j->vdesc_cap = 0;
    int rr = 0;
    rr = rr ? rr : job_alloc(j, &P.rec_xy, c);
    rr = rr ? rr : job_alloc(j, &P.rec_z, c);
    rr = rr ? rr : job_alloc(j, &P.rec_info, c);
    rr = rr ? rr : job_alloc(j, &P.rec_own, c);
    rr = rr ? rr : job_alloc(j, &P.rec_pre, c);
    rr = rr ? rr : job_alloc(j, &P.rec_corners, c * 8);
    rr = rr ? rr : job_alloc(j, &P.rec_vid, (c / MC_CHUNK + 1) * (size_t)MC_VSTRIDE);
    rr = rr ? rr : job_alloc(j, &P.chunkslots, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunktot, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkpre, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkdead, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkwin, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkwin2, c / MC_CHUNK + 2);
    P.cap_active = (uint32_t)c;
    P.chunkscan = c / MC_CHUNK + 1 > MC_SCAN_CHUNKS ? 1 : 0;   // (k_chunkscan: mc_kernels.hip)
    return rr;
}

// classification: sign bits (unless cached) -> ordered compaction -> corner gather ->
// resolve -> chunk scan.  Launches only; nothing here waits for the GPU.
int launch_classify(sdfk_march_job* j, bool publish)
{
    McParams& P = j->P;
    if (!j->have_bits) {
        uint64_t* bits = const_cast<uint64_t*>(P.bits);
        ProfScope ps("k_signbits");
        const int nx8 = (P.nx + 7) / 8, pitch8 = P.nzp;
        if ((P.nz % 256) != 0 || P.ny > 65535 || nx8 > 65535)   // chunks of the (y, z) plane: rows shorter or longer than a z tile
            hipLaunchKernelGGL(k_signbits8<true>, flat_grid((size_t)P.ny * pitch8, nx8), dim3(256), 0, g.stream,
                               P.values, j->bits8, P.nx, P.ny, P.nz, nx8, pitch8, P.iso);
        else
            hipLaunchKernelGGL(k_signbits8<false>, dim3((P.nz + 255) / 256, P.ny, nx8), dim3(256), 0, g.stream, P.values, j->bits8, P.nx, P.ny,
                               P.nz, nx8, pitch8, P.iso);
        hipLaunchKernelGGL(k_bits_transpose, transpose_grid(P.nz, P.ny, P.nxw), dim3(256), 0, g.stream, j->bits8, bits,
                           nx8, P.ny, P.nz, P.nxw, pitch8);
        HIPCHK(hipGetLastError());
        j->have_bits = true;
    }
    {
        ProfScope ps("k_compact");
        // (a workgroup takes the same 1024 segments of K2_LPB consecutive layers: mc_kernels.hip)
        const int nwg = ((P.lay_list_end - P.lay_count_begin + K2_LPB - 1) / K2_LPB) * P.bpl;
        hipLaunchKernelGGL(k_compact<false>, dim3(nwg), dim3(256), 0, g.stream, P);
        // (the count pass clears the culling kernel's counters: host bookkeeping of a device-side effect, so only once the launch is
        // known to be queued and to have workgroups -- a counter block recorded as clean that is not would make the next volume-less
        // job walk stale work lists)
        HIPCHK(hipGetLastError());
        if (P.zero_cull && j->cull_clean && nwg > 0) *j->cull_clean = true;
        if (P.blockpre) hipLaunchKernelGGL(k_blockscan, dim3(1), dim3(1024), 0, g.stream, P);   // (many blocks: their prefix in one pass)
        if (SDFK_COMPACT_STRIDED && K2_LPB == 1)   // (the write pass with interleaved segments: mc_kernels.hip)
            if (P.segmask) hipLaunchKernelGGL(k_compact_write<true>, dim3((P.lay_list_end - P.lay_count_begin) * P.bpl), dim3(256), 0, g.stream, P);
            else hipLaunchKernelGGL(k_compact_write<false>, dim3((P.lay_list_end - P.lay_count_begin) * P.bpl), dim3(256), 0, g.stream, P);
        else
            hipLaunchKernelGGL(k_compact<true>, dim3(nwg), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    const int nchunks = (int)((P.cap_active + MC_CHUNK - 1u) / MC_CHUNK);
    if (j->eval_prog) {   // the volume still is this program's output: evaluate the corners
        hipFunction_t fn_corners = nullptr;
        if (int r = program_fn(j->eval_prog, PK_CORNERS, &fn_corners)) return r;
        ProfScope ps("sdfk_corners_eval");
        const unsigned* n_active = &P.counters->n_active;
        void* params[] = {&j->eval_args, &P.rec_xy, &P.rec_z, &P.rec_corners, &n_active, &P.cap_active, &P.xbits, j->eval_prog->kargs()};
        HIPCHK(hipModuleLaunchKernel(fn_corners, (unsigned)std::min(nchunks, 256 * 8), 1, 1, 256, 1, 1, 0,
                                     g.stream, params, nullptr));
    } else {
        ProfScope ps("k_gather_corners");
        hipLaunchKernelGGL(k_gather_corners, dim3(std::min(nchunks, 256 * 8)), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps("k_resolve");
        constexpr int cap = 256 * 12;
        hipLaunchKernelGGL(k_resolve, dim3(std::min(nchunks, cap)), dim3(256), 0, g.stream, P);
        if (P.chunkscan) hipLaunchKernelGGL(k_chunkscan, dim3(1), dim3(1024), 0, g.stream, P);   // (long lists: the chunks' prefix in one pass)
        // totals for the host: workgroup 0 of k_vertices publishes them, unless the caller
        // needs the counts before (or without) emitting
        if (publish) hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    return SDFK_OK;
}

// the one host<->device rendezvous of a march: drain the stream, read the mirrored counters
int wait_counters(sdfk_march_job* j)
{
    HIPCHK(hipStreamSynchronize(g.stream));
    j->c = g.slots[j->slot].c;
    return SDFK_OK;
}

// Builds the job for `v` (or for a subsampled copy when step > 1); no kernel except the
// optional subsample is launched here.
int setup_job(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, size_t cap_records,
              sdfk_march_job** out)
{
    *out = nullptr;
    if (step < 1) return fail(SDFK_ERR_INVALID, "step must be >= 1");
    const bool slab = (v->z0 != 0 || v->nz != v->nz_global);
    if (step != 1 && (slab || layer_begin != 0)) return fail(SDFK_ERR_UNSUPPORTED, "slab meshing supports step == 1 only");
    sdfk_march_job* j = new sdfk_march_job();
    j->gnx = v->nx; j->gny = v->ny; j->gnz = v->nz_global;
    memcpy(j->gmin, v->gmin, sizeof j->gmin);
    memcpy(j->gmax, v->gmax, sizeof j->gmax);
    j->lane = g.cur_lane;
    j->slot = acquire_slot();
    if (j->slot < 0) { delete j; return fail(SDFK_ERR_NOMEM, "more than %d marching-cubes jobs are alive", Context::NSLOTS); }
    const sdfk_volume* w = v;
    if (step > 1) {
        // MarchingCubes.cs:49-80 touches only voxels at multiples of step
        sdfk_volume* s = new sdfk_volume(*v);
        s->nx = (v->nx - 1) / step + 1; s->ny = (v->ny - 1) / step + 1; s->nz = (v->nz - 1) / step + 1;
        s->nz_global = s->nz;
        s->values = nullptr; s->colors = nullptr; s->bits = nullptr; s->bits8 = nullptr; s->bits_valid = false; s->sampled_by = nullptr;
        j->sub = s;
        int r = dev_alloc((void**)&s->values, s->nalloc() * sizeof(float));
        if (!r && v->colors) r = dev_alloc((void**)&s->colors, s->nalloc() * 3 * sizeof(float));
        if (r) { job_release(j); delete j; return r; }
        ProfScope ps("k_subsample");
        hipLaunchKernelGGL(k_subsample, dim3(grid_for(s->nvox())), dim3(256), 0, g.stream, v->values, v->colors,
                           s->values, s->colors, v->nx, v->ny, v->pitch(), s->nx, s->ny, s->nz, s->pitch(), step);
        w = s;
        layer_end = s->nz - 1;
    }
    McParams& P = j->P;
    memset(&P, 0, sizeof P);
    P.values = w->values; P.colors = w->colors;
    j->colors_elided = w->elided && w->elided_colors;
    // (the counters of the culling kernel that made this volume's sign bits: the count pass clears them for the lane's next job)
    if (w->elided && w->cull_header && w->cull_header_lane == g.cur_lane) { P.zero_cull = w->cull_header; j->cull_clean = &w->cull_header_clean; }
    if (w->elided && (step != 1 || !w->sampled_by || !g_cfg.corner_eval || (w->elided_colors && !g_cfg.vcolor_eval) || !w->bits_valid || w->bits_iso != iso)) {
        job_release(j); delete j;
        return fail(SDFK_ERR_INVALID, "internal: a volume without storage can only be meshed by the program that sampled it (step 1, same iso)");
    }
    P.nx = w->nx; P.ny = w->ny; P.nz = w->nz;
    P.nzp = w->pitch();
    {   // bit split of the packed cell coordinates: 16 + 16 unless one of nx, ny needs more (the other then needs fewer)
        auto bits = [](int n) { int b = 0; while ((1 << b) < n) b++; return b; };
        const int bx = bits(w->nx), by = bits(w->ny);
        P.xbits = by > 16 ? 32 - by : std::max(16, bx);
        P.xmask = P.xbits >= 32 ? 0xffffffffu : ((1u << P.xbits) - 1u);
    }
    P.ncx = w->nx - 1; P.ncy = w->ny - 1; P.ncz = w->nz - 1;
    P.nxw = (w->nx + 63) / 64;
    P.z0 = w->z0;
    P.iso = iso;
    P.step = step;
    const int ncz_global = w->nz_global - 1;
    if (layer_begin < 0 || layer_end > std::max(ncz_global, 0) || layer_begin > layer_end) {
        job_release(j); delete j;
        return fail(SDFK_ERR_INVALID, "layer range [%d,%d) outside [0,%d)", layer_begin, layer_end, ncz_global);
    }
    P.lay_emit_begin = layer_begin - w->z0;
    P.lay_emit_end = layer_end - w->z0;
    P.lay_count_begin = layer_begin > 0 ? P.lay_emit_begin - 1 : P.lay_emit_begin;
    P.lay_list_end = std::min(P.lay_emit_end + 1, P.ncz);   // the layer above feeds seam normals
    j->empty = (P.ncx <= 0 || P.ncy <= 0 || P.ncz <= 0 || layer_begin == layer_end);
    memset(&j->c, 0, sizeof j->c);
    if (j->empty) { *out = j; return SDFK_OK; }
    {   // context planes the slab must hold (see sdfkit_hip.h)
        const int need_lo = std::max(layer_begin - 2, 0), need_hi = std::min(layer_end + 2, w->nz_global);
        if (w->z0 > need_lo || w->z0 + w->nz < need_hi) {
            job_release(j); delete j;
            return fail(SDFK_ERR_INVALID, "slab planes [%d,%d) do not cover the context [%d,%d) of layers [%d,%d)",
                        w->z0, w->z0 + w->nz, need_lo, need_hi, layer_begin, layer_end);
        }
    }
    // logical blocks of k_compact: 1024 consecutive 64-cell segments of one layer
    P.bpl = (int)(((size_t)P.ncy * P.nxw + 1023) / 1024);
    const size_t ncell = (size_t)P.ncx * P.ncy * P.ncz;
    if (cap_records == 0) cap_records = std::max<size_t>(ncell / 12, 1u << 16);
    cap_records = std::min(cap_records, ncell);
    int r = 0;
    if (step == 1 && v->sampled_by && g_cfg.corner_eval) {
        j->eval_prog = v->sampled_by;
        j->eval_prog->refs++;
        j->eval_args = v->sampled_args;
    }
    j->have_bits = (step == 1 && v->bits && v->bits_valid && v->bits_iso == iso);
    if (j->have_bits) P.bits = v->bits;   // written by the fused sampling kernel; owned by the volume
    else {
        uint64_t* bits = nullptr;
        r = r ? r : job_alloc(j, &bits, (size_t)P.nz * P.ny * P.nxw + 8);   // k_compact reads 4 words past a row pair
        P.bits = bits;
        r = r ? r : job_alloc(j, &j->bits8, (size_t)P.ny * ((P.nx + 7) / 8) * ((P.nz + 3) & ~3) + 64);
    }
    r = r ? r : job_alloc(j, &P.blockcnt, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl + 1);
    r = r ? r : job_alloc(j, &P.wavecnt, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl * 4 + 4);
    P.segmask = nullptr;
    if (SDFK_COMPACT_STRIDED && K2_LPB == 1 && SDFK_COMPACT_MASKS)   // (the write pass reads the count pass's masks: mc_kernels.hip)
        r = r ? r : job_alloc(j, &P.segmask, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl * 1024 + 4);
    P.blockpre = nullptr;
    if ((P.lay_list_end - P.lay_count_begin) * P.bpl > MC_SCAN_BLOCKS && SDFK_COMPACT_STRIDED && K2_LPB == 1)   // (k_blockscan: mc_kernels.hip)
        r = r ? r : job_alloc(j, &P.blockpre, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl + 1);
    r = r ? r : job_alloc(j, &P.rowstart, (size_t)(P.lay_list_end - P.lay_count_begin) * P.ncy + 2);
    r = r ? r : job_alloc(j, &P.counters, 1);
    P.host_counters = &g.slots_dev[j->slot].c;
    j->rec_first = j->owned.size();
    r = r ? r : alloc_records(j, cap_records);
    if (r) { job_release(j); delete j; return r; }
    *out = j;
    return SDFK_OK;
}

int alloc_mesh(sdfk_mesh** out, size_t cap_v, size_t cap_i)
{
    sdfk_mesh* m = new sdfk_mesh();
    int r = 0;
    r = r ? r : dev_alloc((void**)&m->vertices, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->colors, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->normals, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->triangles, std::max<size_t>(cap_i, 1) * sizeof(int32_t));
    r = r ? r : dev_alloc((void**)&m->bounds, 8 * sizeof(float));
    if (r) { sdfk_mesh_free(m); return r; }
    m->lane = g.cur_lane;
    m->cap_v = cap_v; m->cap_i = cap_i;
    *out = m;
    return SDFK_OK;
}

// emit: vertices (+ AABB partials) then triangles (+ AABB reduction).  Launches only.
int launch_emit(sdfk_march_job* j, sdfk_mesh* m, int64_t vertex_base)
{
    McMeshOut M;
    memset(&M, 0, sizeof M);
    M.vertices = m->vertices; M.colors = m->colors; M.normals = m->normals; M.triangles = m->triangles;
    m->has_colors = j->P.colors != nullptr || j->colors_elided;
    if (!m->has_colors) {   // a .W-only program: every colour is (0,0,0) (Voxels.cs:88-92) -- nothing is stored, sdfk_mesh_copy clears the host array
        M.colors = nullptr;
        m->colors_valid = false;
    }
    M.cap_vertices = (uint32_t)m->cap_v;
    M.cap_indices = m->cap_i;
    M.vertex_base = vertex_base;
    M.slab_header = m->slab_header;
    M.slab_vbytes = m->has_colors ? 36 : 24;
    // MarchingCubes.cs:85-90 (row-vector T*S*T) and Mesh.cs:49-55, all float32
    const int nn[3] = {j->gnx, j->gny, j->gnz};
    for (int k = 0; k < 3; k++) {
        const float size = j->gmax[k] - j->gmin[k];
        const float sum = j->gmin[k] + j->gmax[k];
        const float center = sum * 0.5f;
        const float t1 = (float)(-(nn[k] - 1)) / 2.0f;
        M.sc[k] = size / (float)(nn[k] - 1);
        const float ts = t1 * M.sc[k];
        M.tr[k] = ts + center;
    }
    {
        const float yz = M.sc[1] * M.sc[2], xz = M.sc[0] * M.sc[2], xy = M.sc[0] * M.sc[1];
        const float det = M.sc[0] * yz;
        const float inv_det = 1.0f / det;
        M.inv[0] = yz * inv_det; M.inv[1] = xz * inv_det; M.inv[2] = xy * inv_det;
    }
#ifndef SDFK_KV_GRIDCAP
#define SDFK_KV_GRIDCAP (256 * 8)
#endif
    constexpr int vcap = SDFK_KV_GRIDCAP, tcap = 256 * 8;   // (persistent-workgroup caps were measured: +-2 us, noise)
    const int vgrid = grid_for(j->P.cap_active, (int)MC_CHUNK, vcap);
    if (!j->bounds_partial || j->bounds_blocks != vgrid) {
        if (int rr = job_alloc(j, &j->bounds_partial, (size_t)vgrid * 6)) return rr;
        j->bounds_blocks = vgrid;
    }
    M.bounds_partial = j->bounds_partial;
    M.bounds_blocks = vgrid;
    // Vertex colours of a volume its own program has just sampled: re-evaluated by the program (sdfk_vertex_colors) from
    // the (creator record, edge) descriptors k_vertices leaves, instead of gathered from the colour volume
    const bool no_vcol = !g_cfg.vcolor_eval;   // (SDFK_OPT_VCOLOR_EVAL = 0: the gather path)
    const bool vcol = j->eval_prog && j->eval_prog->writes_color && (j->P.colors || j->colors_elided) && M.colors && j->P.step == 1 && !no_vcol;
    if (vcol) {
        const size_t need = std::max<size_t>(m->cap_v, 1);
        if (!j->vdesc || j->vdesc_cap < need) {
            if (int rr = job_alloc(j, &j->vdesc, need)) return rr;
            j->vdesc_cap = need;
        }
        M.vdesc = j->vdesc;
    }
    M.bounds = m->bounds;
    M.host_bounds = g.slots_dev[j->slot].bounds;
    phase_token_wait(1);
    {
        ProfScope ps("k_vertices");
        uint32_t iso_bits;
        memcpy(&iso_bits, &j->P.iso, 4);
        if (iso_bits == 0u) hipLaunchKernelGGL(k_vertices<true>, dim3(vgrid), dim3(256), 0, g.stream, j->P, M);   // (+0.0: the usual iso value)
        else hipLaunchKernelGGL(k_vertices<false>, dim3(vgrid), dim3(256), 0, g.stream, j->P, M);
        HIPCHK(hipGetLastError());
    }
    phase_token_pass(1);
    if (vcol) {
        struct VColArgs { const uint2* vdesc; const uint32_t* rec_xy; const uint32_t* rec_z; const McCounters* counters; float* colors;
                          uint32_t cap_vertices; int32_t xbits; float iso; } V;   // (= VColArgs of sample_codegen.h)
        V.vdesc = j->vdesc; V.rec_xy = j->P.rec_xy; V.rec_z = j->P.rec_z; V.counters = j->P.counters; V.colors = M.colors;
        V.cap_vertices = M.cap_vertices; V.xbits = j->P.xbits; V.iso = j->P.iso;
        hipFunction_t fn = nullptr;
        if (int rr = program_fn(j->eval_prog, PK_VCOLORS, &fn)) return rr;
        void* params[] = {&j->eval_args, &V, j->eval_prog->kargs()};
        ProfScope ps("sdfk_vertex_colors");
        HIPCHK(hipModuleLaunchKernel(fn, (unsigned)grid_for(std::max<size_t>(m->cap_v, 1), 256, 256 * 8), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    }
    {
        ProfScope ps("k_triangles");
        hipLaunchKernelGGL(k_triangles, dim3(grid_for(j->P.cap_active, (int)MC_CHUNK, tcap)), dim3(256), 0, g.stream, j->P, M);
        HIPCHK(hipGetLastError());
    }
    return SDFK_OK;
}

void finalize_mesh(sdfk_march_job* j, sdfk_mesh* m, bool have_bounds)
{
    const uint32_t nghost = j->c.nghost;
    m->nv = (int64_t)j->c.total_v - (int64_t)nghost;
    m->ni = (int64_t)j->c.total_t * 3;
    m->n_active = j->c.n_emit_cells;
    m->n_case13 = j->c.n_dead;
    if (m->nv == 0) { m->bounds_valid = true; return; }   // Mesh.Measure leaves Min/Max at zero (Mesh.cs:32)
    if (have_bounds) {
        memcpy(m->h_min, g.slots[j->slot].bounds, 12);
        memcpy(m->h_max, g.slots[j->slot].bounds + 3, 12);
        m->bounds_valid = true;
    }
}

uint64_t hint_key(const sdfk_volume* v, int step, int layer_begin, int layer_end)
{
    uint64_t k = ((uint64_t)v->nx << 44) ^ ((uint64_t)v->ny << 24) ^ ((uint64_t)v->nz << 4) ^ (uint64_t)(step & 15);
    return k * 0x9E3779B97F4A7C15ull ^ ((uint64_t)(uint32_t)layer_begin << 32 | (uint32_t)layer_end) ^ ((uint64_t)v->z0 << 17);
}

// Exact path: classify, wait for the counts, size the outputs exactly, emit.
int march_exact(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base,
                uint64_t key, sdfk_mesh** out)
{
    *out = nullptr;
    sdfk_march_job* j = nullptr;
    int r = setup_job(v, iso, step, layer_begin, layer_end, 0, &j);
    if (r) return r;
    sdfk_mesh* m = nullptr;
    if (j->empty) {
        r = alloc_mesh(&m, 0, 0);
        if (!r) m->bounds_valid = true;
    } else {
        r = launch_classify(j, true);
        r = r ? r : wait_counters(j);
        if (!r && v->elided && j->c.n_case13 != 0) {
            // case-13 sign words: k_resolve's dead-cell test reads neighbouring VOXELS, and this volume has none
            // (SDFK_OPT_ELIDE_VOLUME): give it its storage, sample again with stores, start over
            job_release(j);
            delete j;
            if (int r2 = volume_materialize(const_cast<sdfk_volume*>(v))) return r2;
            return march_exact(v, iso, step, layer_begin, layer_end, vertex_base, key, out);
        }
        if (!r && j->c.n_active > j->P.cap_active) {   // record list too small: exact size, redo
            r = alloc_records(j, j->c.n_active);
            r = r ? r : launch_classify(j, true);
            r = r ? r : wait_counters(j);
        }
        r = r ? r : alloc_mesh(&m, (size_t)(j->c.total_v - j->c.nghost), (size_t)j->c.total_t * 3);
        if (!r && vertex_base + (int64_t)(j->c.total_v - j->c.nghost) >= (int64_t(1) << 31))
            r = fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
        if (!r && j->c.n_active > 0) {
            r = launch_emit(j, m, vertex_base);
            r = r ? r : wait_counters(j);
            if (!r && j->c.overflow) r = fail(SDFK_ERR_HIP, "marching cubes: output capacity exceeded unexpectedly");
        }
        if (!r) {
            finalize_mesh(j, m, j->c.n_active > 0);
            g.hints[key] = Context::Hint{j->c.n_active, (uint32_t)m->nv, (uint32_t)m->ni};
        }
    }
    job_release(j);
    delete j;
    if (r) { if (m) sdfk_mesh_free(m); return r; }
    *out = m;
    return SDFK_OK;
}

void free_mesh_buffers(sdfk_mesh* m)
{
    if (m->borrowed) {   // the buffers belong to a GraphJob
        m->borrowed = false;
        m->vertices = m->colors = m->normals = m->bounds = nullptr;
        m->triangles = nullptr;
        return;
    }
    if (!m->external) {
        dev_free(m->vertices);   // stream-ordered pool: no sync needed
        dev_free(m->colors);
        dev_free(m->normals);
        dev_free(m->triangles);
    }
    m->external = false;
    dev_free(m->bounds);
    m->vertices = m->colors = m->normals = m->bounds = nullptr;
    m->triangles = nullptr;
}

void drop_source(sdfk_mesh* m)
{
    if (m->owns_src && m->src) sdfk_volume_free(const_cast<sdfk_volume*>(m->src));
    m->src = nullptr;
    m->owns_src = false;
}

// Completes a mesh of the speculative path (see sdfk_mesh): no-op for a finished mesh.
int mesh_resolve(sdfk_mesh* m)
{
    if (m->status) { t_err = m->error; return m->status; }
    if (!m->pending) return SDFK_OK;
    LaneScope on_lane(m->lane);   // an exact re-run queues (and allocates) where the first attempt did
    sdfk_march_job* j = m->pending;
    m->pending = nullptr;
    for (auto it = g.pending.begin(); it != g.pending.end(); ++it)
        if (*it == m) { g.pending.erase(it); break; }
    int r = SDFK_OK;
    const hipError_t e = hipEventSynchronize(m->done);
    (void)hipEventDestroy(m->done);
    m->done = nullptr;
    if (e != hipSuccess) r = fail(SDFK_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
    j->c = g.slots[j->slot].c;
    // (an elided volume -- SDFK_OPT_ELIDE_VOLUME -- whose sign words contain case 13 is redone on the exact path, which gives it
    // its storage first: the dead-cell test of k_resolve reads neighbouring voxels)
    const bool needs_voxels = m->src && m->src->elided && j->c.n_case13 != 0;
    const bool fits = !needs_voxels && j->c.n_active <= j->P.cap_active && j->c.overflow == 0 &&
                      (size_t)(j->c.total_v - j->c.nghost) <= m->cap_v && (size_t)j->c.total_t * 3 <= m->cap_i;
    if (!r && fits) {
        if (m->vertex_base + (int64_t)(j->c.total_v - j->c.nghost) >= (int64_t(1) << 31))
            r = fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
        else {
            finalize_mesh(j, m, true);
            g.hints[m->key] = Context::Hint{j->c.n_active, (uint32_t)m->nv, (uint32_t)m->ni};
        }
    }
    if (!m->graph_job) {   // (a job of a captured launch graph belongs to its GraphJob and is replayed)
        job_release(j);
        delete j;
    }
    if (!r && !fits) {   // the guess was too small: the exact two-phase path, into the same handle
        sdfk_mesh* x = nullptr;
        r = march_exact(m->src, m->iso, m->step, m->layer_begin, m->layer_end, m->vertex_base, m->key, &x);
        if (m->graph_job) graph_job_retire(m, true);   // its capacities are too small for this scene: rebuilt on a later call
        if (!r) {
            free_mesh_buffers(m);
            m->nv = x->nv; m->ni = x->ni;
            m->vertices = x->vertices; m->colors = x->colors; m->normals = x->normals; m->triangles = x->triangles;
            m->bounds = x->bounds;
            memcpy(m->h_min, x->h_min, sizeof m->h_min);
            memcpy(m->h_max, x->h_max, sizeof m->h_max);
            m->bounds_valid = x->bounds_valid;
            m->n_active = x->n_active; m->n_case13 = x->n_case13;
            m->cap_v = x->cap_v; m->cap_i = x->cap_i;
            m->has_colors = x->has_colors;
            m->colors_valid = x->colors_valid;
            delete x;
        }
    }
    drop_source(m);
    if (r) { m->nv = m->ni = 0; m->status = r; m->error = t_err; }
    return r;
}

// every pending mesh that still depends on the contents of `v` (called before `v` changes or dies)
void resolve_dependents(const sdfk_volume* v)
{
    for (;;) {
        sdfk_mesh* hit = nullptr;
        for (sdfk_mesh* m : g.pending)
            if (m->src == v) { hit = m; break; }
        if (!hit) return;
        (void)mesh_resolve(hit);   // an error stays in the mesh (sticky)
    }
}

// A mesh whose arrays are sections of a slab payload at `dst` (64-byte header, then V | (C) | N | T laid out for the
// capacities): what a sharded step emits into when it writes straight into its all-gather send buffer.  The
// capacities are the size hints scaled up to what `capacity` bytes hold (at most the usual +25 %); returns false when
// not even the hints fit (the caller then takes the ordinary path and packs).
bool external_mesh(char* dst, int64_t capacity, bool colors, uint32_t nv_hint, uint32_t ni_hint, sdfk_mesh** out)
{
    const int64_t vb = colors ? 36 : 24, avail = capacity - SDFK_SLAB_HEADER_BYTES;
    const int64_t min_v = (int64_t)nv_hint + 64, min_i = (int64_t)ni_hint + 192;
    if (avail < vb * min_v + 4 * min_i) return false;
    const double scale = std::min(1.25, (double)avail / (double)(vb * min_v + 4 * min_i));
    int64_t cap_v = std::max<int64_t>(min_v, (int64_t)((double)min_v * scale));
    int64_t cap_i = (avail - vb * cap_v) / 4;
    cap_i = std::min<int64_t>(cap_i, (int64_t)ni_hint + ni_hint / 4 + 12288);
    cap_i -= cap_i % 3;
    if (cap_i < min_i - 2) return false;
    sdfk_mesh* m = new sdfk_mesh();
    if (dev_alloc((void**)&m->bounds, 8 * sizeof(float))) { delete m; return false; }
    char* q = dst + SDFK_SLAB_HEADER_BYTES;
    m->vertices = (float*)q; q += 12 * cap_v;
    if (colors) { m->colors = (float*)q; q += 12 * cap_v; }
    else m->colors = nullptr;       // no colour section: k_vertices skips the (all-zero) colour stores
    m->normals = (float*)q; q += 12 * cap_v;
    m->triangles = (int32_t*)q;
    m->external = true;
    m->slab_header = dst;
    m->has_colors = colors;
    m->lane = g.cur_lane;
    m->cap_v = (size_t)cap_v; m->cap_i = (size_t)cap_i;
    *out = m;
    return true;
}

// MarchingCubes.CreateMesh on the cell layers [layer_begin, layer_end) of a volume / slab.  emit_dst != nullptr (sharded
// step): on the speculative path the mesh is emitted straight into that slab payload (external_mesh).
int march_range(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base, sdfk_mesh** out,
                char* emit_dst, int64_t emit_capacity)
{
    *out = nullptr;
    const uint64_t key = hint_key(v, step, layer_begin, layer_end);
    auto it = g.hints.find(key);
    if (it != g.hints.end()) {
        // Speculative path: the sizes of the previous mesh of this shape (+25 % and a floor)
        // size every buffer; classification AND emit are queued back to back and the handle
        // is returned without waiting: the host meets the GPU only when a result is read.
        const Context::Hint h = it->second;
        sdfk_march_job* j = nullptr;
        int r = setup_job(v, iso, step, layer_begin, layer_end, (size_t)h.n_active + h.n_active / 4 + 4096, &j);
        if (r) return r;
        if (!j->empty) {
            sdfk_mesh* m = nullptr;
            if (!(emit_dst && external_mesh(emit_dst, emit_capacity, v->colors != nullptr, h.nv, h.ni, &m)))
                r = alloc_mesh(&m, (size_t)h.nv + h.nv / 4 + 4096, (size_t)h.ni + h.ni / 4 + 12288);
            r = r ? r : launch_classify(j, false);
            r = r ? r : launch_emit(j, m, vertex_base);
            if (!r) {
                hipError_t e = hipEventCreateWithFlags(&m->done, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(m->done, g.stream);
                if (e != hipSuccess) r = fail(SDFK_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
            }
            if (r) {
                (void)hipStreamSynchronize(g.stream);
                if (m) { if (m->done) (void)hipEventDestroy(m->done); m->done = nullptr; sdfk_mesh_free(m); }
                job_release(j);
                delete j;
                return r;
            }
            m->pending = j;
            m->src = v;
            m->iso = iso; m->step = step; m->layer_begin = layer_begin; m->layer_end = layer_end;
            m->vertex_base = vertex_base; m->key = key;
            g.pending.push_back(m);
            while (g.pending.size() > Context::MAX_PENDING) (void)mesh_resolve(g.pending.front());
            *out = m;
            return SDFK_OK;
        }
        job_release(j);
        delete j;
    }
    return march_exact(v, iso, step, layer_begin, layer_end, vertex_base, key, out);
}



// ---------------------------------------------------------------------------
// captured launch graphs for repeat sdfk_sample_march jobs
// ---------------------------------------------------------------------------
// On launch-bound grids (<= 2^24 voxels) a job is nine small dependent kernels: queueing them costs the host 23 us, more
// than the GPU needs for the job next to the others in flight (tools/ubench/ub_graph.hip: 9 launches 23.4 us, one
// hipGraphLaunch of the captured chain 6.0 us; the chain takes the GPU the same time either way).  So the speculative
// job of a (program, grid, clip, iso) on a lane is built ONCE -- its own volume, workspace sized from the shape's hints,
// mesh buffers, result slot -- its launches are captured into a hipGraph, and every later call for that key on that lane
// is one hipGraphLaunch.  The mesh handle borrows the GraphJob's buffers until it is freed; a result that does not fit the
// captured capacities is redone on the exact path as always, and the GraphJob is rebuilt with the new hints.


struct GraphJob {
    const sdfk_program* prog = nullptr;
    int nx = 0, ny = 0, nz = 0, clip = 0, lane = 0;
    float mn[3], mx[3], iso = 0.0f;
    sdfk_volume* vol = nullptr;
    sdfk_march_job* job = nullptr;
    sdfk_mesh* proto = nullptr;     // owns the mesh buffers and their capacities
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t key = 0;
    sdfk_mesh* borrower = nullptr;  // the live handle whose arrays are proto's (at most one)
    // slab form (sdfk_slab_enqueue): the caller's slab volume and send buffer, nothing borrowed, never "busy"
    bool slab = false;
    sdfk_volume* ext_vol = nullptr;
    void* dst = nullptr;
    void* post_out = nullptr;       // compact payloads: where k_payload_compact (the graph's last node) writes
    int64_t capacity = 0;
    int lb = 0, le = 0;
    Context::Hint hint{};           // the size hints the capacities were derived from (other hints now: rebuild)
    SampleArgs args;                // what sample_impl recorded in the volume (restored on every replay)
    hipEvent_t ran = nullptr;       // after the latest run (a handle dropped unread leaves its run in flight)
    bool busy = false, stale = false;
    size_t bytes = 0;
    uint64_t last_use = 0;
};

void graph_job_destroy(GraphJob* q)
{
    (void)hipStreamSynchronize(lane_stream(q->lane));   // its kernels may still be queued
    if (q->exec) (void)hipGraphExecDestroy(q->exec);
    if (q->graph) (void)hipGraphDestroy(q->graph);
    if (q->ran) (void)hipEventDestroy(q->ran);
    if (sdfk_mesh* b = q->borrower) {   // (only at shutdown: a handle that outlives the library reads as failed)
        b->graph_job = nullptr; b->borrowed = false;
        b->vertices = b->colors = b->normals = b->bounds = nullptr; b->triangles = nullptr;
        b->nv = b->ni = 0; b->pending = nullptr; b->src = nullptr;
        if (!b->status) { b->status = SDFK_ERR_INVALID; b->error = "the library was shut down"; }
    }
    LaneScope on_lane(q->lane);
    if (q->job) { job_release(q->job); delete q->job; }
    if (q->proto) { q->proto->graph_job = nullptr; q->proto->borrowed = false; sdfk_mesh_free(q->proto); }
    if (q->vol) sdfk_volume_free(q->vol);
    if (q->prog) program_release(const_cast<sdfk_program*>(q->prog));
    g.graph_bytes -= std::min(g.graph_bytes, q->bytes);
    for (auto it = g.graph_jobs.begin(); it != g.graph_jobs.end(); ++it)
        if (*it == q) { g.graph_jobs.erase(it); break; }
    delete q;
}

void graph_jobs_destroy_all()
{
    while (!g.graph_jobs.empty()) graph_job_destroy(g.graph_jobs.back());
}

// the mesh no longer needs its GraphJob (freed, or redone on the exact path because the captured capacities were too small)
void graph_job_retire(sdfk_mesh* m, bool too_small)
{
    GraphJob* q = m->graph_job;
    if (!q) return;
    if (too_small) q->stale = true;
    q->busy = false;
    q->borrower = nullptr;
    m->graph_job = nullptr;
    if (q->prog && q->prog->orphaned) {   // its program handle was destroyed while this mesh was out
        if (m->src == q->vol) m->src = nullptr;
        if (m->pending == q->job) m->pending = nullptr;
        graph_job_destroy(q);
    }
}

void graph_jobs_forget_program(const sdfk_program* p)
{
    for (size_t i = 0; i < g.graph_jobs.size();) {
        if (g.graph_jobs[i]->prog == p && !g.graph_jobs[i]->busy) graph_job_destroy(g.graph_jobs[i]);   // (erases the entry)
        else i++;
    }
}

void graph_jobs_forget_volume(const sdfk_volume* v)
{
    for (size_t i = 0; i < g.graph_jobs.size();) {
        if (g.graph_jobs[i]->slab && g.graph_jobs[i]->ext_vol == v) graph_job_destroy(g.graph_jobs[i]);   // (erases the entry)
        else i++;
    }
}

bool graphs_enabled(int64_t nvox)
{
    const int mode = g_cfg.graphs;   // SDFK_OPT_GRAPHS -- 0: never, 1: launch-bound grids, 2: every grid
    return mode == 2 || (mode == 1 && nvox <= (int64_t(1) << 24));   // (measured: 64^3 35 -> 23.5 us per job, 128^3 33 -> 26, 256^3 42 -> 39; 320^3 and up 0-3 % slower)
}

bool graphs_enabled_slab(int64_t nvox)   // (a sharded step also pays for a collective call on the host: graphs pay up to larger slabs)
{
    const int mode = g_cfg.graphs;
    return mode == 2 || (mode == 1 && nvox <= (int64_t(1) << 25));
}

// Queue the job of (p, grid, clip, iso) on the current lane from a captured graph.  *out stays null when graphs do not
// apply (no size hints yet, no free GraphJob and the limits are reached, capture not possible): the caller then takes
// the ordinary path.
int graph_sample_march(const sdfk_program* p, const float mn[3], const float mx[3], int nx, int ny, int nz, int clip, float iso, sdfk_mesh** out)
{
    *out = nullptr;
    const int lane = g.cur_lane;
    if (lane == 0 || g.prof_on || g.sampler_only || !mn || !mx || g_graph_build_failures >= 8) return SDFK_OK;
    GraphJob* q = nullptr;
    size_t alive = 0;
    for (size_t i = 0; i < g.graph_jobs.size();) {
        GraphJob* c = g.graph_jobs[i];
        const bool same = !c->slab && c->prog == p && c->nx == nx && c->ny == ny && c->nz == nz && c->clip == clip && c->lane == lane &&
                          memcmp(&c->iso, &iso, 4) == 0 && memcmp(c->mn, mn, 12) == 0 && memcmp(c->mx, mx, 12) == 0;
        if (same && !c->busy && c->stale) { graph_job_destroy(c); continue; }   // (erases g.graph_jobs[i])
        if (same) alive++;
        if (same && !c->busy && !q) q = c;
        i++;
    }
    if (!q) {
        // build: everything a job needs, allocated up front, sized from the hints of this grid shape
        sdfk_volume probe;
        probe.nx = nx; probe.ny = ny; probe.nz = nz; probe.nz_global = nz; probe.z0 = 0;
        const uint64_t key = hint_key(&probe, 1, 0, std::max(nz - 1, 0));
        auto hit = g.hints.find(key);
        if (hit == g.hints.end() || nx < 2 || ny < 2 || nz < 2) return SDFK_OK;
        {   // A captured job only pays when the IDENTICAL job comes again: it is built on the second sighting of the full key
            // (program, grid, bounds, clip, iso) on this lane, not as soon as the grid shape has hints -- a caller whose
            // program or bounds change per call (an animated SDF) would otherwise pay an un-captured run, a capture, an
            // instantiation and a set of allocations on every call, and an eviction (a stream synchronisation) from the 25th on.
            uint64_t fk = fnv1a64(std::string((const char*)&p, sizeof p), 0xcbf29ce484222325ull);
            const int dims[5] = {nx, ny, nz, clip, lane};
            fk = fnv1a64(std::string((const char*)dims, sizeof dims), fk);
            fk = fnv1a64(std::string((const char*)mn, 12) + std::string((const char*)mx, 12) + std::string((const char*)&iso, 4), fk);
            if (g.graph_sightings.size() > 4096) g.graph_sightings.clear();
            if (g.graph_sightings[fk]++ == 0) return SDFK_OK;
        }
        if (alive >= 3 || g.graph_jobs.size() >= 24 || g.graph_bytes > (size_t(4) << 30)) {
            GraphJob* lru = nullptr;   // make room: the least recently used free one, if any
            for (GraphJob* c : g.graph_jobs)
                if (!c->busy && (!lru || c->last_use < lru->last_use)) lru = c;
            if (!lru || alive >= 3) return SDFK_OK;
            graph_job_destroy(lru);
        }
        const Context::Hint h = hit->second;
        q = new GraphJob();
        q->prog = p; const_cast<sdfk_program*>(p)->refs++;
        q->nx = nx; q->ny = ny; q->nz = nz; q->clip = clip; q->lane = lane; q->iso = iso; q->key = key;
        memcpy(q->mn, mn, 12); memcpy(q->mx, mx, 12);
        g.graph_jobs.push_back(q);
        const size_t before = [] { size_t b = 0; for (auto& kv : g.live_blocks) b += kv.second.size; return b; }();
        int r = sdfk_volume_create(nx, ny, nz, mn, mx, p->writes_color ? 1 : 0, &q->vol);
        // (first run outside the capture: compiles / loads the kernels, allocates the sign-bit arrays, marks the volume as this program's output)
        if (!r) r = sample_impl(p, q->vol, clip, iso);
        if (!r) r = setup_job(q->vol, iso, 1, 0, std::max(nz - 1, 0), (size_t)h.n_active + h.n_active / 4 + 4096, &q->job);
        if (!r && (q->job->empty || !q->job->have_bits)) r = -1;   // (an iso value that never compares equal, NaN, leaves the sign-bit pass to the job)
        if (!r) r = alloc_mesh(&q->proto, (size_t)h.nv + h.nv / 4 + 4096, (size_t)h.ni + h.ni / 4 + 12288);
        if (!r) r = launch_classify(q->job, false);
        if (!r) r = launch_emit(q->job, q->proto, 0);   // (allocates the AABB partials: the captured run below does not allocate)
        if (!r) {
            hipStream_t st = lane_stream(lane);
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rc = sample_impl(p, q->vol, clip, iso);
                if (!rc) rc = launch_classify(q->job, false);
                if (!rc) rc = launch_emit(q->job, q->proto, 0);
                e = hipStreamEndCapture(st, &q->graph);
                if (e == hipSuccess && rc) e = hipErrorUnknown;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&q->ran, hipEventDisableTiming);
            if (e != hipSuccess) { (void)hipGetLastError(); r = -1; }
        }
        if (r) {   // graphs are an optimisation: any failure here means "take the ordinary path"
            g_graph_build_failures++;
            graph_job_destroy(q);
            return SDFK_OK;
        }
        const size_t after = [] { size_t b = 0; for (auto& kv : g.live_blocks) b += kv.second.size; return b; }();
        q->bytes = after > before ? after - before : 0;
        g.graph_bytes += q->bytes;
        // (the un-captured first run above already queued this call's job: no graph launch for it)
    } else {
        if (hipEventQuery(q->ran) != hipSuccess) (void)hipEventSynchronize(q->ran);   // the previous run raises `overflow` in the slot itself
        g.slots[q->job->slot].c.overflow = 0;
        const hipError_t e = hipGraphLaunch(q->exec, lane_stream(lane));
        if (e != hipSuccess) { (void)hipGetLastError(); q->stale = true; return SDFK_OK; }
        g.graph_launches++;
    }
    q->busy = true;
    q->last_use = ++g.graph_clock;
    sdfk_mesh* m = new sdfk_mesh();
    const sdfk_mesh* pr = q->proto;
    m->vertices = pr->vertices; m->colors = pr->colors; m->normals = pr->normals; m->triangles = pr->triangles; m->bounds = pr->bounds;
    m->cap_v = pr->cap_v; m->cap_i = pr->cap_i; m->has_colors = pr->has_colors; m->colors_valid = pr->colors_valid; m->lane = lane;
    m->borrowed = true;
    m->graph_job = q;
    q->borrower = m;
    hipError_t e = hipEventCreateWithFlags(&m->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(m->done, lane_stream(lane));
    if (e == hipSuccess) e = hipEventRecord(q->ran, lane_stream(lane));
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(lane_stream(lane));
        if (m->done) (void)hipEventDestroy(m->done);
        m->borrowed = false; m->vertices = m->colors = m->normals = m->bounds = nullptr; m->triangles = nullptr;
        q->busy = false;
        q->borrower = nullptr;
        delete m;
        return fail(SDFK_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
    }
    m->pending = q->job;
    m->src = q->vol;
    m->iso = iso; m->step = 1; m->layer_begin = 0; m->layer_end = std::max(nz - 1, 0);
    m->vertex_base = 0; m->key = q->key;
    g.pending.push_back(m);
    while (g.pending.size() > Context::MAX_PENDING) (void)mesh_resolve(g.pending.front());
    *out = m;
    return SDFK_OK;
}

// What a sharded step with compact payloads queues right behind its last kernel: the plain payload at the step's dst ->
// the compact payload at `out` (k_payload_compact, mc_kernels.hip).  Part of the captured step graph.
int launch_post_compact(const PostCompact* pc, const void* plain, int64_t plain_capacity)
{
    if (!pc) return SDFK_OK;
    const int64_t words = plain_capacity / 4;
    ProfScope ps("k_payload_compact");
    hipLaunchKernelGGL(k_payload_compact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((words + 4095) / 4096, 1024))), dim3(256), 0, g.stream,
                       (const char*)plain, pc->out, pc->out_capacity, pc->ticket);
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

// The same for a sharded step (sdfk_slab_enqueue on a lane): sample the caller's slab, mesh its layers straight into the
// caller's send buffer, header written by the last kernel -- eleven launches as ONE hipGraphLaunch per step.  Key: program,
// slab volume, clip, iso, layer range, destination and capacity, lane; the GraphJob keeps the job's workspace and result
// slot, the mesh arrays ARE sections of the destination.  A step whose capacities were too small says so in its header
// (every rank then redoes it exactly, sdfkit_amd/dist.py), which renews the size hints: a GraphJob built from other hints
// than today's is rebuilt.  *handled stays false when graphs do not apply (the caller then takes the ordinary path).
int graph_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int clip, float iso, int lb, int le, void* dst, int64_t capacity, bool* handled,
                       const PostCompact* pc)
{
    *handled = false;
    const int lane = g.cur_lane;
    if (lane == 0 || g.prof_on || g.sampler_only || g_graph_build_failures >= 8) return SDFK_OK;
    const uint64_t key = hint_key(slab, 1, lb, le);
    auto hit = g.hints.find(key);
    if (hit == g.hints.end()) return SDFK_OK;
    const Context::Hint h = hit->second;
    GraphJob* q = nullptr;
    for (GraphJob* c : g.graph_jobs)
        if (c->slab && c->prog == p && c->ext_vol == slab && c->clip == clip && memcmp(&c->iso, &iso, 4) == 0 && c->lb == lb && c->le == le &&
            c->dst == dst && c->capacity == capacity && c->lane == lane && c->post_out == (pc ? (void*)pc->out : nullptr)) { q = c; break; }
    if (q && memcmp(&q->hint, &h, sizeof h) != 0) { graph_job_destroy(q); q = nullptr; }
    if (q) {
        resolve_dependents(slab);
        volume_values_changed(slab);
        const hipError_t e = hipGraphLaunch(q->exec, lane_stream(lane));
        if (e != hipSuccess) { (void)hipGetLastError(); graph_job_destroy(q); return SDFK_OK; }
        // (what sample_impl leaves in the volume: its values, sign bits and colours are this program's output again)
        sdfk_program* pp = const_cast<sdfk_program*>(p);
        pp->refs++;
        slab->sampled_by = pp;
        slab->sampled_args = q->args;
        slab->bits_iso = iso;
        slab->bits_valid = true;
        g.graph_launches++;
        q->last_use = ++g.graph_clock;
        *handled = true;
        return SDFK_OK;
    }
    // build.  Cheap misfits first (nothing queued yet): they are not failures, the ordinary path packs instead.
    if (slab->nx < 2 || slab->ny < 2 || slab->nz < 2 || lb >= le) return SDFK_OK;
    sdfk_mesh* proto = nullptr;
    if (!external_mesh((char*)dst, capacity, slab->colors != nullptr, h.nv, h.ni, &proto)) return SDFK_OK;
    if (g.graph_jobs.size() >= 24) {
        GraphJob* lru = nullptr;
        for (GraphJob* c : g.graph_jobs)
            if (!c->busy && (!lru || c->last_use < lru->last_use)) lru = c;
        if (!lru) { sdfk_mesh_free(proto); return SDFK_OK; }
        graph_job_destroy(lru);
    }
    q = new GraphJob();
    q->slab = true;
    q->prog = p; const_cast<sdfk_program*>(p)->refs++;
    q->ext_vol = slab; q->clip = clip; q->iso = iso; q->lb = lb; q->le = le; q->dst = dst; q->capacity = capacity; q->lane = lane;
    q->nx = slab->nx; q->ny = slab->ny; q->nz = slab->nz;
    q->key = key; q->hint = h; q->proto = proto;
    q->post_out = pc ? (void*)pc->out : nullptr;
    g.graph_jobs.push_back(q);
    int r = sample_impl(p, slab, clip, iso);   // (outside the capture: loads the kernels, allocates the sign-bit arrays; this call's run)
    if (!r) r = setup_job(slab, iso, 1, lb, le, (size_t)h.n_active + h.n_active / 4 + 4096, &q->job);
    if (!r && (q->job->empty || !q->job->have_bits)) r = -1;
    if (!r) r = launch_classify(q->job, false);
    if (!r) r = launch_emit(q->job, q->proto, 0);
    if (!r) r = launch_post_compact(pc, dst, capacity);
    if (!r) {
        q->args = slab->sampled_args;
        hipStream_t st = lane_stream(lane);
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            int rc = sample_impl(p, slab, clip, iso);
            if (!rc) rc = launch_classify(q->job, false);
            if (!rc) rc = launch_emit(q->job, q->proto, 0);
            if (!rc) rc = launch_post_compact(pc, dst, capacity);
            e = hipStreamEndCapture(st, &q->graph);
            if (e == hipSuccess && rc) e = hipErrorUnknown;
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); r = -1; }
    }
    if (r) {   // (whatever was queued is harmless: the ordinary path redoes the step behind it on the same lane)
        g_graph_build_failures++;
        graph_job_destroy(q);
        return SDFK_OK;
    }
    q->last_use = ++g.graph_clock;
    *handled = true;
    return SDFK_OK;
}



extern "C" int sdfk_graph_stats(int64_t* jobs, int64_t* launches, int64_t* device_bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (jobs) *jobs = (int64_t)g.graph_jobs.size();
    if (launches) *launches = g.graph_launches;
    if (device_bytes) *device_bytes = (int64_t)g.graph_bytes;
    return SDFK_OK;
}

extern "C" int sdfk_march_begin(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                                sdfk_march_job** job, int64_t* n_vertices, int64_t* n_indices)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !job) return fail(SDFK_ERR_INVALID, "sdfk_march_begin: null argument");
    *job = nullptr;
    if (int r = require_init()) return r;
    sdfk_march_job* j = nullptr;
    int r = setup_job(v, iso_value, 1, layer_begin, layer_end, 0, &j);
    if (r) return r;
    if (!j->empty) {
        r = launch_classify(j, true);
        r = r ? r : wait_counters(j);
        if (!r && j->c.n_active > j->P.cap_active) {
            r = alloc_records(j, j->c.n_active);
            r = r ? r : launch_classify(j, true);
            r = r ? r : wait_counters(j);
        }
        if (r) { job_release(j); delete j; return r; }
    }
    if (n_vertices) *n_vertices = (int64_t)j->c.total_v - (int64_t)j->c.nghost;
    if (n_indices) *n_indices = (int64_t)j->c.total_t * 3;
    *job = j;
    return SDFK_OK;
}

extern "C" int sdfk_march_finish(sdfk_march_job* j, int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!j || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_finish: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (j->finished) return fail(SDFK_ERR_INVALID, "march job already finished");
    const int64_t nv = (int64_t)j->c.total_v - (int64_t)j->c.nghost;
    if (vertex_base + nv >= (int64_t(1) << 31)) return fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
    sdfk_mesh* m = nullptr;
    int r = alloc_mesh(&m, (size_t)nv, (size_t)j->c.total_t * 3);
    if (r) return r;
    j->finished = true;
    const bool work = !j->empty && j->c.n_active > 0;
    if (work) {
        r = launch_emit(j, m, vertex_base);
        r = r ? r : wait_counters(j);
        if (!r && j->c.overflow) r = fail(SDFK_ERR_HIP, "marching cubes: output capacity exceeded unexpectedly");
        if (r) { sdfk_mesh_free(m); return r; }
    }
    finalize_mesh(j, m, work);
    *out = m;
    return SDFK_OK;
}

extern "C" void sdfk_march_job_free(sdfk_march_job* job)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!job) return;
    job_release(job);   // stream-ordered pool: no sync needed
    delete job;
}

extern "C" int sdfk_march(const sdfk_volume* v, float iso_value, int32_t step, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !out) return fail(SDFK_ERR_INVALID, "sdfk_march: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (v->z0 != 0 || v->nz != v->nz_global) return fail(SDFK_ERR_INVALID, "sdfk_march needs a whole volume; use sdfk_march_slab for slabs");
    return march_range(v, iso_value, step, 0, std::max(v->nz_global - 1, 0), 0, out);
}

extern "C" int sdfk_march_slab(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                               int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_slab: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    return march_range(v, iso_value, 1, layer_begin, layer_end, vertex_base, out);
}

extern "C" int sdfk_sample_march_slab(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                                      int32_t layer_begin, int32_t layer_end, int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !slab || !out) return fail(SDFK_ERR_INVALID, "sdfk_sample_march_slab: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (int r = sample_impl(p, slab, clip_to_bounds, iso_value)) return r;
    return march_range(slab, iso_value, 1, layer_begin, layer_end, vertex_base, out);
}

extern "C" int sdfk_mesh_pack(const sdfk_mesh* m, void* dst, int64_t capacity_bytes, int64_t* needed_bytes)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m || !dst) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: null argument");
    if (int r = require_init()) return r;
    if (m->lane != g.cur_lane) const_cast<sdfk_mesh*>(m)->used_on_main = true;
    if (m->pending && !m->status) {
        // the job is still queued: the device packs from the job's own counters, the host is
        // not involved (needed_bytes is unknown here: -1; the header carries the counts)
        if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: capacity below the header size");
        PackArgs A;
        A.counters = m->pending->P.counters;
        A.cap_active = m->pending->P.cap_active;
        A.cap_v = (uint32_t)std::min<size_t>(m->cap_v, 0xffffffffu);
        A.cap_i = m->cap_i;
        A.vertices = m->vertices; A.colors = m->colors; A.normals = m->normals; A.triangles = m->triangles;
        A.bounds = m->bounds;
        A.dst = (char*)dst;
        A.capacity = capacity_bytes;
        A.vbytes = m->has_colors ? 36 : 24;
        if (m->lane != g.cur_lane) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: queued mesh belongs to another stream");
        ProfScope ps("k_pack");
        hipLaunchKernelGGL(k_pack_pending, dim3(grid_for(m->cap_v * 9 + m->cap_i, 256, 1024)), dim3(256), 0, g.stream, A);
        HIPCHK(hipGetLastError());
        if (needed_bytes) *needed_bytes = -1;
        return SDFK_OK;
    }
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const int vbytes = m->has_colors ? 36 : 24;
    const int64_t vb = m->nv * 12, need = SDFK_SLAB_HEADER_BYTES + (int64_t)vbytes * m->nv + m->ni * 4;
    if (needed_bytes) *needed_bytes = need;
    if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: capacity below the header size");
    hipLaunchKernelGGL(k_slab_header, dim3(1), dim3(64), 0, g.stream, (SlabHeader*)dst, (int64_t)m->nv, (int64_t)m->ni,
                       (const float*)m->bounds, vbytes);
    HIPCHK(hipGetLastError());
    if (need > capacity_bytes) return SDFK_OK;
    char* q = (char*)dst + SDFK_SLAB_HEADER_BYTES;
    if (vb) {
        HIPCHK(hipMemcpyAsync(q, m->vertices, vb, hipMemcpyDeviceToDevice, g.stream));
        q += vb;
        if (m->has_colors) {
            HIPCHK(hipMemcpyAsync(q, m->colors, vb, hipMemcpyDeviceToDevice, g.stream));
            q += vb;
        }
        HIPCHK(hipMemcpyAsync(q, m->normals, vb, hipMemcpyDeviceToDevice, g.stream));
        q += vb;
    }
    if (m->ni) HIPCHK(hipMemcpyAsync(q, m->triangles, m->ni * 4, hipMemcpyDeviceToDevice, g.stream));
    return SDFK_OK;
}

// One sharded step of a rank: [lane section] sample the slab -> mesh it into the payload at dst.  caller_stream_waits: the
// caller's stream waits for the section (sdfk_slab_enqueue); the library's own step driver (dist_rccl.h) orders its exchange
// stream with an event of its own instead.
int slab_enqueue_impl(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value, int32_t layer_begin,
                             int32_t layer_end, void* dst, int64_t capacity_bytes, int32_t lane, void* wait_hip_event, bool caller_stream_waits,
                             const PostCompact* pc)
{
    int r = lane > 0 ? sdfk_lane_begin(lane, wait_hip_event) : SDFK_OK;
    if (r) return r;
    sdfk_mesh* m = nullptr;
    r = require_init();
    if (!r && lane > 0 && graphs_enabled_slab((int64_t)slab->nx * slab->ny * slab->nz)) {   // the repeat step as ONE captured graph launch
        bool handled = false;
        r = graph_slab_enqueue(p, slab, clip_to_bounds ? 1 : 0, iso_value, layer_begin, layer_end, dst, capacity_bytes, &handled, pc);
        if (handled || r) {
            const int r2 = sdfk_lane_end(caller_stream_waits ? 1 : 0);
            return r ? r : r2;
        }
    }
    if (!r) r = sample_impl(p, slab, clip_to_bounds, iso_value);
    // With size hints for this slab shape the mesh is emitted STRAIGHT into the payload at dst (its arrays are the
    // payload's sections, k_triangles writes the header): no pack launch, no second copy of the mesh.
    if (!r) r = march_range(slab, iso_value, 1, layer_begin, layer_end, 0, &m, (char*)dst, capacity_bytes);
    if (!r && !(m->external && m->pending)) {   // first call of a shape (exact path) or hints that do not fit: pack
        int64_t need = 0;
        r = sdfk_mesh_pack(m, dst, capacity_bytes, &need);
    }
    if (m) sdfk_mesh_free(m);   // stream-ordered: the kernels above still use it
    if (!r) r = launch_post_compact(pc, dst, capacity_bytes);
    if (lane > 0) {
        const int r2 = sdfk_lane_end(caller_stream_waits ? 1 : 0);
        if (!r) r = r2;
    }
    return r;
}

extern "C" int sdfk_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                                 int32_t layer_begin, int32_t layer_end, void* dst, int64_t capacity_bytes,
                                 int32_t lane, void* wait_hip_event)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !slab || !dst) return fail(SDFK_ERR_INVALID, "sdfk_slab_enqueue: null argument");
    if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_slab_enqueue: capacity below the header size");
    return slab_enqueue_impl(p, slab, clip_to_bounds, iso_value, layer_begin, layer_end, dst, capacity_bytes, lane, wait_hip_event, true);
}

int slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!gathered || world < 1 || stride_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_slabs_rebase: bad argument");
    if (int r = require_init()) return r;
    if (world == 1 && !headers_mirror) return SDFK_OK;
    hipLaunchKernelGGL(k_slabs_rebase, dim3(64, world), dim3(256), 0, g.stream, (char*)gathered, (int)world, (int64_t)stride_bytes,
                       (SlabHeader*)headers_mirror, 0);
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

extern "C" int sdfk_slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes)
{
    return slabs_rebase(gathered, world, stride_bytes, nullptr);
}

extern "C" int sdfk_slabs_rebase_mirror(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror)
{
    if (!headers_mirror) return fail(SDFK_ERR_INVALID, "sdfk_slabs_rebase_mirror: null mirror");
    return slabs_rebase(gathered, world, stride_bytes, headers_mirror);
}

extern "C" int sdfk_march_host(const float* values, const float* colors3, int32_t nx, int32_t ny, int32_t nz,
                               const float min[3], const float max[3], float iso_value, int32_t step,
                               sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!values || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_host: null argument");
    sdfk_volume* v = nullptr;
    int r = sdfk_volume_create(nx, ny, nz, min, max, colors3 ? 1 : 0, &v);
    if (r) return r;
    r = sdfk_volume_upload(v, values, colors3);
    if (!r) r = sdfk_march(v, iso_value, step, out);
    if (!r && (*out)->pending && (*out)->src == v) (*out)->owns_src = true;   // freed when the mesh is resolved
    else sdfk_volume_free(v);
    return r;
}

extern "C" int sdfk_sample_march(const sdfk_program* p, const float min[3], const float max[3],
                                 int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds,
                                 float iso_value, int32_t step, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !out) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: null argument");
    if (int r = require_init()) return r;
    // self-contained job (no input but the program; the output is only read after a host-side
    // wait): consecutive calls alternate between the side lanes and overlap on the GPU
    int lane = g.cur_lane;   // (inside sdfk_lane_begin/end: the caller's lane)
    // SDFK_OPT_LANES, default three: small grids are launch-latency bound (9 dependent launches; 256^3: 39 instead of 48 us
    // per step), and from 512^3 up a third job in flight is worth 4-5 % since the sampler of a .W-only program holds no LDS
    // any more and the meshing kernels of two other jobs fit next to it (512^3 sphere 0.172 -> 0.164 ms, 384^3 0.092 ->
    // 0.087, 1024^3 1.23 -> 1.18; colour scenes unchanged)
    g.side_lanes = std::max(0, std::min(Context::NSIDE, g_cfg.lanes));
    if (g.side_lanes == 1) g.side_lanes = 0;   // one side lane overlaps nothing
    // SDFK_OPT_IDLE_LANE: a fourth lane for launch-bound grids (the captured-graph jobs) while the caller's stream has nothing
    // queued -- the fourth lane's stream sits in the class of the caller's stream ("stream placement": they must not be busy
    // together), which is the one class a job can use when the caller itself is not using it: 256^3 35 instead of 39 us per step
    // ... and for volume-less jobs (no sampling kernel: the same kind of chain of short dependent launches; 512^3 sphere 0.0870 ->
    // 0.0846 ms per pipelined step with the fourth lane, while a job that STORES its volume loses 0.6 % to it)
    const bool chain_only = step == 1 && g_cfg.elide_volume && g_cfg.corner_eval && g_cfg.vcolor_eval && !p->no_elide && iso_value == iso_value &&
                            !g.sampler_only;
    if (g.side_lanes == 3 && g_cfg.idle_lane && Context::NSIDE >= 4 && g.cur_lane == 0 && g.placed && step == 1 &&
        (graphs_enabled((int64_t)nx * ny * nz) || chain_only)) {
        bool placed4 = false;
        for (const auto& q : g.pool) placed4 = placed4 || (q.user == 4 && q.s == g.lanes[4].stream);
        if (placed4) {
            if (hipStreamQuery(g.lanes[0].stream) == hipSuccess) g.side_lanes = 4;
            else (void)hipGetLastError();   // (hipErrorNotReady is an answer, not an error)
        }
    }
    if (g.side_lanes > 0 && g.cur_lane == 0) {
        if (g.next_side >= g.side_lanes) g.next_side = 0;
        lane = 1 + g.next_side;
        g.next_side = (g.next_side + 1) % g.side_lanes;
    }
    LaneScope on_lane(lane);
    *out = nullptr;
    if (step == 1 && graphs_enabled((int64_t)nx * ny * nz)) {   // launch-bound grids: the whole job as one captured graph
        if (int r = graph_sample_march(p, min, max, nx, ny, nz, clip_to_bounds ? 1 : 0, iso_value, out)) return r;
        if (*out) return SDFK_OK;
    }
    sdfk_volume* v = nullptr;
    // (SDFK_OPT_ELIDE_VOLUME applies to grids above the captured-graph limit: a launch-bound grid gains nothing from one launch more
    // and 4 bytes per voxel less, and its captured job -- built on the second sighting of a key -- stores its volume: one kernel
    // set per program structure either way)
    const bool elidable = step == 1 && !graphs_enabled((int64_t)nx * ny * nz);
    int r = elidable ? job_volume_create(p, nx, ny, nz, min, max, iso_value, &v) : sdfk_volume_create(nx, ny, nz, min, max, p->writes_color ? 1 : 0, &v);
    if (r) return r;
    r = require_init();
    {
        // SDFK_OPT_TOKENS -- bit 0: sampling kernels apart, bit 1: k_vertices apart.  Default: the sampling kernels of grids from
        // 2^27 voxels up (512^3 sphere 0.162 -> 0.156 ms per step, 768^3 0.475 -> 0.462, 1024^3 1.12-1.20 -> 1.10, README scene
        // 0.49 -> 0.478; 384^3 and below lose 2-3 %: there a sampling kernel is too short to be worth a cross-stream wait).
        // k_vertices apart costs 2-3 % at every size: its workgroups are long-lived, and a second one fills the first one's tail.
        const int dflt = (int64_t)nx * ny * nz >= (int64_t(1) << 27) ? 1 : 0;
        g.token_mask = (lane > 0 && g.side_lanes > 1) ? (g_cfg.tokens >= 0 ? g_cfg.tokens : dflt) : 0;
    }
    if (!r) r = sample_impl(p, v, clip_to_bounds, step == 1 ? iso_value : 0.0f);
    if (!r) r = sdfk_march(v, iso_value, step, out);
    g.token_mask = 0;
    if (!r && (*out)->pending && (*out)->src == v) (*out)->owns_src = true;   // freed when the mesh is resolved
    else sdfk_volume_free(v);
    return r;
}

