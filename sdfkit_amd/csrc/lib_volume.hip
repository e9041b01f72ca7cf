// lib_volume.hip -- volumes (Voxels, Voxels.cs:8-65) and the sampling side of the path: Voxels.SampleSdf, ClipToBounds, SdfEx.Sample, RayMarcher.
#include "lib_internal.h"

// `values` no longer are what a program computed / what the cached sign bits describe
void volume_values_changed(sdfk_volume* v)
{
    v->bits_valid = false;
    if (v->sampled_by) program_release(v->sampled_by);
    v->sampled_by = nullptr;
}


// ---------------------------------------------------------------------------
// volumes
// ---------------------------------------------------------------------------
extern "C" int sdfk_volume_create_slab(int32_t nx, int32_t ny, int32_t nz_global, const float min[3],
                                       const float max[3], int32_t z0, int32_t nz_local,
                                       int32_t with_colors, sdfk_volume** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out || !min || !max) return fail(SDFK_ERR_INVALID, "sdfk_volume_create: null argument");
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz_global < 1 || nz_local < 1 || z0 < 0 || z0 + nz_local > nz_global)
        return fail(SDFK_ERR_INVALID, "sdfk_volume_create: bad dimensions %dx%dx%d (slab z0=%d nz=%d)", nx, ny, nz_global, z0, nz_local);
    if ((int64_t)nx * ny * nz_global >= (int64_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "grid exceeds the reference's int32 linear index (Voxels.cs:82)");
    if ((int64_t)nx * ny * ((nz_local + 3) & ~3) >= (int64_t(1) << 31))   // (the kernels index the PITCHED rows with 32-bit plane offsets)
        return fail(SDFK_ERR_INVALID, "grid with rows padded to %d voxels exceeds 2^31 voxel slots", (nz_local + 3) & ~3);
    if (int r = require_init()) return r;
    sdfk_volume* v = new sdfk_volume();
    v->nx = nx; v->ny = ny; v->nz = nz_local; v->nz_global = nz_global; v->z0 = z0;
    memcpy(v->gmin, min, sizeof v->gmin);
    memcpy(v->gmax, max, sizeof v->gmax);
    int r = dev_alloc((void**)&v->values, v->nalloc() * sizeof(float));
    if (!r && with_colors) r = dev_alloc((void**)&v->colors, v->nalloc() * 3 * sizeof(float));
    if (r) { dev_free(v->values); delete v; return r; }
    *out = v;
    return SDFK_OK;
}

extern "C" int sdfk_volume_create(int32_t nx, int32_t ny, int32_t nz, const float min[3], const float max[3],
                                  int32_t with_colors, sdfk_volume** out)
{
    return sdfk_volume_create_slab(nx, ny, nz, min, max, 0, nz, with_colors, out);
}


// The temporary volume of a self-contained sample -> mesh job.  With SDFK_OPT_ELIDE_VOLUME (and both re-evaluation paths on) it
// has no Values / Colors storage at all: nobody can ask for them (the volume never leaves the library).
int job_volume_create(const sdfk_program* p, int nx, int ny, int nz, const float mn[3], const float mx[3], float iso, sdfk_volume** out)
{
    if (!out || !mn || !mx) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: null argument");   // (as sdfk_volume_create_slab answers on the stored path)
    // Not elided: a NaN iso value (it never compares equal, so the cached sign bits never match and the sign-bit pass of the
    // meshing job would have to READ the voxels: the stored path returns its empty mesh) and the sampler-only measurement mode
    // (sdfk_profile_enable(2) leaves no valid sign bits behind) -- the option must never change a status code.
    const bool elide = g_cfg.elide_volume && g_cfg.corner_eval && g_cfg.vcolor_eval && !p->no_elide && iso == iso && !g.sampler_only;
    if (!elide) return sdfk_volume_create(nx, ny, nz, mn, mx, p->writes_color ? 1 : 0, out);
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz < 1) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: bad dimensions %dx%dx%d", nx, ny, nz);
    if ((int64_t)nx * ny * ((nz + 3) & ~3) >= (int64_t(1) << 31)) return fail(SDFK_ERR_INVALID, "grid exceeds the reference's int32 linear index (Voxels.cs:82)");
    sdfk_volume* v = new sdfk_volume();
    v->nx = nx; v->ny = ny; v->nz = nz; v->nz_global = nz; v->z0 = 0;
    memcpy(v->gmin, mn, sizeof v->gmin);
    memcpy(v->gmax, mx, sizeof v->gmax);
    v->elided = true;
    v->elided_colors = p->writes_color != 0;
    *out = v;
    return SDFK_OK;
}

// An elided volume gets its storage after all (the rare fall-back: a volume with case-13 sign words needs the dead-cell test of
// k_resolve, which reads neighbouring voxels) and is sampled again, this time with stores.
int sample_impl(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds, float iso_hint);
int volume_materialize(sdfk_volume* v)
{
    if (!v->elided) return SDFK_OK;
    sdfk_program* p = v->sampled_by;
    if (!p) return fail(SDFK_ERR_INVALID, "an elided volume without its program");
    int r = dev_alloc((void**)&v->values, v->nalloc() * sizeof(float));
    if (!r && v->elided_colors) r = dev_alloc((void**)&v->colors, v->nalloc() * 3 * sizeof(float));
    if (r) { dev_free(v->values); v->values = nullptr; return r; }
    v->elided = false;
    p->no_elide = true;
    const int clip = v->sampled_args.clip;
    const float iso = v->bits_iso;
    p->refs++;                       // (sample_impl drops the volume's reference before it takes a new one)
    r = sample_impl(p, v, clip, iso);
    program_release(p);
    return r;
}


extern "C" void sdfk_volume_free(sdfk_volume* v)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!v) return;
    graph_jobs_forget_volume(v);   // (captured slab steps write into it)
    resolve_dependents(v);
    volume_values_changed(v);   // (drops the reference to the program that sampled it)
    // no sync: the pool is stream-ordered (every kernel and copy runs on g.stream, so a block
    // handed out again is only touched by work queued after its previous user)
    dev_free(v->values);
    dev_free(v->colors);
    dev_free(v->bits);
    dev_free(v->bits8);
    dev_free(v->cull_list);
    if (v->cull_header) {   // (zero again, in its lane's order: the next volume-less job of that lane takes it as it is)
        if (v->cull_header_clean) g.lanes[v->cull_header_lane].clean_cull_headers.push_back(v->cull_header);
        else dev_free(v->cull_header);
    }
    delete v;
}

extern "C" int sdfk_volume_upload(sdfk_volume* v, const float* values, const float* colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !values) return fail(SDFK_ERR_INVALID, "sdfk_volume_upload: null argument");
    if (int r = require_init()) return r;
    if (colors3 && !v->colors) return fail(SDFK_ERR_INVALID, "sdfk_volume_upload: volume was created without colours");
    resolve_dependents(v);
    volume_values_changed(v);
    if (v->pitch() == v->nz) {
        HIPCHK(hipMemcpyAsync(v->values, values, v->nvox() * sizeof(float), hipMemcpyHostToDevice, g.stream));
        if (colors3) HIPCHK(hipMemcpyAsync(v->colors, colors3, v->nvox() * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));  // the caller's arrays are not retained
        return SDFK_OK;
    }
    // rows of nz % 4 != 0 voxels: the dense host layout goes to a temporary device array, a kernel spreads the rows out
    const size_t rows = (size_t)v->nx * v->ny;
    float* tmp = nullptr;
    if (int r = dev_alloc((void**)&tmp, v->nvox() * (colors3 ? 3 : 1) * sizeof(float))) return r;
    hipError_t e = hipMemcpyAsync(tmp, values, v->nvox() * sizeof(float), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_repitch<true>, dim3(grid_for(v->nvox())), dim3(256), 0, g.stream, tmp, v->values, rows, v->nz, v->pitch());
        if (colors3) {
            e = hipMemcpyAsync(tmp, colors3, v->nvox() * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream);
            if (e == hipSuccess)
                hipLaunchKernelGGL(k_repitch<true>, dim3(grid_for(v->nvox() * 3)), dim3(256), 0, g.stream, tmp, v->colors, rows, v->nz * 3, v->pitch() * 3);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);   // the caller's arrays are not retained
    dev_free(tmp);
    if (e != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_volume_upload: %s", hipGetErrorString(e));
    return SDFK_OK;
}

extern "C" int sdfk_volume_download(const sdfk_volume* v, float* values, float* colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v) return fail(SDFK_ERR_INVALID, "sdfk_volume_download: null volume");
    if (int r = require_init()) return r;
    std::vector<CopyPiece> pieces;
    float* dense_v = nullptr;   // rows of nz % 4 != 0 voxels: packed into dense temporaries first
    float* dense_c = nullptr;
    if (v->pitch() != v->nz) {
        const size_t rows = (size_t)v->nx * v->ny;
        if (values) {
            if (int r = dev_alloc((void**)&dense_v, v->nvox() * sizeof(float))) return r;
            hipLaunchKernelGGL(k_repitch<false>, dim3(grid_for(v->nvox())), dim3(256), 0, g.stream, v->values, dense_v, rows, v->nz, v->pitch());
        }
        if (colors3 && v->colors) {
            if (int r = dev_alloc((void**)&dense_c, v->nvox() * 3 * sizeof(float))) { dev_free(dense_v); return r; }
            hipLaunchKernelGGL(k_repitch<false>, dim3(grid_for(v->nvox() * 3)), dim3(256), 0, g.stream, v->colors, dense_c, rows, v->nz * 3, v->pitch() * 3);
        }
    }
    struct FreeTmp { float *a, *b; ~FreeTmp() { dev_free(a); dev_free(b); } } free_tmp{dense_v, dense_c};   // (stream-ordered pool; the copies below are synchronous)
    if (values) pieces.push_back({dense_v ? dense_v : v->values, values, v->nvox() * sizeof(float)});
    if (colors3 && v->colors) pieces.push_back({dense_c ? dense_c : v->colors, colors3, v->nvox() * 3 * sizeof(float)});
    if (colors3 && !v->colors) {   // colours that were never written are zero (Voxels.cs:88-92): cleared on the pool
        std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
        const size_t nb = v->nvox() * 3 * sizeof(float), per = size_t(2) << 20;
        char* c = (char*)colors3;
        g_pool.start((int)((nb + per - 1) / per), [=](int t) { const size_t a = (size_t)t * per; memset(c + a, 0, std::min(per, nb - a)); });
        g_pool.wait();
    }
    return copy_to_host(pieces);
}

extern "C" int sdfk_volume_row_pitch(const sdfk_volume* v, int32_t* pitch_voxels)
{
    if (!v || !pitch_voxels) return fail(SDFK_ERR_INVALID, "sdfk_volume_row_pitch: null argument");
    *pitch_voxels = v->pitch();
    return SDFK_OK;
}

extern "C" int sdfk_volume_device_ptrs(const sdfk_volume* v, void** values, void** colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!v) return fail(SDFK_ERR_INVALID, "null volume");
    resolve_dependents(v);   // the caller may write through these pointers: no cached view of
    volume_values_changed(const_cast<sdfk_volume*>(v));   // the values stays valid
    if (values) *values = v->values;
    if (colors3) *colors3 = v->colors;
    return SDFK_OK;
}

// Grid dimensions y and z hold at most 65535 workgroups; extents that may exceed that go to x, or are split over z and y.
dim3 transpose_grid(int nz, int ny, int nxw)
{
    const unsigned groups = (unsigned)((nxw + 7) / 8);
    return dim3((unsigned)((nz + 127) / 128) * (unsigned)ny, (groups + 65534u) / 65535u, std::min(groups, 65535u));
}
dim3 flat_grid(size_t plane, int nx8)   // plane-chunk kernels: (chunks of the (y, z) plane, x groups beyond 65535, x groups)
{
    return dim3((unsigned)((plane + 255) / 256), ((unsigned)nx8 + 65534u) / 65535u, std::min((unsigned)nx8, 65535u));
}

// Voxels.cs:32-34,81,139: cell size, first cell centre and ClipToBounds value, in float
void grid_constants(const sdfk_volume* v, float d[3], float m[3], float* outside)
{
    const int n[3] = {v->nx, v->ny, v->nz_global};
    for (int k = 0; k < 3; k++) {
        d[k] = n[k] >= 1 ? (v->gmax[k] - v->gmin[k]) / (float)n[k] : 0.0f;
        const float h = 0.5f * d[k];
        m[k] = v->gmin[k] + h;
    }
    *outside = (v->gmax[0] - v->gmin[0]) / (float)v->nx;
}


int sample_impl(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds, float iso_hint)
{
    SampleArgs A;
    memset(&A, 0, sizeof A);
    float d[3], m[3], outside;
    grid_constants(v, d, m, &outside);
    A.values = v->values;
    A.colors = v->colors;
    A.mx = m[0]; A.my = m[1]; A.mz = m[2];
    A.dx = d[0]; A.dy = d[1]; A.dz = d[2];
    A.nx = v->nx; A.ny = v->ny; A.nz = v->nz;
    A.z0 = v->z0; A.nz_global = v->nz_global;
    A.clip = clip_to_bounds ? 1 : 0;
    A.outside = outside;
    resolve_dependents(v);
    volume_values_changed(v);
    auto sampled = [&]() {   // from here on `values` are this program's output
        sdfk_program* q = const_cast<sdfk_program*>(p);
        q->refs++;
        v->sampled_by = q;
        v->sampled_args = A;
    };
    void* params[] = {&A, p->kargs()};
    {
        // fused sampling + sign bits (iso known or guessed 0): marching cubes then skips its
        // dense pass over the volume
        if (!v->bits) {
            if (int r = dev_alloc((void**)&v->bits, v->nbitwords() * sizeof(uint64_t))) return r;
        }
        if (!v->bits8 && !(v->elided && g_cfg.elide_volume >= 2)) {   // (block culling writes the words directly: no byte form)
            if (int r = dev_alloc((void**)&v->bits8, (size_t)v->ny * v->nx8() * v->pitch8() + 64)) return r;
        }
        A.bits8 = v->bits8;
        A.nx8 = v->nx8();
        A.pitch8 = v->pitch8();
        A.iso = iso_hint;
        {
            const unsigned tpb = 512u / (unsigned)kSampleRpw;
            // 0: z tiles of one y row (nz % 256 == 0); 1: 256-voxel chunks of the (y, z) plane of an x row (any nz:
            // rows are padded to a multiple of 4 voxels)
            const int force = g_cfg.sample_mode;   // (debugging: SDFK_SAMPLE_MODE at start-up)
            int mode = (v->nz % 256) == 0 ? 0 : 1;
            if (force == 0 && (v->nz & 3) == 0) mode = 0;
            if (force == 1 || v->ny > 65535 || v->nx8() > 65535) mode = 1;   // (the row-tiled form has y and x/8 in 16-bit grid dimensions)
            static const char* const names[2][2] = {{"sdfk_sample_bits", "sdfk_sample_bits_flat"}, {"sdfk_sample_bits_clip", "sdfk_sample_bits_clip_flat"}};
            hipFunction_t fn = nullptr, fn_colors = nullptr;   // (compiled on first use)
            // Two passes for a colour volume (sample_codegen.h, "two-pass sampling"): values + sign bytes with this tile's kernel, then the
            // colour array as ONE linear stream.  Worth it when the program is cheap enough to evaluate twice and the grid is large enough
            // for the store rate to matter; SDFK_OPT_COLOR_PASSES = 1 / 2 force one / two passes at any size.
            const bool two_pass = !v->elided && p->writes_color && v->colors &&
                                  (g_cfg.color_passes == 2 || (g_cfg.color_passes == 0 && p->n_ops <= kTwoPassMaxOps && v->nvox() >= kTwoPassMinVoxels));
            const int pk = v->elided ? (mode ? PK_SIGNS_FLAT : PK_SIGNS)
                         : two_pass ? (clip_to_bounds ? PK_BITS_NC_CLIP : PK_BITS_NC) + mode
                                    : (clip_to_bounds ? PK_BITS_CLIP : PK_BITS) + mode;
            if (!(v->elided && g_cfg.elide_volume >= 2))
                if (int r = program_fn(p, pk, &fn)) return r;
            if (two_pass)   // (after the first pass's kernel, whose module brings sdfk_sample_colors along: one compile, not two)
                if (int r = program_fn(p, PK_COLORS, &fn_colors)) return r;
            const bool cull = v->elided && g_cfg.elide_volume >= 2;
            // (the name rocprofv3 shows for the entry point launched; the two culling kernels have scopes of their own)
            static const char* const names_nc[2][2] = {{"sdfk_sample_bits_nc", "sdfk_sample_bits_nc_flat"}, {"sdfk_sample_bits_nc_clip", "sdfk_sample_bits_nc_clip_flat"}};
            std::unique_ptr<ProfScope> ps(new ProfScope(cull ? nullptr : (v->elided ? (mode ? "sdfk_sample_signs_flat" : "sdfk_sample_signs")
                                                                                      : (two_pass ? names_nc : names)[clip_to_bounds ? 1 : 0][mode])));
            const size_t plane = (size_t)v->ny * v->pitch();
            phase_token_wait(0);
            if (cull) {
                // block culling: one lane per 64 x 4 x 4 block decides it by interval arithmetic (constant sign words) or lists it; the
                // listed blocks -- those the surface passes through -- are evaluated voxel by voxel; both write the X-packed sign
                // words themselves (sample_codegen.h): no byte form, no transposer
                struct { unsigned long long* bits; unsigned* worklist; unsigned* counter; int nbx, nby, nbz; int cpw; unsigned region; } Cargs;   // (= CullArgs of sample_codegen.h)
                Cargs.bits = (unsigned long long*)v->bits;
                Cargs.nbx = v->nxw(); Cargs.nby = (v->ny + 3) / 4; Cargs.nbz = (v->nz + 3) / 4;
                const size_t nblocks = (size_t)Cargs.nbx * Cargs.nby * Cargs.nbz;
                // 64 sub-lists (SDFK_CULL_LISTS), their counters 128 bytes apart in front: workgroup w appends to sub-list w % 64, whose region
                // holds what its share of the workgroups can list (at most 1024 blocks each)
                constexpr size_t kLists = 64, kHeader = kLists * 32;

                // coarse boxes of 2 x 2 x 2 blocks, cpw of them per wavefront (its one coarse evaluation is the overhead when every box needs
                // the closer look: 1 / cpw): as many as leave >= 4096 wavefronts to the launch, 4 at most.  Measured (sphere / README scene,
                // us) at 512^3, 16 384 coarse boxes: cpw 1: 16.7 / 25.4, 2: 14.1 / 21.0, 4: 13.4 / 18.9, 16: 19.3 / 35.0, 32: 27.2 / 57.6 (few, long
                // wavefronts), without the coarse pass 15.1 / 23.3 at cpw 1; at 1024^3: cpw 4: 44.8 / 59.0, 8: 57.6 / 79.3, 16: 59.1 / 77.4,
                // without the coarse pass 89.3 / 119.1 at cpw 8
                const size_t ncoarse = (size_t)((Cargs.nbx + 1) / 2) * ((Cargs.nby + 1) / 2) * ((Cargs.nbz + 1) / 2);
                static const int cpw_env = [] { const char* e = getenv("SDFK_CULL_CPW"); return e ? atoi(e) : 0; }();   // (experiments)
                Cargs.cpw = cpw_env > 0 ? std::min(cpw_env, 4) : (int)std::min<size_t>(4, std::max<size_t>(1, ncoarse / 4096));   // (<= 4: the kernel's list)
                static const bool coarse_off = [] { const char* e = getenv("SDFK_CULL_COARSE"); return e && atoi(e) == 0; }();
                if (coarse_off) Cargs.cpw = -Cargs.cpw;
                const size_t cull_wgs = (ncoarse + (size_t)std::abs(Cargs.cpw) * 4 - 1) / ((size_t)std::abs(Cargs.cpw) * 4);
                Cargs.region = (unsigned)(((cull_wgs + kLists - 1) / kLists) * (size_t)std::abs(Cargs.cpw) * 32);   // (32 blocks per coarse box of a workgroup's four wavefronts)
                if (!v->cull_list) {   // (a volume's dimensions never change: neither does the size of its regions)
                    // (the blocks' sub-box masks, a byte each, behind the regions)
                    if (int r = dev_alloc((void**)&v->cull_list, kLists * (size_t)Cargs.region * sizeof(uint32_t) + kLists * (size_t)Cargs.region + 64)) return r;
                }
                if (!v->cull_header) {   // the counters: a block this lane knows to be zero, or a new one, cleared once
                    auto& clean = g.lanes[g.cur_lane].clean_cull_headers;
                    if (!clean.empty()) { v->cull_header = clean.back(); clean.pop_back(); }
                    else {
                        if (int r = dev_alloc((void**)&v->cull_header, kHeader * sizeof(uint32_t))) return r;
                        HIPCHK(hipMemsetAsync(v->cull_header, 0, kHeader * sizeof(uint32_t), g.stream));
                    }
                    v->cull_header_lane = g.cur_lane;
                } else if (!v->cull_header_clean || v->cull_header_lane != g.cur_lane) {
                    HIPCHK(hipMemsetAsync(v->cull_header, 0, kHeader * sizeof(uint32_t), g.stream));
                    v->cull_header_lane = g.cur_lane;
                }
                v->cull_header_clean = false;
                Cargs.counter = v->cull_header; Cargs.worklist = v->cull_list;
                hipFunction_t fn_cull = nullptr, fn_eval = nullptr;
                if (int r = program_fn(p, PK_CULL, &fn_cull)) return r;
                if (int r = program_fn(p, PK_EVAL_BLOCKS, &fn_eval)) return r;
                void* cparams[] = {&A, &Cargs, p->kargs()};
                {
                    ProfScope ps2("sdfk_cull_blocks");
                    HIPCHK(hipModuleLaunchKernel(fn_cull, (unsigned)cull_wgs, 1, 1, 256, 1, 1, 0, g.stream, cparams, nullptr));   // (a wavefront per cpw coarse boxes)
                }
                {
                    ProfScope ps2("sdfk_eval_blocks");
                    HIPCHK(hipModuleLaunchKernel(fn_eval, (unsigned)std::min<size_t>((nblocks + 3) / 4, 4096), 1, 1, 256, 1, 1, 0, g.stream, cparams, nullptr));
                }
            } else if (mode == 1) {
                const dim3 fg = flat_grid(plane, v->nx8());
                HIPCHK(hipModuleLaunchKernel(fn, fg.x, fg.y, fg.z, tpb, 1, 1, 0, g.stream, params, nullptr));
            }
            else
                HIPCHK(hipModuleLaunchKernel(fn, (unsigned)((v->nz + 255) / 256), (unsigned)v->ny,
                                             (unsigned)v->nx8(), tpb, 1, 1, 0, g.stream, params, nullptr));
            ps.reset();
            if (two_pass) {   // the colours: 256 consecutive voxels of the padded volume per workgroup, three contiguous KiB each
                ProfScope ps2("sdfk_sample_colors");
                const size_t total = (size_t)v->nx * plane;
                HIPCHK(hipModuleLaunchKernel(fn_colors, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
            }
        }
        phase_token_pass(0);
        if (g.sampler_only) {   // measurement mode (sdfk_profile_enable(2)): the sampling kernel alone, back to back
            v->bits_valid = false;
            return SDFK_OK;
        }
        if (!(v->elided && g_cfg.elide_volume >= 2)) {   // (the block-culling kernels write the words themselves)
            ProfScope ps("k_bits_transpose");
            hipLaunchKernelGGL(k_bits_transpose, transpose_grid(v->nz, v->ny, v->nxw()), dim3(256), 0, g.stream,
                               v->bits8, v->bits, v->nx8(), v->ny, v->nz, v->nxw(), v->pitch8());
            HIPCHK(hipGetLastError());
        }
        v->bits_iso = iso_hint;
        v->bits_valid = true;
        sampled();
        return SDFK_OK;
    }
}


extern "C" int sdfk_sample(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !v) return fail(SDFK_ERR_INVALID, "sdfk_sample: null argument");
    if (int r = require_init()) return r;
    return sample_impl(p, v, clip_to_bounds, 0.0f);
}

extern "C" int sdfk_volume_clip_to_bounds(sdfk_volume* v)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v) return fail(SDFK_ERR_INVALID, "null volume");
    if (int r = require_init()) return r;
    float d[3], m[3], outside;
    grid_constants(v, d, m, &outside);
    resolve_dependents(v);
    // The clip rule is known: the cached views of the values are patched instead of dropped.
    // Sign bits: the six faces become (outside > iso); the sampling program that produced the
    // values (if any) is remembered with clip = 1, so re-evaluated cell corners see the same faces.
    const size_t face = std::max({(size_t)v->ny * v->nz, (size_t)v->nx * v->nz, (size_t)v->nx * v->ny});
    ProfScope ps("k_clip");
    hipLaunchKernelGGL(k_clip, dim3((unsigned)((face + 255) / 256)), dim3(256), 0, g.stream, v->values, v->nx, v->ny, v->nz, v->pitch(), v->z0,
                       v->nz_global, outside);
    if (v->bits && v->bits_valid)
        hipLaunchKernelGGL(k_clip_bits, dim3((unsigned)(((size_t)v->nz * v->ny + 255) / 256)), dim3(256), 0, g.stream, v->bits, v->nx, v->ny,
                           v->nz, v->z0, v->nz_global, v->nxw(), outside > v->bits_iso ? 1 : 0);
    if (v->sampled_by) v->sampled_args.clip = 1;
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points
// ---------------------------------------------------------------------------
int eval_points_launch(const sdfk_program* p, const float* points_dev, int64_t n, float* rgbw_dev)
{
    struct { const float* points; float* rgbw; long n; } A{points_dev, rgbw_dev, (long)n};   // (= PointArgs of sample_codegen.h)
    hipFunction_t fn = nullptr;
    if (int r = program_fn(p, PK_POINTS, &fn)) return r;
    void* params[] = {&A, p->kargs()};
    ProfScope ps("sdfk_eval_points");
    HIPCHK(hipModuleLaunchKernel(fn, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    return SDFK_OK;
}

extern "C" int sdfk_eval_points_device(const sdfk_program* p, const void* points3_dev, int64_t n, void* rgbw4_dev)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || n < 0 || (n > 0 && (!points3_dev || !rgbw4_dev))) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: null / negative argument");
    if (n >= (int64_t(1) << 31) * 256) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: too many points");
    if (int r = require_init()) return r;
    if (n == 0) return SDFK_OK;
    return eval_points_launch(p, (const float*)points3_dev, n, (float*)rgbw4_dev);
}

extern "C" int sdfk_eval_points(const sdfk_program* p, const float* points3, int64_t n, float* rgbw4)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || n < 0 || (n > 0 && (!points3 || !rgbw4))) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: null / negative argument");
    if (n >= (int64_t(1) << 31) * 256) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: too many points");
    if (int r = require_init()) return r;
    if (n == 0) return SDFK_OK;
    float* pd = nullptr;
    float* od = nullptr;
    int r = dev_alloc((void**)&pd, (size_t)n * 3 * sizeof(float));
    if (!r) r = dev_alloc((void**)&od, (size_t)n * 4 * sizeof(float));
    hipError_t e = hipSuccess;
    if (!r) e = hipMemcpyAsync(pd, points3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream);
    // (a program that only assigns .W leaves X, Y, Z of the caller's elements alone, as the reference's delegates do: they travel there and back)
    if (!r && e == hipSuccess && !p->writes_color) e = hipMemcpyAsync(od, rgbw4, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, g.stream);
    if (!r && e == hipSuccess) r = eval_points_launch(p, pd, n, od);
    if (!r && e == hipSuccess) e = hipMemcpyAsync(rgbw4, od, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, g.stream);
    const hipError_t es = hipStreamSynchronize(g.stream);   // (the caller's arrays are not retained; the pool is stream-ordered)
    dev_free(pd);
    dev_free(od);
    if (r) return r;
    if (e != hipSuccess || es != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_eval_points: %s", hipGetErrorString(e != hipSuccess ? e : es));
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// RayMarcher
// ---------------------------------------------------------------------------
int raymarch_launch(const sdfk_program* p, int32_t width, int32_t height, const float cam[3], const float vpi[16],
                           float nearp, float farp, int32_t iters, float* depth_dev, float* rgb_dev)
{
    RayArgs A;
    memset(&A, 0, sizeof A);
    A.depth = depth_dev; A.rgb = rgb_dev;
    memcpy(A.cam, cam, sizeof A.cam);
    memcpy(A.m, vpi, sizeof A.m);
    A.width = width; A.height = height; A.nearp = nearp; A.farp = farp; A.iters = iters;
    void* params[] = {&A, p->kargs()};
    const size_t n = (size_t)width * height;
    ProfScope ps("sdfk_raymarch");
    hipFunction_t fn_raymarch = nullptr;
    if (int r = program_fn(p, PK_RAYMARCH, &fn_raymarch)) return r;
    HIPCHK(hipModuleLaunchKernel(fn_raymarch, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    return SDFK_OK;
}

extern "C" int sdfk_raymarch_device(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                                    const float view_projection_inverse[16], float near_plane, float far_plane,
                                    int32_t depth_iterations, void* depth_dev, void* rgb_dev)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !camera_position || !view_projection_inverse) return fail(SDFK_ERR_INVALID, "sdfk_raymarch: null argument");
    if (width < 1 || height < 1 || depth_iterations < 0 || (size_t)width * height > (size_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "sdfk_raymarch: bad image size %d x %d or iteration count %d", width, height, depth_iterations);
    if (int r = require_init()) return r;
    if (!depth_dev && !rgb_dev) return SDFK_OK;
    return raymarch_launch(p, width, height, camera_position, view_projection_inverse, near_plane, far_plane, depth_iterations,
                           (float*)depth_dev, (float*)rgb_dev);
}

extern "C" int sdfk_raymarch(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                             const float view_projection_inverse[16], float near_plane, float far_plane,
                             int32_t depth_iterations, float* depth, float* rgb)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !camera_position || !view_projection_inverse) return fail(SDFK_ERR_INVALID, "sdfk_raymarch: null argument");
    if (width < 1 || height < 1 || depth_iterations < 0 || (size_t)width * height > (size_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "sdfk_raymarch: bad image size %d x %d or iteration count %d", width, height, depth_iterations);
    if (int r = require_init()) return r;
    if (!depth && !rgb) return SDFK_OK;
    const size_t n = (size_t)width * height;
    float* d = nullptr;
    float* c = nullptr;
    int r = SDFK_OK;
    if (depth) r = dev_alloc((void**)&d, n * sizeof(float));
    if (!r && rgb) r = dev_alloc((void**)&c, n * 3 * sizeof(float));
    if (!r) r = raymarch_launch(p, width, height, camera_position, view_projection_inverse, near_plane, far_plane, depth_iterations, d, c);
    if (!r && depth && hipMemcpyAsync(depth, d, n * sizeof(float), hipMemcpyDeviceToHost, g.stream) != hipSuccess) r = fail(SDFK_ERR_HIP, "copy of the depth image failed");
    if (!r && rgb && hipMemcpyAsync(rgb, c, n * 3 * sizeof(float), hipMemcpyDeviceToHost, g.stream) != hipSuccess) r = fail(SDFK_ERR_HIP, "copy of the colour image failed");
    if (hipStreamSynchronize(g.stream) != hipSuccess && !r) r = fail(SDFK_ERR_HIP, "hipStreamSynchronize failed");
    dev_free(d);
    dev_free(c);
    return r;
}

