// lib_jit.hip -- SDF programs: source generation (sample_codegen.h), hiprtc, the on-disk code-object cache, kernel sets per program structure.
#include "lib_internal.h"

#include "sample_codegen.h"

// ---------------------------------------------------------------------------
// programs (JIT, counterpart of SdfExprCompiler.Compile, SdfExpr.cs:225-273)
// ---------------------------------------------------------------------------
// ---- on-disk cache of compiled code objects ------------------------------------------------
// hiprtc takes 0.3-1 s per program, once per PROCESS without this: the reference's counterpart
// (SdfExprCompiler.Compile, SdfExpr.cs:234-238) is also a JIT, but what an `sdf.ToMesh()` user sees
// is the first-call latency.  A compiled code object is stored under
//   $SDFK_CACHE_DIR | $XDG_CACHE_HOME/sdfkit_hip | $HOME/.cache/sdfkit_hip | /tmp/sdfkit_hip-<uid>
// as <hash of (flags, hiprtc version, source)>.co = { magic, lengths, the key text itself, code }: a
// hit compares the whole key text, so a hash collision cannot return foreign code.  Files appear
// atomically (write to a temporary, rename).  SDFK_NO_CACHE=1 switches it off.

struct JitStats { std::atomic<int64_t> compiled{0}, cache_hits{0}; std::atomic<int64_t> compile_us{0}; } g_jit;   // (process-wide: hiprtc may run for two devices at once)

uint64_t fnv1a64(const std::string& s, uint64_t h)
{
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    return h;
}

// The directory must be OURS: created here with mode 0700, or an existing directory owned by this user that nobody else
// can write to -- another local user who pre-creates /tmp/sdfkit_hip-<uid> (the fallback when HOME is unset) could plant
// code objects otherwise.  Anything else: no cache.
std::string cache_dir()
{
    if (!g_cfg.code_cache) return std::string();
    std::string d;
    if (g_cfg.cache_dir_set) d = g_cfg.cache_dir;
    else if (!g_cfg.env_cache_dir.empty()) d = g_cfg.env_cache_dir;
    else if (!g_cfg.env_xdg.empty()) d = g_cfg.env_xdg + "/sdfkit_hip";
    else if (!g_cfg.env_home.empty()) d = g_cfg.env_home + "/.cache/sdfkit_hip";
    else d = "/tmp/sdfkit_hip-" + std::to_string((long)getuid());
    if (d.empty()) return d;
    // mkdir -p
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') {
            const std::string sub = d.substr(0, i);
            if (mkdir(sub.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
        }
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)))
        return std::string();
    return d;
}

constexpr uint64_t kCacheMagic = 0x31304f434b464453ull;   // "SDFKCO01"

bool cache_load(const std::string& path, const std::string& key, std::vector<char>& code)
{
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    FILE* f = fdopen(fd, "rb");
    if (!f) { close(fd); return false; }
    uint64_t hdr[3] = {0, 0, 0};
    bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == kCacheMagic && hdr[1] == key.size() && hdr[2] > 0 && hdr[2] < (1ull << 31);
    if (ok) {
        std::string k(key.size(), '\0');
        ok = fread(&k[0], 1, k.size(), f) == k.size() && k == key;
    }
    if (ok) {
        code.resize(hdr[2]);
        ok = fread(code.data(), 1, code.size(), f) == code.size() && fgetc(f) == EOF;
    }
    fclose(f);
    if (!ok) code.clear();
    return ok;
}

void cache_store(const std::string& path, const std::string& key, const std::vector<char>& code)
{
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); (void)remove(tmp.c_str()); return; }
    const uint64_t hdr[3] = {kCacheMagic, key.size(), code.size()};
    bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(key.data(), 1, key.size(), f) == key.size() &&
              fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}


// hiprtc (or the on-disk cache) for the kernels `mask` of a generated source
int compile_source(const std::string& src, unsigned mask, std::vector<char>& code, bool use_cache, bool* from_cache,
                          bool refresh)
{
    if (from_cache) *from_cache = false;
    const std::string dr = "-DSDFK_SAMPLE_RPW=" + std::to_string(kSampleRpw);
    const std::string dk = "-DSDFK_KERNELS=" + std::to_string(mask);
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-DSDFK_SAMPLE_NT=1", dr.c_str(), dk.c_str()};
    std::vector<std::string> extra;   // SDFK_JIT_FLAGS at start-up: extra hiprtc options, space separated (experiments)
    {
        std::string t;
        for (const char* q = g_cfg.jit_flags.c_str();; q++) {
            if (*q == ' ' || *q == 0) { if (!t.empty()) extra.push_back(t); t.clear(); if (!*q) break; }
            else t += *q;
        }
        for (auto& e : extra) opts.push_back(e.c_str());
    }
    // cache key: everything the code object depends on
    std::string key, path;
    if (use_cache) {
        int vmaj = 0, vmin = 0;
        (void)hiprtcVersion(&vmaj, &vmin);
        key = "sdfkit_hip abi " + std::to_string(SDFK_ABI_VERSION) + " hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin) + " opts";
        for (const char* o : opts) { key += ' '; key += o; }
        key += '\n';
        key += src;
        const std::string dir = cache_dir();
        if (!dir.empty()) {
            char name[64];
            snprintf(name, sizeof name, "/%016llx%016llx.co", (unsigned long long)fnv1a64(key, 0xcbf29ce484222325ull),
                     (unsigned long long)fnv1a64(key, 0x84222325cbf29ce4ull));
            path = dir + name;
            if (!refresh && cache_load(path, key, code)) {
                g_jit.cache_hits++;
                if (from_cache) *from_cache = true;
                return SDFK_OK;
            }
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "sdfk_sample.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(SDFK_ERR_COMPILE, "hiprtcCreateProgram failed");
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(SDFK_ERR_COMPILE, "hiprtc: %s\n%s", hiprtcGetErrorString(rc), log.c_str());
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    g_jit.compiled++;
    g_jit.compile_us += (int64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (!path.empty()) cache_store(path, key, code);
    return SDFK_OK;
}

extern "C" int sdfk_jit_stats(int64_t* n_compiled, int64_t* n_cache_hits, double* compile_ms_total)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (n_compiled) *n_compiled = g_jit.compiled;
    if (n_cache_hits) *n_cache_hits = g_jit.cache_hits;
    if (compile_ms_total) *compile_ms_total = (double)g_jit.compile_us.load() * 1e-3;
    return SDFK_OK;
}


// structures with loaded modules, keyed by the generated source (the whole text is the key: no collisions)

void code_unload(ProgCode* c)
{
    if (!c->modules.empty() && g.inited) sync_all_lanes();   // kernels of these modules may still be queued
    for (hipModule_t m : c->modules) (void)hipModuleUnload(m);
    delete c;
}

ProgCode* code_acquire(std::string&& src)
{
    ProgCode*& slot = g_codes[src];
    if (!slot) {
        slot = new ProgCode();
        slot->source = std::move(src);
    }
    slot->refs++;
    slot->last_use = ++g_code_clock;
    return slot;
}

// structures no program uses at the moment stay loaded, up to SDFK_OPT_IDLE_PROGRAMS of them: the one used longest ago goes first
void codes_trim()
{
    for (;;) {
        size_t idle = 0;
        ProgCode* oldest = nullptr;
        for (auto& kv : g_codes)
            if (kv.second->refs == 0) {
                idle++;
                if (!oldest || kv.second->last_use < oldest->last_use) oldest = kv.second;
            }
        if (idle <= (size_t)g_cfg.idle_programs || !oldest) return;
        g_codes.erase(oldest->source);
        code_unload(oldest);
    }
}

void codes_drop_idle()
{
    for (auto it = g_codes.begin(); it != g_codes.end();) {
        if (it->second->refs == 0) { ProgCode* c = it->second; it = g_codes.erase(it); code_unload(c); }
        else ++it;
    }
}

void code_release(ProgCode* c)
{
    if (!c || --c->refs > 0) return;
    c->last_use = ++g_code_clock;
    codes_trim();
}


int generate_source(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color, std::string& src,
                           std::vector<float>* params)
{
    std::string err;
    if (!generate_sample_source(ops, n_ops, out_rgbw, writes_color, src, err, params))
        return fail(SDFK_ERR_INVALID, "SDF program: %s", err.c_str());
    if (!g_cfg.dump_source.empty()) {   // debugging aid (SDFK_DUMP_SOURCE at start-up): the generated HIP source of the last program
        if (FILE* f = fopen(g_cfg.dump_source.c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    return SDFK_OK;
}

extern "C" int sdfk_program_check(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color)
{
    if (!ops || !out_rgbw || n_ops <= 0) return fail(SDFK_ERR_INVALID, "sdfk_program_check: null/empty argument");
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    std::string src;
    std::vector<char> code;
    config_from_env();   // (first: SDFK_DUMP_SOURCE is honoured by generate_source)
    if (int r = generate_source(ops, n_ops, out_rgbw, writes_color, src)) return r;
    return compile_source(src, (1u << PK_COUNT) - 1u, code, false);   // every kernel, a real compile: this IS the check
}

extern "C" int sdfk_program_create(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4],
                                   int32_t writes_color, sdfk_program** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out || !ops || !out_rgbw || n_ops <= 0) return fail(SDFK_ERR_INVALID, "sdfk_program_create: null/empty argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    sdfk_program* p = new sdfk_program();
    std::string src;
    if (int r = generate_source(ops, n_ops, out_rgbw, writes_color, src, &p->params)) { delete p; return r; }   // validates the op list
    p->writes_color = writes_color;
    p->n_ops = n_ops;
    p->code = code_acquire(std::move(src));
    *out = p;
    return SDFK_OK;
}


// The device function of kernel `k` of a program, compiled and loaded on first use.  A sampler instantiation brings
// sdfk_corners_eval along (same module): marching cubes on the volume it sampled re-evaluates cell corners with it.
int program_fn(const sdfk_program* cp, int k, hipFunction_t* fn)
{
    ProgCode* p = cp->code;
    p->last_use = ++g_code_clock;
    if (!p->fn[k]) {
        static const char* const names[PK_COUNT] = {"sdfk_sample_bits", "sdfk_sample_bits_flat", "sdfk_sample_signs", "sdfk_sample_bits_clip",
                                                    "sdfk_sample_bits_clip_flat", "sdfk_vertex_colors", "sdfk_corners_eval", "sdfk_raymarch",
                                                    "sdfk_sample_signs_flat", "sdfk_cull_blocks", "sdfk_eval_blocks", "sdfk_eval_points",
                                                    "sdfk_sample_bits_nc", "sdfk_sample_bits_nc_flat", "sdfk_sample_bits_nc_clip", "sdfk_sample_bits_nc_clip_flat",
                                                    "sdfk_sample_colors"};
        unsigned mask = 1u << k;
        if (k >= PK_BITS_NC && k <= PK_BITS_NC_CLIP_FLAT && !p->fn[PK_COLORS]) mask |= 1u << PK_COLORS;   // (the second pass: same module)
        if (k == PK_CULL || k == PK_EVAL_BLOCKS) mask |= (1u << PK_CULL) | (1u << PK_EVAL_BLOCKS);   // (a pair)
        if (pk_is_sampler(k) && !p->fn[PK_CORNERS]) {   // (and, for a program that writes colours, sdfk_vertex_colors)
            mask |= 1u << PK_CORNERS;
            if (cp->writes_color && !p->fn[PK_VCOLORS]) mask |= 1u << PK_VCOLORS;
        }
        std::vector<char> code;
        bool cached = false;
        if (int r = compile_source(p->source, mask, code, true, &cached)) return r;
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, code.data());
        if (e != hipSuccess && cached) {   // a damaged cache entry: compile again and replace it
            if (int r = compile_source(p->source, mask, code, true, nullptr, true)) return r;
            e = hipModuleLoadData(&mod, code.data());
        }
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "loading JIT module: %s", hipGetErrorString(e));
        p->modules.push_back(mod);
        for (int q = 0; q < PK_COUNT; q++)
            if ((mask >> q) & 1u) {
                e = hipModuleGetFunction(&p->fn[q], mod, names[q]);
                if (e != hipSuccess) { p->fn[q] = nullptr; return fail(SDFK_ERR_HIP, "JIT module lacks %s: %s", names[q], hipGetErrorString(e)); }
            }
    }
    *fn = p->fn[k];
    return SDFK_OK;
}


extern "C" const char* sdfk_program_source(const sdfk_program* p) { return p && p->code ? p->code->source.c_str() : ""; }

extern "C" void sdfk_program_destroy(sdfk_program* p)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!p) return;
    // captured jobs of this program can never be asked for again (the key holds the handle): the free ones go now -- and
    // with them their references, volumes and modules --, a borrowed one when its mesh handle is freed (graph_job_retire)
    p->orphaned = true;
    graph_jobs_forget_program(p);
    program_release(p);
}


void program_release(sdfk_program* p)
{
    if (!p || --p->refs > 0) return;
    code_release(p->code);   // (the modules stay loaded for the next program of this structure)
    delete p;
}

