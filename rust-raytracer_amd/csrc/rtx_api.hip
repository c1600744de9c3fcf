// rtx_api.hip -- the extern "C" entry points declared in include/rtx.h.
#include <chrono>
#include <cmath>
#include <cstring>

#include "rtx_internal.h"

namespace {
thread_local std::string g_create_error;

// Copies `count` elements into a fresh allocation owned by `store`; kind: where `src` lives.
template <typename T>
int upload(rtx_context* ctx, DevStore& store, const T* src, size_t count, T** out, hipMemcpyKind kind = hipMemcpyHostToDevice) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, std::string("hipMalloc(scene): ") + hipGetErrorString(e));
    store.ptrs.push_back(p);
    if (count && src) {
        e = hipMemcpyAsync(p, src, count * sizeof(T), kind, ctx->stream);
        if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_HIP, std::string("hipMemcpy(scene): ") + hipGetErrorString(e));
    }
    *out = (T*)p;
    return RTX_OK;
}

void free_list(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

// Staging for the entry points that take host buffers: carved out of one buffer of the context that
// only grows, so repeated calls of the same size allocate nothing.
struct Staging {
    rtx_context* ctx;
    size_t used = 0;
    bool ok = true;
    size_t sizes[4] = {0, 0, 0, 0};
    int n = 0;
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
    void want(size_t bytes) { sizes[n++] = pad(std::max<size_t>(bytes, 1)); }
    int reserve() {
        size_t total = 0;
        for (int k = 0; k < n; ++k) total += sizes[k];
        if (ctx->io_size < total) {
            if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
            if (ctx->io_buf) (void)hipFree(ctx->io_buf);
            ctx->io_buf = nullptr;
            ctx->io_size = 0;
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, total);
            if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, std::string("hipMalloc(staging): ") + hipGetErrorString(e));
            ctx->io_buf = p;
            ctx->io_size = total;
        }
        return RTX_OK;
    }
    template <typename T>
    T* get(int k) {
        size_t off = 0;
        for (int j = 0; j < k; ++j) off += sizes[j];
        return (T*)((char*)ctx->io_buf + off);
    }
};

int check_ready(rtx_context* ctx) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->has_scene) return rtx_fail(ctx, RTX_ERR_NO_SCENE, "no scene has been set (rtx_scene_set)");
    if (!ctx->built) return rtx_fail(ctx, RTX_ERR_NOT_BUILT, "scene has not been built (rtx_build)");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return RTX_OK;
}
}  // namespace

int rtx_fail(rtx_context* ctx, int status, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return status;
}

extern "C" {

int rtx_abi_version(void) { return (int)RTX_ABI_VERSION; }

#ifndef RTX_SOURCE_HASH
#define RTX_SOURCE_HASH "unknown"
#endif
const char* rtx_source_hash(void) { return RTX_SOURCE_HASH; }

const char* rtx_last_error(const rtx_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rtx_create(int device, rtx_context** out_ctx) {
    if (!out_ctx) return rtx_fail(nullptr, RTX_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return rtx_fail(nullptr, RTX_ERR_NO_DEVICE, "no HIP device is visible");
    if (device < 0 || device >= count) return rtx_fail(nullptr, RTX_ERR_NO_DEVICE, "device index out of range");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return rtx_fail(nullptr, RTX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    rtx_context* ctx = new (std::nothrow) rtx_context();
    if (!ctx) return rtx_fail(nullptr, RTX_ERR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cus = (uint32_t)prop.multiProcessorCount;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return rtx_fail(nullptr, RTX_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    {   // address of this device's copy of the static scratch array (rtx_intersect_batch_device_noalloc)
        std::string err;
        const int rc = rtx_noalloc_init_device(device, &err);
        if (rc != RTX_OK) {
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return rtx_fail(nullptr, rc, err);
        }
    }
    *out_ctx = ctx;
    return RTX_OK;
}

void rtx_destroy(rtx_context* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)rtx_comm_destroy(ctx);
    rtx_clear_graphs(ctx);
    ctx->scene_store.reset();
    ctx->build_store.reset();
    free_list(ctx->trav_allocs);
    free_list(ctx->work_allocs);
    if (ctx->io_buf) (void)hipFree(ctx->io_buf);
    if (ctx->build_arena) (void)hipFree(ctx->build_arena);
    for (auto& e : ctx->order_bufs) (void)hipFree(e.second);
    ctx->order_bufs.clear();
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    for (auto& slot : ctx->async_slots)
        for (hipEvent_t e : slot.events) (void)hipEventDestroy(e);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->cam_dev) (void)hipFree(ctx->cam_dev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void* rtx_stream(rtx_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

uint64_t rtx_debug_stamps(rtx_context* ctx, int kind, uint64_t* out, uint64_t capacity) {
    if (!ctx || !ctx->built || !ctx->ds.stamps || kind < 0 || kind > 2) return 0;
    // kind 2 (diagnostic builds): per-unit records of the last closest-hit launch, 2^19 x 2 u64 read as 2^18 x 4
    const uint64_t waves = ctx->ds.spill_stride / 64, count = (kind == 2) ? (1ull << 18) : waves;
    const uint64_t n = std::min<uint64_t>(count, capacity);
    if (out && n) {
        if (hipSetDevice(ctx->device) != hipSuccess) return 0;
        if (hipMemcpy(out, ctx->ds.stamps + (size_t)std::min(kind, 2) * waves * 4, n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    }
    return count;
}

int rtx_set_option(rtx_context* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return RTX_ERR_INVALID_ARG;
    const std::string k(key);
    if (k != "kernel_timing") rtx_clear_graphs(ctx);   // recorded frames depend on the options (timed frames are never recorded)
    if (k == "force_mode") {
        if (value < 0 || value > 2) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "force_mode must be 0, 1 or 2");
        ctx->opt_force_mode = value;
        ctx->built = false;   // takes effect at the next rtx_build
    } else if (k.size() == 5 && k.compare(0, 4, "tune") == 0 && k[4] >= '0' && k[4] <= '3') {
        if (k[4] == '3' && value != 1) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "tune3 (primitives per leaf) is fixed at 1");
        ctx->opt_tune[k[4] - '0'] = (int32_t)value;
        for (int i = 0; i < 4; ++i) ctx->ds.tune[i] = ctx->opt_tune[i];
    } else if (k == "graphs") {
        ctx->opt_graphs = value ? 1 : 0;
    } else if (k == "kernel_timing") {
        ctx->opt_kernel_timing = value ? 1 : 0;
    } else if (k == "level_hints") {
        ctx->opt_level_hints = value ? 1 : 0;
    } else if (k == "cost_order") {
        ctx->opt_cost_order = value ? 1 : 0;
    } else if (k == "verify_build") {
        ctx->opt_verify_build = value ? 1 : 0;
    } else if (k == "stamps") {
        ctx->opt_stamps = value ? 1 : 0;
        ctx->built = false;   // the stamp buffer is allocated by rtx_build
    } else if (k == "batch_pixels") {
        if (value < 64) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "batch_pixels must be at least 64");
        ctx->opt_batch_pixels = value;
    } else if (k == "pool_factor") {
        if (value < 1 || value > 64) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "pool_factor must be in 1..64");
        ctx->opt_pool_factor = value;
        ctx->work_batch = 0;  // workspace is re-sized at the next frame
    } else {
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "unknown option: " + k);
    }
    return RTX_OK;
}

// Shared by rtx_scene_set (host arrays) and rtx_scene_set_device (device arrays; materials and lights on the host).
// Everything is validated and uploaded into a new store first; the context's previous scene is
// replaced only when that has succeeded, so a rejected descriptor leaves it usable.
static int scene_set_impl(rtx_context* ctx, const rtx_scene_desc* d, bool device_arrays) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!d) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "scene descriptor is NULL");
    const uint64_t n64 = (uint64_t)d->n_tris + d->n_spheres;
    if (n64 >= (1ull << 28)) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "more than 2^28 primitives");
    const uint32_t n = (uint32_t)n64;
    if ((d->n_tris && (!d->tri_pos || !d->tri_mat)) || (d->n_spheres && (!d->sph_center_radius || !d->sph_mat)) ||
        (d->n_materials && !d->materials) || (d->n_lights && !d->lights))
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "scene descriptor has a NULL array with a non-zero count");
    for (uint32_t i = 0; i < d->n_lights; ++i)
        if (d->lights[i].type > RTX_LIGHT_DIRECTIONAL) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "unknown light type");

    // host arrays: prim_ref (reference primitive id -> (kind, index)) is built and checked here;
    // device arrays: a kernel does the same after the copy (rtx_build_prim_ref)
    std::vector<uint32_t> ref;
    bool need_normals = false;
    if (!device_arrays) {
        ref.assign(n, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < d->n_tris; ++i) {
            const uint32_t id = d->tri_prim_index ? d->tri_prim_index[i] : i;
            if (id >= n || ref[id] != 0xFFFFFFFFu) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "prim_index arrays are not a permutation of 0..n-1");
            ref[id] = i;
        }
        for (uint32_t i = 0; i < d->n_spheres; ++i) {
            const uint32_t id = d->sph_prim_index ? d->sph_prim_index[i] : d->n_tris + i;
            if (id >= n || ref[id] != 0xFFFFFFFFu) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "prim_index arrays are not a permutation of 0..n-1");
            ref[id] = i | 0x80000000u;
        }
        if (d->tri_flags)
            for (uint32_t i = 0; i < d->n_tris && !need_normals; ++i) need_normals = (d->tri_flags[i] & RTX_TRI_VERTEX_NORMAL) != 0;
        if (need_normals && !d->tri_normals) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "a triangle has RTX_TRI_VERTEX_NORMAL but tri_normals is NULL");
    }
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // frames in flight still read the old scene
    const auto t0 = std::chrono::steady_clock::now();

    auto store = std::make_shared<DevStore>();
    const hipMemcpyKind kind = device_arrays ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    DevScene s{};
    s.n_prims = n;
    s.n_tris = d->n_tris;
    s.n_spheres = d->n_spheres;
    s.n_lights = d->n_lights;
    s.n_materials = d->n_materials;
    uint32_t* d_ref = nullptr;
    float *d_pos = nullptr, *d_nrm = nullptr;
    uint32_t *d_tmat = nullptr, *d_smat = nullptr;
    uint8_t* d_flags = nullptr;
    float4* d_sph = nullptr;
    rtx_material* d_mats = nullptr;
    DevLight* d_lights = nullptr;
    int rc;
#define S_TRY(x) do { rc = (x); if (rc != RTX_OK) return rc; } while (0)
    S_TRY(upload(ctx, *store, device_arrays ? (const uint32_t*)nullptr : ref.data(), n, &d_ref));
    S_TRY(upload(ctx, *store, d->tri_pos, (size_t)d->n_tris * 9, &d_pos, kind));
    S_TRY(upload(ctx, *store, d->tri_mat, (size_t)d->n_tris * 3, &d_tmat, kind));
    S_TRY(upload(ctx, *store, d->tri_flags, (size_t)d->n_tris, &d_flags, kind));
    if (!d->tri_flags) RTX_HIP(ctx, hipMemsetAsync(d_flags, 0, std::max<size_t>(d->n_tris, 1), ctx->stream));
    S_TRY(upload(ctx, *store, (const float4*)d->sph_center_radius, (size_t)d->n_spheres, &d_sph, kind));
    S_TRY(upload(ctx, *store, d->sph_mat, (size_t)d->n_spheres, &d_smat, kind));
    // Colours are clamped to [0, 1] on the way in, as every Color the reference builds is (Color::init, material.rs:17-27;
    // parser/mod.rs:111-114): the kernels then use them as stored.
    std::vector<rtx_material> mats(d->materials, d->materials + d->n_materials);
    for (rtx_material& m : mats)
        for (int k = 0; k < 3; ++k) {
            m.diffuse[k] = clamp01(m.diffuse[k]);
            m.ambient[k] = clamp01(m.ambient[k]);
            m.specular[k] = clamp01(m.specular[k]);
            m.emissive[k] = clamp01(m.emissive[k]);
        }
    S_TRY(upload(ctx, *store, mats.data(), mats.size(), &d_mats));
    if (device_arrays) {
        uint32_t *d_ti = nullptr, *d_si = nullptr, *d_status = nullptr;
        auto scratch_store = std::make_shared<DevStore>();   // index arrays are only needed to build prim_ref
        if (d->tri_prim_index) S_TRY(upload(ctx, *scratch_store, d->tri_prim_index, (size_t)d->n_tris, &d_ti, kind));
        if (d->sph_prim_index) S_TRY(upload(ctx, *scratch_store, d->sph_prim_index, (size_t)d->n_spheres, &d_si, kind));
        S_TRY(upload(ctx, *scratch_store, (const uint32_t*)nullptr, 2, &d_status));
        S_TRY(rtx_build_prim_ref(ctx, d->n_tris, d->n_spheres, d_ti, d_si, d_flags, d_ref, d_status));
        uint32_t status[2] = {0, 0};
        RTX_HIP(ctx, hipMemcpyAsync(status, d_status, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
        RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (status[0]) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "prim_index arrays are not a permutation of 0..n-1");
        need_normals = status[1] != 0;
        if (need_normals && !d->tri_normals) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "a triangle has RTX_TRI_VERTEX_NORMAL but tri_normals is NULL");
    }
    if (need_normals) S_TRY(upload(ctx, *store, d->tri_normals, (size_t)d->n_tris * 9, &d_nrm, kind));
    s.unit_light_dirs = 1;
    std::vector<DevLight> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        DevLight l{};
        l.type = d->lights[i].type;
        for (int k = 0; k < 3; ++k) {
            l.a[k] = d->lights[i].a[k];
            l.b[k] = d->lights[i].b[k];
            l.color[k] = clamp01(d->lights[i].color[k]);   // Color::init clamps (parser/mod.rs:111-114)
        }
        if (l.type == RTX_LIGHT_DIRECTIONAL && !(l.a[0] * l.a[0] + l.a[1] * l.a[1] + l.a[2] * l.a[2] <= 1.00001f)) s.unit_light_dirs = 0;   // also a NaN
        l.area_before = s.n_area_lights;   // counts the area lights seen so far
        if (l.type == RTX_LIGHT_AREA) ++s.n_area_lights;
        lights[i] = l;
    }
    S_TRY(upload(ctx, *store, lights.data(), lights.size(), &d_lights));
    s.prim_ref = d_ref;
    s.tri_pos = d_pos;
    s.tri_nrm = d_nrm;
    s.tri_mat = d_tmat;
    s.tri_flags = d_flags;
    s.sph = d_sph;
    s.sph_mat = d_smat;
    s.mats = d_mats;
    s.lights = d_lights;
    uint4* d_srec = nullptr;
    S_TRY(upload(ctx, *store, (const uint4*)nullptr, (size_t)n * 4, &d_srec));
    S_TRY(rtx_build_shade_recs(ctx, s, d_srec));
    s.shade_recs = d_srec;
#undef S_TRY
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors go out of scope below

    // the new scene is complete: it replaces the old one
    rtx_clear_graphs(ctx);
    ctx->hit_fraction = 1.0f;
    ctx->hit_fraction_known = false;
    ctx->scene_store = store;
    ctx->build_store.reset();
    free_list(ctx->trav_allocs);
    for (auto& e : ctx->order_bufs) (void)hipFree(e.second);   // unit costs of the previous scene
    ctx->order_bufs.clear();
    ctx->built = false;
    ctx->lights.assign(d->lights, d->lights + d->n_lights);
    ctx->has_secondary = false;
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rtx_material& m = mats[i];
        if (m.transparency != 0.0f) s.has_transparent = 1;
        const float sc = std::sqrt(m.specular[0] * m.specular[0] + m.specular[1] * m.specular[1] + m.specular[2] * m.specular[2]);
        if (sc > 0.0f || m.transparency > 0.0f || sc != sc || m.transparency != m.transparency) ctx->has_secondary = true;
    }
    for (int i = 0; i < 4; ++i) s.tune[i] = ctx->opt_tune[i];
    ctx->ds = s;
    ctx->camera = d->camera;
    ctx->build_stats = rtx_build_stats{};
    ctx->build_stats.upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->has_scene = true;
    return RTX_OK;
}

int rtx_scene_set(rtx_context* ctx, const rtx_scene_desc* d) { return scene_set_impl(ctx, d, false); }
int rtx_scene_set_device(rtx_context* ctx, const rtx_scene_desc* d) { return scene_set_impl(ctx, d, true); }

int rtx_clone(rtx_context* src, rtx_context** out_ctx) {
    if (!out_ctx) return rtx_fail(src, RTX_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (!src) return RTX_ERR_INVALID_ARG;
    if (!src->has_scene) return rtx_fail(src, RTX_ERR_NO_SCENE, "no scene has been set (rtx_scene_set)");
    rtx_context* ctx = nullptr;
    int rc = rtx_create(src->device, &ctx);
    if (rc != RTX_OK) return rtx_fail(src, rc, rtx_last_error(nullptr));
    ctx->camera = src->camera;
    ctx->lights = src->lights;
    ctx->ds = src->ds;
    ctx->ds.spill = nullptr;     // per context: two streams must not share stack overflow slots
    ctx->ds.stamps = nullptr;
    ctx->build_stats = src->build_stats;
    for (int k = 0; k < 3; ++k) { ctx->scene_min[k] = src->scene_min[k]; ctx->scene_max[k] = src->scene_max[k]; }
    ctx->scene_store = src->scene_store;
    ctx->build_store = src->build_store;
    ctx->has_secondary = src->has_secondary;
    ctx->has_scene = true;
    ctx->opt_force_mode = src->opt_force_mode;
    ctx->opt_batch_pixels = src->opt_batch_pixels;
    ctx->opt_pool_factor = src->opt_pool_factor;
    for (int k = 0; k < 4; ++k) ctx->opt_tune[k] = src->opt_tune[k];
    ctx->opt_stamps = src->opt_stamps;
    ctx->opt_verify_build = src->opt_verify_build;
    ctx->opt_graphs = src->opt_graphs;
    ctx->opt_kernel_timing = src->opt_kernel_timing;
    ctx->opt_cost_order = src->opt_cost_order;
    ctx->opt_level_hints = src->opt_level_hints;
    ctx->trav_spill_need = src->trav_spill_need;
    if (src->built) {
        const hipError_t se = hipStreamSynchronize(src->stream);   // the tree the clone reads is complete
        if (se != hipSuccess) {
            rtx_destroy(ctx);
            return rtx_fail(src, RTX_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(se));
        }
        rc = rtx_alloc_traversal_scratch(ctx);
        if (rc != RTX_OK) {
            rtx_fail(src, rc, ctx->error);
            rtx_destroy(ctx);
            return rc;
        }
        ctx->built = true;
    }
    *out_ctx = ctx;
    return RTX_OK;
}

int rtx_build(rtx_context* ctx, rtx_build_stats* stats) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->has_scene) return rtx_fail(ctx, RTX_ERR_NO_SCENE, "no scene has been set (rtx_scene_set)");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // frames in flight still read the old tree
    rtx_clear_graphs(ctx);
    int rc = rtx_bvh_build(ctx);
    if (rc != RTX_OK) return rc;
    if (stats) *stats = ctx->build_stats;
    return RTX_OK;
}

int rtx_intersect_batch_device(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                               int32_t* d_prim_id, float* d_t) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (n && (!d_ori3 || !d_dir3 || !d_prim_id)) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "NULL ray or output pointer");
    rc = rtx_launch_intersect(ctx, n, d_ori3, d_dir3, d_prim_id, d_t, 0);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

int rtx_intersect_batch_device_noalloc(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3,
                                       int32_t* d_prim_id, float* d_t) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (n && (!d_ori3 || !d_dir3 || !d_prim_id)) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "NULL ray or output pointer");
    return rtx_launch_intersect_noalloc(ctx, n, d_ori3, d_dir3, d_prim_id, d_t);
}

static int intersect_host(rtx_context* ctx, uint64_t n, const float* ori3, const float* dir3, int32_t* prim_id, float* t, int mode) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (n == 0) return RTX_OK;
    if (!ori3 || !dir3 || !prim_id) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "NULL ray or output pointer");
    Staging sg{ctx};
    sg.want(n * 12); sg.want(n * 12); sg.want(n * 4); sg.want(n * 4);
    rc = sg.reserve();
    if (rc != RTX_OK) return rc;
    float* d_o = sg.get<float>(0);
    float* d_d = sg.get<float>(1);
    int32_t* d_p = sg.get<int32_t>(2);
    float* d_t = sg.get<float>(3);
    RTX_HIP(ctx, hipMemcpyAsync(d_o, ori3, n * 12, hipMemcpyHostToDevice, ctx->stream));
    RTX_HIP(ctx, hipMemcpyAsync(d_d, dir3, n * 12, hipMemcpyHostToDevice, ctx->stream));
    rc = rtx_launch_intersect(ctx, n, d_o, d_d, d_p, d_t, mode);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipMemcpyAsync(prim_id, d_p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (t) RTX_HIP(ctx, hipMemcpyAsync(t, d_t, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

int rtx_intersect_batch(rtx_context* ctx, uint64_t n, const float* ori3, const float* dir3, int32_t* prim_id, float* t) {
    return intersect_host(ctx, n, ori3, dir3, prim_id, t, 0);
}
int rtx_intersect_batch_exhaustive(rtx_context* ctx, uint64_t n, const float* ori3, const float* dir3, int32_t* prim_id, float* t) {
    return intersect_host(ctx, n, ori3, dir3, prim_id, t, 1);
}

int rtx_primary_hits(rtx_context* ctx, uint32_t width, uint32_t height, int32_t* prim_id, float* t, float* ray_dir) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    const size_t np = (size_t)width * height;
    Staging sg{ctx};
    sg.want(np * 4); sg.want(np * 4); sg.want(np * 12);
    rc = sg.reserve();
    if (rc != RTX_OK) return rc;
    int32_t* d_p = prim_id ? sg.get<int32_t>(0) : nullptr;
    float* d_t = t ? sg.get<float>(1) : nullptr;
    float* d_d = ray_dir ? sg.get<float>(2) : nullptr;
    rc = rtx_primary_impl(ctx, width, height, d_p, d_t, d_d);
    if (rc != RTX_OK) return rc;
    if (prim_id) RTX_HIP(ctx, hipMemcpy(prim_id, d_p, np * 4, hipMemcpyDeviceToHost));
    if (t) RTX_HIP(ctx, hipMemcpy(t, d_t, np * 4, hipMemcpyDeviceToHost));
    if (ray_dir) RTX_HIP(ctx, hipMemcpy(ray_dir, d_d, np * 12, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_render_tiles(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                     uint32_t rank, uint32_t world, uint8_t* d_rgb_tiles, rtx_render_stats* stats) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (world == 0 || rank >= world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rank must be below world");
    if (!d_rgb_tiles) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "d_rgb_tiles is NULL");
    return rtx_render_impl(ctx, width, height, depth, samples, rank, world, d_rgb_tiles, stats, false);
}

int rtx_assemble_tiles(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t world, const uint8_t* d_slabs, uint8_t* d_rgb) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (width == 0 || height == 0 || world == 0 || !d_slabs || !d_rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "bad argument");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rtx_assemble_impl(ctx, width, height, world, d_slabs, d_rgb);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

int rtx_assemble_tiles_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t world, const uint8_t* d_slabs,
                             uint8_t* d_rgb) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (width == 0 || height == 0 || world == 0 || !d_slabs || !d_rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "bad argument");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    return rtx_assemble_impl(ctx, width, height, world, d_slabs, d_rgb);
}

int rtx_render_device(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples, uint8_t* d_rgb,
                      rtx_render_stats* stats) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (!d_rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "d_rgb is NULL");
    return rtx_render_impl(ctx, width, height, depth, samples, 0, 1, d_rgb, stats, true);
}

int rtx_render_device_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples, uint8_t* d_rgb) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (!d_rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "d_rgb is NULL");
    return rtx_render_async_impl(ctx, width, height, depth, samples, 0, 1, d_rgb, true);
}

int rtx_render_tiles_async(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples, uint32_t rank,
                           uint32_t world, uint8_t* d_rgb_tiles) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (world == 0 || rank >= world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rank must be below world");
    if (!d_rgb_tiles) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "d_rgb_tiles is NULL");
    return rtx_render_async_impl(ctx, width, height, depth, samples, rank, world, d_rgb_tiles, false);
}

static int check_frames(rtx_context* ctx, uint32_t n_frames, uint32_t width, uint32_t height, uint32_t rank, uint32_t world,
                        const uint8_t* d_out, int row_major) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (n_frames == 0 || n_frames > RTX_MAX_BATCH_FRAMES) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "n_frames must be 1 .. RTX_MAX_BATCH_FRAMES");
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (world == 0 || rank >= world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rank must be below world");
    if (row_major && world != 1) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "row-major output needs world == 1");
    if (!d_out) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "d_out is NULL");
    return RTX_OK;
}

int rtx_render_frames(rtx_context* ctx, uint32_t n_frames, const rtx_camera* cameras, uint32_t width, uint32_t height,
                      uint32_t depth, uint32_t samples, uint32_t rank, uint32_t world, uint8_t* d_out, int row_major,
                      rtx_render_stats* stats) {
    const int rc = check_frames(ctx, n_frames, width, height, rank, world, d_out, row_major);
    if (rc != RTX_OK) return rc;
    return rtx_render_impl(ctx, width, height, depth, samples, rank, world, d_out, stats, row_major != 0, -1, n_frames, cameras);
}

int rtx_render_frames_async(rtx_context* ctx, uint32_t n_frames, const rtx_camera* cameras, uint32_t width, uint32_t height,
                            uint32_t depth, uint32_t samples, uint32_t rank, uint32_t world, uint8_t* d_out, int row_major) {
    const int rc = check_frames(ctx, n_frames, width, height, rank, world, d_out, row_major);
    if (rc != RTX_OK) return rc;
    return rtx_render_async_impl(ctx, width, height, depth, samples, rank, world, d_out, row_major != 0, n_frames, cameras);
}

int rtx_assemble_frames_async(rtx_context* ctx, uint32_t n_frames, uint32_t width, uint32_t height, uint32_t world,
                              const uint8_t* d_slabs, uint8_t* d_rgb) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (n_frames == 0 || n_frames > RTX_MAX_BATCH_FRAMES || width == 0 || height == 0 || world == 0 || !d_slabs || !d_rgb)
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "bad argument");
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    return rtx_assemble_impl(ctx, width, height, world, d_slabs, d_rgb, n_frames);
}

int rtx_render_wait(rtx_context* ctx, rtx_render_stats* stats) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    return rtx_render_wait_impl(ctx, stats);
}

int rtx_render(rtx_context* ctx, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples, uint8_t* rgb,
               rtx_render_stats* stats) {
    int rc = check_ready(ctx);
    if (rc != RTX_OK) return rc;
    if (width == 0 || height == 0) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "zero image size");
    if (!rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rgb is NULL");
    const size_t img = (size_t)width * height * 3;
    Staging sg{ctx};
    sg.want(img);
    rc = sg.reserve();
    if (rc != RTX_OK) return rc;
    uint8_t* d_img = sg.get<uint8_t>(0);
    rc = rtx_render_impl(ctx, width, height, depth, samples, 0, 1, d_img, stats, true);
    if (rc != RTX_OK) return rc;
    RTX_HIP(ctx, hipMemcpyAsync(rgb, d_img, img, hipMemcpyDeviceToHost, ctx->stream));
    RTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTX_OK;
}

}  // extern "C"
