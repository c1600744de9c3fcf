// rtx_comm.hip -- the two exchange steps of a tile-sharded frame behind the C ABI, over RCCL (xGMI inside a node):
// the one-time replication of the scene from a root rank into every rank's HBM, and the per-frame gather of the
// ranks' tile slabs to the root, where the image is assembled. Rendering itself needs no collective: pixels are
// independent (src/lib.rs:262-271). One process per GPU, one context per process takes part.
// librccl is loaded on first use (dlopen), so callers that never shard a frame do not pay for it.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include <rccl/rccl.h>

#include "rtx_internal.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// nullptr (and g_rccl.error set) when the library or one of its entry points is missing.
Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        g_rccl.error = std::string("librccl.so cannot be loaded: ") + dlerror();
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) {
            ok = false;
            g_rccl.error = std::string("librccl.so lacks ") + name;
        }
        return p;
    };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(h);
        return nullptr;
    }
    g_rccl.handle = h;
    return &g_rccl;
}

#define RTX_NCCL(ctx, R, call)                                                                         \
    do {                                                                                               \
        const ncclResult_t r_ = (call);                                                                \
        if (r_ != ncclSuccess) return rtx_fail((ctx), RTX_ERR_HIP, std::string(#call) + ": " + (R)->GetErrorString(r_)); \
    } while (0)

// What the root tells the other ranks before the arrays travel.
struct SceneHeader {
    uint32_t n_tris, n_spheres, n_materials, n_lights;
    uint32_t has_normals, has_flags, has_tri_index, has_sph_index;
    rtx_camera camera;
};

struct DevBuf {   // device allocations of one call, freed on every way out
    std::vector<void*> ptrs;
    ~DevBuf() { for (void* p : ptrs) (void)hipFree(p); }
    void* get(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return p;
    }
};

}  // namespace

extern "C" {

int rtx_comm_unique_id(uint8_t* id128) {
    if (!id128) return rtx_fail(nullptr, RTX_ERR_INVALID_ARG, "id128 is NULL");
    Rccl* R = rccl();
    if (!R) return rtx_fail(nullptr, RTX_ERR_UNSUPPORTED, g_rccl.error);
    ncclUniqueId id;
    static_assert(sizeof(id) == RTX_COMM_ID_BYTES, "RTX_COMM_ID_BYTES must be the size of RCCL's unique id");
    RTX_NCCL(nullptr, R, R->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return RTX_OK;
}

int rtx_comm_init(rtx_context* ctx, uint32_t rank, uint32_t world, const uint8_t* id128) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!id128 || world == 0 || rank >= world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rank must be below world and the id non-NULL");
    if (ctx->comm) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "the context already belongs to a communicator (rtx_comm_destroy first)");
    Rccl* R = rccl();
    if (!R) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, g_rccl.error);
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    RTX_NCCL(ctx, R, R->CommInitRank(&comm, (int)world, id, (int)rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return RTX_OK;
}

int rtx_comm_destroy(rtx_context* ctx) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->comm) return RTX_OK;
    Rccl* R = rccl();
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (R) (void)R->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    if (ctx->gather_buf) (void)hipFree(ctx->gather_buf);
    ctx->gather_buf = nullptr;
    ctx->gather_size = 0;
    return RTX_OK;
}

int rtx_scene_broadcast(rtx_context* ctx, const rtx_scene_desc* scene, uint32_t root) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->comm) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rtx_comm_init has not been called on this context");
    if (root >= ctx->comm_world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "root must be below world");
    const bool is_root = ctx->comm_rank == root;
    if (is_root && !scene) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "the root rank needs a scene descriptor");
    Rccl* R = rccl();
    if (!R) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, g_rccl.error);
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    DevBuf bufs;

    // A byte range replicated from the root: device buffer on every rank, filled from `host` on the root.
    auto replicate = [&](const void* host, size_t bytes, void** out) -> int {
        *out = nullptr;
        if (bytes == 0) return RTX_OK;
        void* d = bufs.get(bytes);
        if (!d) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc(broadcast buffer) failed");
        if (is_root) RTX_HIP(ctx, hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st));
        RTX_NCCL(ctx, R, R->Broadcast(d, d, bytes, ncclUint8, (int)root, comm, st));
        *out = d;
        return RTX_OK;
    };
#define B_TRY(x) do { const int rc_ = (x); if (rc_ != RTX_OK) return rc_; } while (0)

    SceneHeader hdr{};
    if (is_root) {
        if ((scene->n_tris && (!scene->tri_pos || !scene->tri_mat)) || (scene->n_spheres && (!scene->sph_center_radius || !scene->sph_mat)) ||
            (scene->n_materials && !scene->materials) || (scene->n_lights && !scene->lights))
            hdr.n_tris = 0xFFFFFFFFu;   // every rank learns that the descriptor is unusable and fails the same way
        else
            hdr = SceneHeader{scene->n_tris, scene->n_spheres, scene->n_materials, scene->n_lights, scene->tri_normals ? 1u : 0u,
                              scene->tri_flags ? 1u : 0u, scene->tri_prim_index ? 1u : 0u, scene->sph_prim_index ? 1u : 0u, scene->camera};
    }
    void* d_hdr = nullptr;
    B_TRY(replicate(&hdr, sizeof(hdr), &d_hdr));
    RTX_HIP(ctx, hipMemcpyAsync(&hdr, d_hdr, sizeof(hdr), hipMemcpyDeviceToHost, st));
    RTX_HIP(ctx, hipStreamSynchronize(st));
    if (hdr.n_tris == 0xFFFFFFFFu) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "the root's scene descriptor has a NULL array with a non-zero count");

    const size_t nt = hdr.n_tris, ns = hdr.n_spheres;
    void *d_pos, *d_nrm, *d_tmat, *d_flags, *d_tidx, *d_sph, *d_smat, *d_sidx, *d_mats, *d_lights;
    B_TRY(replicate(is_root ? scene->tri_pos : nullptr, nt * 36, &d_pos));
    B_TRY(replicate(is_root ? scene->tri_normals : nullptr, hdr.has_normals ? nt * 36 : 0, &d_nrm));
    B_TRY(replicate(is_root ? scene->tri_mat : nullptr, nt * 12, &d_tmat));
    B_TRY(replicate(is_root ? scene->tri_flags : nullptr, hdr.has_flags ? nt : 0, &d_flags));
    B_TRY(replicate(is_root ? scene->tri_prim_index : nullptr, hdr.has_tri_index ? nt * 4 : 0, &d_tidx));
    B_TRY(replicate(is_root ? scene->sph_center_radius : nullptr, ns * 16, &d_sph));
    B_TRY(replicate(is_root ? scene->sph_mat : nullptr, ns * 4, &d_smat));
    B_TRY(replicate(is_root ? scene->sph_prim_index : nullptr, hdr.has_sph_index ? ns * 4 : 0, &d_sidx));
    B_TRY(replicate(is_root ? scene->materials : nullptr, (size_t)hdr.n_materials * sizeof(rtx_material), &d_mats));
    B_TRY(replicate(is_root ? scene->lights : nullptr, (size_t)hdr.n_lights * sizeof(rtx_light), &d_lights));
#undef B_TRY
    // materials and lights are read by the host side of rtx_scene_set_device
    std::vector<rtx_material> mats(hdr.n_materials);
    std::vector<rtx_light> lights(hdr.n_lights);
    if (hdr.n_materials) RTX_HIP(ctx, hipMemcpyAsync(mats.data(), d_mats, mats.size() * sizeof(rtx_material), hipMemcpyDeviceToHost, st));
    if (hdr.n_lights) RTX_HIP(ctx, hipMemcpyAsync(lights.data(), d_lights, lights.size() * sizeof(rtx_light), hipMemcpyDeviceToHost, st));
    RTX_HIP(ctx, hipStreamSynchronize(st));

    rtx_scene_desc d{};
    d.n_tris = hdr.n_tris;
    d.tri_pos = (const float*)d_pos;
    d.tri_normals = (const float*)d_nrm;
    d.tri_mat = (const uint32_t*)d_tmat;
    d.tri_flags = (const uint8_t*)d_flags;
    d.tri_prim_index = (const uint32_t*)d_tidx;
    d.n_spheres = hdr.n_spheres;
    d.sph_center_radius = (const float*)d_sph;
    d.sph_mat = (const uint32_t*)d_smat;
    d.sph_prim_index = (const uint32_t*)d_sidx;
    d.n_materials = hdr.n_materials;
    d.materials = mats.data();
    d.n_lights = hdr.n_lights;
    d.lights = lights.data();
    d.camera = hdr.camera;
    return rtx_scene_set_device(ctx, &d);   // copies device-to-device; the receive buffers are freed on return
}

int rtx_gather_frames(rtx_context* ctx, uint32_t n_frames, uint32_t width, uint32_t height, const uint8_t* d_rgb_tiles, uint8_t* d_rgb,
                      uint32_t root) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->comm) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rtx_comm_init has not been called on this context");
    if (n_frames == 0 || n_frames > RTX_MAX_BATCH_FRAMES || width == 0 || height == 0 || !d_rgb_tiles || root >= ctx->comm_world)
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "bad argument");
    const bool is_root = ctx->comm_rank == root;
    if (is_root && !d_rgb) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "the root rank needs an image buffer");
    Rccl* R = rccl();
    if (!R) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, g_rccl.error);
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    const uint32_t world = ctx->comm_world;
    const size_t slab = (size_t)rtx_tiles_per_rank(width, height, world) * 64 * 3 * n_frames;   // a rank's slabs of the batch
    if (is_root && ctx->gather_size < slab * world) {   // kept for the next batch
        RTX_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->gather_buf) (void)hipFree(ctx->gather_buf);
        ctx->gather_buf = nullptr;
        ctx->gather_size = 0;
        void* p = nullptr;
        if (hipMalloc(&p, slab * world) != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc(gather buffer) failed");
        ctx->gather_buf = p;
        ctx->gather_size = slab * world;
    }
    // A gather, not an all-gather: every slab crosses one link once, all into the root (7 links in parallel on an
    // 8-GPU node); the root's own slabs are a device copy.
    RTX_NCCL(ctx, R, R->GroupStart());
    if (is_root) {
        for (uint32_t r = 0; r < world; ++r) {
            uint8_t* dst = (uint8_t*)ctx->gather_buf + (size_t)r * slab;
            if (r == root) RTX_HIP(ctx, hipMemcpyAsync(dst, d_rgb_tiles, slab, hipMemcpyDeviceToDevice, st));
            else RTX_NCCL(ctx, R, R->Recv(dst, slab, ncclUint8, (int)r, comm, st));
        }
    } else {
        RTX_NCCL(ctx, R, R->Send(d_rgb_tiles, slab, ncclUint8, (int)root, comm, st));
    }
    RTX_NCCL(ctx, R, R->GroupEnd());
    if (is_root) {
        const int rc = rtx_assemble_impl(ctx, width, height, world, (const uint8_t*)ctx->gather_buf, d_rgb, n_frames);
        if (rc != RTX_OK) return rc;
    }
    RTX_HIP(ctx, hipStreamSynchronize(st));
    return RTX_OK;
}

int rtx_gather_frame(rtx_context* ctx, uint32_t width, uint32_t height, const uint8_t* d_rgb_tiles, uint8_t* d_rgb, uint32_t root) {
    return rtx_gather_frames(ctx, 1, width, height, d_rgb_tiles, d_rgb, root);
}

}  // extern "C"
