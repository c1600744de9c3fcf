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
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
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
    if (ctx->comm_status) (void)hipFree(ctx->comm_status);
    ctx->comm_status = nullptr;
    ctx->gather_agreed_root = 0xFFFFFFFFu;
    ctx->gather_agreed_size = 0;
    return RTX_OK;
}

// Error handling of the collective calls below. A rank that returns from a collective call before its peers have
// matched every RCCL operation leaves them waiting for ever, so nothing returns between the first and the last RCCL
// call of an exchange: local failures are noted (first one wins), the remaining operations are still issued, a group
// that was opened is always closed, and the error is returned afterwards. Failures that can be foreseen -- a rank that
// cannot allocate its receive buffers -- are agreed on by all ranks (an all-reduce of a status word) BEFORE the data
// moves, so that every rank fails the same way. After a call that returned an RCCL error the communicator is not
// usable any more: rtx_comm_destroy on every rank.
struct FirstError {
    int status = RTX_OK;
    std::string msg;
    void note(int st, const std::string& m) {
        if (status == RTX_OK) { status = st; msg = m; }
    }
    void hip(hipError_t e, const char* what) {
        if (e != hipSuccess) note(RTX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    void nccl(Rccl* R, ncclResult_t r, const char* what) {
        if (r != ncclSuccess) note(RTX_ERR_HIP, std::string(what) + ": " + R->GetErrorString(r));
    }
};

int rtx_scene_broadcast(rtx_context* ctx, const rtx_scene_desc* scene, uint32_t root) {
    if (!ctx) return RTX_ERR_INVALID_ARG;
    if (!ctx->comm) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "rtx_comm_init has not been called on this context");
    if (root >= ctx->comm_world) return rtx_fail(ctx, RTX_ERR_INVALID_ARG, "root must be below world");
    const bool is_root = ctx->comm_rank == root;
    Rccl* R = rccl();
    if (!R) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, g_rccl.error);
    RTX_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t st = ctx->stream;
    DevBuf bufs;
    FirstError err;

    // Step 1: the header. A root without a usable descriptor still takes part and sends a header that says so:
    // every rank then fails the same way instead of waiting for a root that has already returned.
    SceneHeader hdr{};
    if (is_root) {
        if (!scene || (scene->n_tris && (!scene->tri_pos || !scene->tri_mat)) || (scene->n_spheres && (!scene->sph_center_radius || !scene->sph_mat)) ||
            (scene->n_materials && !scene->materials) || (scene->n_lights && !scene->lights))
            hdr.n_tris = 0xFFFFFFFFu;
        else
            hdr = SceneHeader{scene->n_tris, scene->n_spheres, scene->n_materials, scene->n_lights, scene->tri_normals ? 1u : 0u,
                              scene->tri_flags ? 1u : 0u, scene->tri_prim_index ? 1u : 0u, scene->sph_prim_index ? 1u : 0u, scene->camera};
    }
    void* d_hdr = bufs.get(sizeof(hdr));
    uint32_t* d_status = (uint32_t*)bufs.get(sizeof(uint32_t));
    if (!d_hdr || !d_status)   // 16 bytes: a rank that cannot have them cannot take part in anything; its peers time out in RCCL
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc(broadcast header) failed");
    if (is_root) err.hip(hipMemcpyAsync(d_hdr, &hdr, sizeof(hdr), hipMemcpyHostToDevice, st), "hipMemcpyAsync(header)");
    err.nccl(R, R->Broadcast(d_hdr, d_hdr, sizeof(hdr), ncclUint8, (int)root, comm, st), "ncclBroadcast(header)");
    err.hip(hipMemcpyAsync(&hdr, d_hdr, sizeof(hdr), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(header)");
    err.hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (err.status != RTX_OK) return rtx_fail(ctx, err.status, err.msg);
    if (hdr.n_tris == 0xFFFFFFFFu)
        return rtx_fail(ctx, RTX_ERR_INVALID_ARG, is_root && !scene ? "the root rank needs a scene descriptor"
                                                                    : "the root's scene descriptor is missing or has a NULL array with a non-zero count");

    // Step 2: every rank allocates ALL its receive buffers, then the ranks agree on whether everybody could.
    const size_t nt = hdr.n_tris, ns = hdr.n_spheres;
    struct Part { const void* host; size_t bytes; void* dev; };
    Part parts[10] = {
        {is_root ? scene->tri_pos : nullptr, nt * 36, nullptr},
        {is_root ? scene->tri_normals : nullptr, hdr.has_normals ? nt * 36 : 0, nullptr},
        {is_root ? scene->tri_mat : nullptr, nt * 12, nullptr},
        {is_root ? scene->tri_flags : nullptr, hdr.has_flags ? nt : 0, nullptr},
        {is_root ? scene->tri_prim_index : nullptr, hdr.has_tri_index ? nt * 4 : 0, nullptr},
        {is_root ? scene->sph_center_radius : nullptr, ns * 16, nullptr},
        {is_root ? scene->sph_mat : nullptr, ns * 4, nullptr},
        {is_root ? scene->sph_prim_index : nullptr, hdr.has_sph_index ? ns * 4 : 0, nullptr},
        {is_root ? scene->materials : nullptr, (size_t)hdr.n_materials * sizeof(rtx_material), nullptr},
        {is_root ? scene->lights : nullptr, (size_t)hdr.n_lights * sizeof(rtx_light), nullptr},
    };
    uint32_t my_status = 0;
    for (Part& p : parts)
        if (p.bytes && !(p.dev = bufs.get(p.bytes))) my_status = 1;
    uint32_t all_status = 0;
    err.hip(hipMemcpyAsync(d_status, &my_status, sizeof(my_status), hipMemcpyHostToDevice, st), "hipMemcpyAsync(status)");
    err.nccl(R, R->AllReduce(d_status, d_status, 1, ncclUint32, ncclMax, comm, st), "ncclAllReduce(status)");
    err.hip(hipMemcpyAsync(&all_status, d_status, sizeof(all_status), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(status)");
    err.hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (err.status != RTX_OK) return rtx_fail(ctx, err.status, err.msg);
    if (all_status)
        return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, my_status ? "hipMalloc(broadcast buffers) failed on this rank"
                                                              : "another rank could not allocate its broadcast buffers");

    // Step 3: the arrays, as one group; nothing returns before the group is closed.
    if (is_root)
        for (const Part& p : parts)
            if (p.bytes) err.hip(hipMemcpyAsync(p.dev, p.host, p.bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync(scene array)");
    err.nccl(R, R->GroupStart(), "ncclGroupStart");
    for (const Part& p : parts)
        if (p.bytes) err.nccl(R, R->Broadcast(p.dev, p.dev, p.bytes, ncclUint8, (int)root, comm, st), "ncclBroadcast(scene array)");
    err.nccl(R, R->GroupEnd(), "ncclGroupEnd");
    // materials and lights are read by the host side of rtx_scene_set_device
    std::vector<rtx_material> mats(hdr.n_materials);
    std::vector<rtx_light> lights(hdr.n_lights);
    if (hdr.n_materials) err.hip(hipMemcpyAsync(mats.data(), parts[8].dev, mats.size() * sizeof(rtx_material), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(materials)");
    if (hdr.n_lights) err.hip(hipMemcpyAsync(lights.data(), parts[9].dev, lights.size() * sizeof(rtx_light), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(lights)");
    err.hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (err.status != RTX_OK) return rtx_fail(ctx, err.status, err.msg);

    rtx_scene_desc d{};
    d.n_tris = hdr.n_tris;
    d.tri_pos = (const float*)parts[0].dev;
    d.tri_normals = (const float*)parts[1].dev;
    d.tri_mat = (const uint32_t*)parts[2].dev;
    d.tri_flags = (const uint8_t*)parts[3].dev;
    d.tri_prim_index = (const uint32_t*)parts[4].dev;
    d.n_spheres = hdr.n_spheres;
    d.sph_center_radius = (const float*)parts[5].dev;
    d.sph_mat = (const uint32_t*)parts[6].dev;
    d.sph_prim_index = (const uint32_t*)parts[7].dev;
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
    // Everything that can fail locally happens before the first data-moving RCCL call. The root's gather buffer only
    // grows, so a steady stream of batches allocates once. When it has to be (re)allocated -- every rank can tell: the
    // ranks all see the same sequence of (root, size) -- the root first tells the senders whether it could (one 4-byte
    // broadcast), so that a root out of memory makes every rank return the same error instead of leaving the senders
    // waiting for a receiver that has returned.
    FirstError err;
    const bool renegotiate = root != ctx->gather_agreed_root || slab * world > ctx->gather_agreed_size;
    if (renegotiate) {
        uint32_t my_status = 0;
        if (is_root && ctx->gather_size < slab * world) {
            (void)hipStreamSynchronize(st);
            if (ctx->gather_buf) (void)hipFree(ctx->gather_buf);
            ctx->gather_buf = nullptr;
            ctx->gather_size = 0;
            void* p = nullptr;
            if (hipMalloc(&p, slab * world + 256) != hipSuccess) {
                (void)hipGetLastError();
                my_status = 1;
            } else {
                ctx->gather_buf = p;
                ctx->gather_size = slab * world;
            }
        }
        uint32_t root_status = my_status;
        if (world > 1) {
            if (!ctx->comm_status && hipMalloc(&ctx->comm_status, 256) != hipSuccess) {
                (void)hipGetLastError();
                ctx->comm_status = nullptr;
                // without even this word the rank cannot take part; its peers see an RCCL error or time out
                return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, "hipMalloc(status word) failed");
            }
            if (is_root) err.hip(hipMemcpyAsync(ctx->comm_status, &my_status, 4, hipMemcpyHostToDevice, st), "hipMemcpyAsync(status)");
            err.nccl(R, R->Broadcast(ctx->comm_status, ctx->comm_status, 4, ncclUint8, (int)root, comm, st), "ncclBroadcast(status)");
            err.hip(hipMemcpyAsync(&root_status, ctx->comm_status, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(status)");
            err.hip(hipStreamSynchronize(st), "hipStreamSynchronize");
            if (err.status != RTX_OK) return rtx_fail(ctx, err.status, err.msg);
        }
        if (root_status)
            return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, is_root ? "hipMalloc(gather buffer) failed" : "the root rank could not allocate its gather buffer");
        ctx->gather_agreed_root = root;
        ctx->gather_agreed_size = slab * world;
    }
    // A gather, not an all-gather: every slab crosses one link once, all into the root (7 links in parallel on an
    // 8-GPU node); the root's own slabs are a device copy (outside the group: it is not an RCCL operation).
    if (is_root)
        err.hip(hipMemcpyAsync((uint8_t*)ctx->gather_buf + (size_t)root * slab, d_rgb_tiles, slab, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync(own slab)");
    err.nccl(R, R->GroupStart(), "ncclGroupStart");
    if (is_root) {
        for (uint32_t r = 0; r < world; ++r)
            if (r != root) err.nccl(R, R->Recv((uint8_t*)ctx->gather_buf + (size_t)r * slab, slab, ncclUint8, (int)r, comm, st), "ncclRecv");
    } else {
        err.nccl(R, R->Send(d_rgb_tiles, slab, ncclUint8, (int)root, comm, st), "ncclSend");
    }
    err.nccl(R, R->GroupEnd(), "ncclGroupEnd");   // always closed, whatever happened inside
    if (is_root && err.status == RTX_OK) {
        const int rc = rtx_assemble_impl(ctx, width, height, world, (const uint8_t*)ctx->gather_buf, d_rgb, n_frames);
        if (rc != RTX_OK) err.note(rc, ctx->error);
    }
    err.hip(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (err.status != RTX_OK) return rtx_fail(ctx, err.status, err.msg);
    return RTX_OK;
}

int rtx_gather_frame(rtx_context* ctx, uint32_t width, uint32_t height, const uint8_t* d_rgb_tiles, uint8_t* d_rgb, uint32_t root) {
    return rtx_gather_frames(ctx, 1, width, height, d_rgb_tiles, d_rgb, root);
}

}  // extern "C"
