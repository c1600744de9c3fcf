// rtx_trace.hip -- batched IntersectableScene::intersects (src/scene/mod.rs:153,181-201)
// for caller-supplied rays: one ray per lane, grid-stride over a device-sized grid.
#include <algorithm>
#include <mutex>

#include "rtx_coop.h"

// Scratch of the entry point that allocates nothing (rtx_intersect_batch_device_noalloc): the overflow
// area of the per-lane traversal stacks, [entry][thread slot] like DevScene::spill, for a grid of at
// most RTX_NOALLOC_BLOCKS workgroups and trees that need at most RTX_NOALLOC_ENTRIES entries beyond
// the LDS part of a stack. 48 * 512 * 256 * 4 B = 24 MiB per GPU, held until the process ends.
// In host code the array's name is only a stand-in: the device address comes from
// hipGetSymbolAddress, once per device, and is what the kernel receives.
#define RTX_NOALLOC_BLOCKS 512
#define RTX_NOALLOC_ENTRIES 48
#define RTX_NOALLOC_DEVICES 64
static __device__ __align__(256) int scratch[(size_t)RTX_NOALLOC_ENTRIES * RTX_NOALLOC_BLOCKS * RTX_BLOCK];
static int* g_scratch_ptr[RTX_NOALLOC_DEVICES];   // per device: each GPU holds its own copy of the array at its own address
static std::mutex g_scratch_mutex;

namespace {

template <bool SPH, bool ALWAYS = false>
__global__ __launch_bounds__(RTX_BLOCK) void k_intersect(DevScene s, uint64_t n, const float* __restrict__ ori3,
                                                         const float* __restrict__ dir3, int32_t* __restrict__ prim,
                                                         float* __restrict__ tout, int exhaustive) {
    __shared__ int lds[RTX_STACK_LDS * RTX_BLOCK];
    __shared__ CoopLds coop;
    constexpr uint32_t kTop = top_nodes(4);   // this kernel takes the compiler's register budget: four workgroups fit in any case
    __shared__ uint4 top_lds[kTop * 4];
    TravStack st;
    st.set(lds, s);
    TopCache tc;
    tc.load(top_lds, s, kTop);
    // block-uniform trip count: the BVH query is cooperative inside a wavefront
    for (uint64_t base = (uint64_t)blockIdx.x * RTX_BLOCK; base < n; base += (uint64_t)gridDim.x * RTX_BLOCK) {
        const uint64_t i = base + threadIdx.x;
        const bool live = i < n;
        V3 o = v3(0.0f, 0.0f, 0.0f), d = o;
        if (live) {
            o = v3(ori3[i * 3], ori3[i * 3 + 1], ori3[i * 3 + 2]);
            d = v3(dir3[i * 3], dir3[i * 3 + 1], dir3[i * 3 + 2]);
        }
        float t = 0.0f;
        int id = -1;
        if (exhaustive || !s.uses_bvh) {
            if (live) trace_exhaustive(s, o, d, t, id);
        } else {
            coop_trace<false, SPH, ALWAYS>(s, tc, live, o, d, 0.0f, st, coop, s.tune[0], t, id);
        }
        if (live) {
            prim[i] = id;
            if (tout) tout[i] = (id >= 0) ? t : 0.0f;
        }
    }
}

}  // namespace

int rtx_launch_intersect(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3, int32_t* d_prim,
                         float* d_t, int mode) {
    if (n == 0) return RTX_OK;
    const DevScene& s = ctx->ds;
    const int exhaustive = (mode == 1) ? 1 : 0;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + RTX_BLOCK - 1) / RTX_BLOCK, s.spill_stride / RTX_BLOCK);
    if (s.n_always) hipLaunchKernelGGL((k_intersect<true, true>), dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    else if (s.n_spheres) hipLaunchKernelGGL(k_intersect<true>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    else hipLaunchKernelGGL(k_intersect<false>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

int rtx_noalloc_init_device(int device, std::string* err) {
    if (device < 0 || device >= RTX_NOALLOC_DEVICES) return RTX_OK;   // such a device simply has no no-allocation entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RTX_OK;
    int* p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) {
        if (err) *err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        return RTX_ERR_HIP;
    }
    g_scratch_ptr[device] = p;
    return RTX_OK;
}

// The same kernel as rtx_launch_intersect with the stack overflow area in the static array: no
// allocation, no free, no synchronisation. Nothing in the area has to start at zero (a stack entry
// is written before it is read), so consecutive calls need no clearing; calls that overlap on one
// GPU (two contexts, two streams) would share thread slots and race.
int rtx_launch_intersect_noalloc(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3, int32_t* d_prim,
                                 float* d_t) {
    if (n > RTX_NOALLOC_MAX_RAYS)
        return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "rtx_intersect_batch_device_noalloc: more than RTX_NOALLOC_MAX_RAYS (2^26) rays in one call");
    if (ctx->trav_spill_need > RTX_NOALLOC_ENTRIES)
        return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "rtx_intersect_batch_device_noalloc: the tree is deeper than the static stack area allows (" +
                                                      std::to_string(ctx->trav_spill_need) + " > " + std::to_string(RTX_NOALLOC_ENTRIES) + " overflow entries)");
    int* area = nullptr;
    if (ctx->device >= 0 && ctx->device < RTX_NOALLOC_DEVICES) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        area = g_scratch_ptr[ctx->device];
    }
    if (!area) return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "rtx_intersect_batch_device_noalloc: no static scratch on this device");
    if (n == 0) return RTX_OK;
    DevScene s = ctx->ds;
    s.spill = area;
    s.spill_stride = RTX_NOALLOC_BLOCKS * RTX_BLOCK;
    s.spill_entries = RTX_NOALLOC_ENTRIES;
    s.stamps = nullptr;   // the stamp buffer is sized for the context's own grids
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + RTX_BLOCK - 1) / RTX_BLOCK, RTX_NOALLOC_BLOCKS);
    if (s.n_always) hipLaunchKernelGGL((k_intersect<true, true>), dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, 0);
    else if (s.n_spheres) hipLaunchKernelGGL(k_intersect<true>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, 0);
    else hipLaunchKernelGGL(k_intersect<false>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, 0);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}
