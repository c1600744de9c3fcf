// rtx_trace.hip -- batched IntersectableScene::intersects (src/scene/mod.rs:153,181-201)
// for caller-supplied rays: one ray per lane, grid-stride over a device-sized grid.
#include <algorithm>

#include "rtx_coop.h"

namespace {

__global__ __launch_bounds__(RTX_BLOCK) void k_intersect(DevScene s, uint64_t n, const float* __restrict__ ori3,
                                                         const float* __restrict__ dir3, int32_t* __restrict__ prim,
                                                         float* __restrict__ tout, int exhaustive) {
    __shared__ int lds[RTX_STACK_LDS * RTX_BLOCK];
    __shared__ CoopLds coop;
    TravStack st;
    st.set(lds, s);
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
            coop_trace<false>(s, live, o, d, 0.0f, st, coop, s.tune[0], t, id);
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
    hipLaunchKernelGGL(k_intersect, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}
