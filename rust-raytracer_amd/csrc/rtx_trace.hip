// rtx_trace.hip -- batched IntersectableScene::intersects (src/scene/mod.rs:153,181-201)
// for caller-supplied rays: one ray per lane, grid-stride over the batch.
#include <algorithm>

#include "rtx_traverse.h"

namespace {

template <int STACK>
__global__ __launch_bounds__(RTX_BLOCK) void k_intersect(DevScene s, uint64_t n, const float* __restrict__ ori3,
                                                         const float* __restrict__ dir3, int32_t* __restrict__ prim,
                                                         float* __restrict__ tout, int exhaustive) {
    __shared__ int lds[STACK * RTX_BLOCK];
    LaneStack<STACK> st;
    st.set(lds);
    for (uint64_t i = (uint64_t)blockIdx.x * RTX_BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RTX_BLOCK) {
        const V3 o = v3(ori3[i * 3], ori3[i * 3 + 1], ori3[i * 3 + 2]);
        const V3 d = v3(dir3[i * 3], dir3[i * 3 + 1], dir3[i * 3 + 2]);
        float t;
        int id;
        if (exhaustive || !s.uses_bvh) trace_exhaustive(s, o, d, t, id);
        else trace_bvh_closest<STACK>(s, o, d, st, t, id);
        prim[i] = id;
        if (tout) tout[i] = (id >= 0) ? t : 0.0f;
    }
}

}  // namespace

int rtx_stack_for(uint32_t depth_needed) { return depth_needed <= 31 ? 32 : (depth_needed <= 63 ? 64 : 128); }

int rtx_launch_intersect(rtx_context* ctx, uint64_t n, const float* d_ori3, const float* d_dir3, int32_t* d_prim,
                         float* d_t, int mode) {
    if (n == 0) return RTX_OK;
    const DevScene& s = ctx->ds;
    const int exhaustive = (mode == 1) ? 1 : 0;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + RTX_BLOCK - 1) / RTX_BLOCK, 1u << 20);
    const int stack = rtx_stack_for(ctx->build_stats.max_depth + 1);
    if (exhaustive || !s.uses_bvh || stack == 32)
        hipLaunchKernelGGL(k_intersect<32>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    else if (stack == 64)
        hipLaunchKernelGGL(k_intersect<64>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    else
        hipLaunchKernelGGL(k_intersect<128>, dim3(grid), dim3(RTX_BLOCK), 0, ctx->stream, s, n, d_ori3, d_dir3, d_prim, d_t, exhaustive);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}
