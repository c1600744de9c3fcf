// rtx_bvh.hip -- GPU build of the linear BVH that replaces the reference's
// median-split tree (Tree::init / Tree::build, src/scene/bvh.rs:82-113):
//   bounds (Shape::get_bbox: poly.rs:122-136, sphere.rs:45-52; centroid: shapes/mod.rs:40-42)
//   -> Morton keys -> LSD radix sort staged through LDS -> Karras hierarchy
//   -> bottom-up refit in one launch (arrival counters, agent-scope release / acquire) -> 64-byte traversal
//   nodes emitted level by level into a dense array (level order); a leaf is one primitive (RTX_LEAF_MAX).
// Queries on the result return what the exhaustive loop returns (see
// rtx_traverse.h); the reference BVH's own tie rule (bvh.rs:134) is not copied.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "rtx_internal.h"
#include "rtx_traverse.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSortItems = 16;                       // keys per thread
constexpr int kSortTile = kThreads * kSortItems;     // 4096 keys per workgroup
constexpr int kScanChunk = kThreads * 16;

// Order-preserving map f32 -> u32 so that integer atomics implement float min/max.
__device__ __forceinline__ uint32_t enc_f32(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
inline float dec_f32(uint32_t e) {
    const uint32_t b = e ^ ((e >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

struct SceneBounds {   // encoded with enc_f32
    uint32_t cmin[3], cmax[3];   // centroids
    uint32_t bmin[3], bmax[3];   // boxes
};

__device__ __forceinline__ void prim_box(const DevScene& s, uint32_t id, V3& mn, V3& mx) {
    const uint32_t ref = s.prim_ref[id];
    const uint32_t idx = ref & 0x7fffffffu;
    if (ref >> 31) {  // sphere.rs:45-52
        const float4 sp = s.sph[idx];
        const V3 c = v3(sp.x, sp.y, sp.z);
        mn = vmult(v3(-1.0f, -1.0f, -1.0f), sp.w) + c;
        mx = vmult(v3(1.0f, 1.0f, 1.0f), sp.w) + c;
    } else {          // poly.rs:122-136
        const float* p = s.tri_pos + (size_t)idx * 9;
        mn = v3(fminf(p[0], fminf(p[3], p[6])), fminf(p[1], fminf(p[4], p[7])), fminf(p[2], fminf(p[5], p[8])));
        mx = v3(fmaxf(p[0], fmaxf(p[3], p[6])), fmaxf(p[1], fmaxf(p[4], p[7])), fmaxf(p[2], fmaxf(p[5], p[8])));
    }
}

// A primitive the tree is not asked about: a coordinate or the radius is not a number, infinite, or beyond 1e18 in
// magnitude (squares of differences must stay finite in the boxes' arithmetic, and a scene's bounds must not be set by
// a stray vertex). Such a primitive goes to DevScene::always -- every ray tests it, exactly as the loop of
// scene/mod.rs:186-199 does, Hit(inf) of an infinite sphere included -- and its leaf is a point inside the scene.
__device__ __forceinline__ bool prim_wild(const DevScene& s, uint32_t id) {
    const uint32_t ref = s.prim_ref[id];
    const uint32_t idx = ref & 0x7fffffffu;
    bool tame = true;
    if (ref >> 31) {
        const float4 sp = s.sph[idx];
        tame = fabsf(sp.x) <= 1e18f && fabsf(sp.y) <= 1e18f && fabsf(sp.z) <= 1e18f && fabsf(sp.w) <= 1e18f;
    } else {
        const float* p = s.tri_pos + (size_t)idx * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) tame = tame && fabsf(p[k]) <= 1e18f;   // false for NaN
    }
    return !tame;
}

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(kThreads) void k_bounds(DevScene s, float4* __restrict__ bmin, float4* __restrict__ bmax,
                                                      SceneBounds* __restrict__ sb) {
    const float inf = __builtin_inff();
    V3 cmn = v3(inf, inf, inf), cmx = v3(-inf, -inf, -inf), amn = cmn, amx = cmx;
    for (uint32_t id = blockIdx.x * kThreads + threadIdx.x; id < s.n_prims; id += gridDim.x * kThreads) {
        if (prim_wild(s, id)) {   // marked in bmin.w, kept out of the scene's bounds
            bmin[id] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            bmax[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            continue;
        }
        V3 mn, mx;
        prim_box(s, id, mn, mx);
        bmin[id] = make_float4(mn.x, mn.y, mn.z, 0.0f);
        bmax[id] = make_float4(mx.x, mx.y, mx.z, 0.0f);
        const V3 c = vmult(mn, 0.5f) + vmult(mx, 0.5f);  // BoundingBox::centroid, shapes/mod.rs:40-42
        cmn = v3(fminf(cmn.x, c.x), fminf(cmn.y, c.y), fminf(cmn.z, c.z));
        cmx = v3(fmaxf(cmx.x, c.x), fmaxf(cmx.y, c.y), fmaxf(cmx.z, c.z));
        amn = v3(fminf(amn.x, mn.x), fminf(amn.y, mn.y), fminf(amn.z, mn.z));
        amx = v3(fmaxf(amx.x, mx.x), fmaxf(amx.y, mx.y), fmaxf(amx.z, mx.z));
    }
    float lo[6] = {cmn.x, cmn.y, cmn.z, amn.x, amn.y, amn.z};
    float hi[6] = {cmx.x, cmx.y, cmx.z, amx.x, amx.y, amx.z};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        lo[k] = wave_min(lo[k]);
        hi[k] = wave_max(hi[k]);
    }
    __shared__ float slo[kThreads / 64][6], shi[kThreads / 64][6];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            slo[threadIdx.x >> 6][k] = lo[k];
            shi[threadIdx.x >> 6][k] = hi[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {   // one atomic per workgroup and value
        const int k = threadIdx.x;
        float a = slo[0][k], b = shi[0][k];
        for (int j = 1; j < kThreads / 64; ++j) {
            a = fminf(a, slo[j][k]);
            b = fmaxf(b, shi[j][k]);
        }
        if (k < 3) {
            atomicMin(&sb->cmin[k], enc_f32(a));
            atomicMax(&sb->cmax[k], enc_f32(b));
        } else {
            atomicMin(&sb->bmin[k - 3], enc_f32(a));
            atomicMax(&sb->bmax[k - 3], enc_f32(b));
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {  // bit i -> bit 3i
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// Morton keys of the box centroids. The grid spans the centroids' bounding box with ONE cell size for all three
// axes (the largest extent spans the grid), so cells are cubes: a thin scene such as a height field is not stretched
// along its thin axis, which would interleave splits that separate nothing a ray cares about. The bounds are read
// from device memory (k_bounds of the same stream), so the host does not wait for them.
__global__ __launch_bounds__(kThreads) void k_morton(uint32_t n, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                                                      const SceneBounds* __restrict__ sb, int bits, uint64_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
    float cmin[3], max_ext = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t lo = sb->cmin[k], hi = sb->cmax[k];
        cmin[k] = __uint_as_float(lo ^ ((lo >> 31) ? 0x80000000u : 0xFFFFFFFFu));     // inverse of enc_f32
        const float mx = __uint_as_float(hi ^ ((hi >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        const float ext = mx - cmin[k];
        if (ext < __builtin_inff() && ext > max_ext) max_ext = ext;                    // also skips NaN
    }
    const float qmax = (float)((1u << bits) - 1u);
    const float scale = (max_ext > 0.0f) ? (float)(1u << bits) / max_ext : 0.0f;
    for (uint32_t id = blockIdx.x * kThreads + threadIdx.x; id < n; id += gridDim.x * kThreads) {
        const float4 a = bmin[id], b = bmax[id];
        if (a.w != 0.0f) {   // a primitive kept out of the boxes (prim_wild): sorted behind all others
            keys[id] = (1ull << (3 * bits)) - 1ull;
            vals[id] = id;
            continue;
        }
        const float cx = a.x * 0.5f + b.x * 0.5f, cy = a.y * 0.5f + b.y * 0.5f, cz = a.z * 0.5f + b.z * 0.5f;
        const float qx = fminf(fmaxf((cx - cmin[0]) * scale, 0.0f), qmax);
        const float qy = fminf(fmaxf((cy - cmin[1]) * scale, 0.0f), qmax);
        const float qz = fminf(fmaxf((cz - cmin[2]) * scale, 0.0f), qmax);
        // Triangles and spheres share one order and one tree. (Keeping the kinds apart -- the kind as the top bit of the
        // key, so that no triangle sits under a node the traversal widens for spheres -- was measured on a mesh of 63 M
        // small triangles with 1.3 M small spheres above it and lost 5-9 %: a tree of spheres alone culls worse than
        // spheres filed among the triangles next to them, whichever subtree a ray walks first.)
        keys[id] = (spread21((uint64_t)qx) << 2) | (spread21((uint64_t)qy) << 1) | spread21((uint64_t)qz);
        vals[id] = id;
    }
}

// ---- exclusive scan over u32 (three launches) ------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
// Exclusive scan of one value per thread across a 256-thread workgroup; *total gets the sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* lds4, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) lds4[w] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int k = 0; k < kThreads / 64; ++k) {
        const uint32_t t = lds4[k];
        if (k < w) base += t;
        sum += t;
    }
    if (total) *total = sum;
    return base + inc - v;
}

__global__ __launch_bounds__(kThreads) void k_scan_partial(const uint32_t* __restrict__ data, uint32_t m, uint32_t* __restrict__ partial) {
    __shared__ uint32_t lds4[4];
    const uint32_t base = blockIdx.x * kScanChunk;
    uint32_t s = 0;
    for (int k = 0; k < 16; ++k) {
        const uint32_t i = base + k * kThreads + threadIdx.x;
        if (i < m) s += data[i];
    }
    uint32_t total;
    block_excl_scan(s, lds4, &total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}
__global__ __launch_bounds__(kThreads) void k_scan_top(uint32_t* __restrict__ partial, uint32_t p) {
    __shared__ uint32_t lds4[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < p; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < p) ? partial[i] : 0;
        uint32_t total;
        const uint32_t e = block_excl_scan(v, lds4, &total);
        if (i < p) partial[i] = carry + e;
        carry += total;
        __syncthreads();
    }
}
__global__ __launch_bounds__(kThreads) void k_scan_apply(uint32_t* __restrict__ data, uint32_t m, const uint32_t* __restrict__ partial) {
    __shared__ uint32_t lds4[4];
    const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 16;
    uint32_t v[16];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        v[k] = (base + k < m) ? data[base + k] : 0;
        s += v[k];
    }
    uint32_t run = block_excl_scan(s, lds4, nullptr) + partial[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (base + k < m) data[base + k] = run;
        run += v[k];
    }
}

// ---- LSD radix sort, 8-bit digits, staged through LDS ----------------------
__global__ __launch_bounds__(kThreads) void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift,
                                                         uint32_t* __restrict__ hist, uint32_t nb) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = base + k * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];   // digit-major: one scan gives every offset
}

__global__ __launch_bounds__(kThreads) void k_sort_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                            uint32_t n, int shift, const uint32_t* __restrict__ offsets, uint32_t nb) {
    __shared__ volatile uint32_t whist[4][256];   // volatile: lanes of one wavefront hand counts to each other
    __shared__ uint32_t dig_excl[256];
    __shared__ uint32_t gbase[256];
    __shared__ uint32_t lds4[4];
    __shared__ uint64_t skeys[kSortTile];
    __shared__ uint32_t svals[kSortTile];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tile = blockIdx.x * kSortTile;
    for (int k = 0; k < 4; ++k) whist[k][tid] = 0;
    __syncthreads();
    uint64_t key[kSortItems];
    uint32_t val[kSortItems];
    uint16_t rank[kSortItems];
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Each wavefront ranks a contiguous 1024-key chunk in 16 rounds of 64 keys; keys
    // with equal digits keep their input order (stable), as LSD sorting requires.
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t i = tile + w * (kSortTile / 4) + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : ~0ull;
        val[r] = valid ? vals_in[i] : 0u;
        const uint32_t dig = (uint32_t)(key[r] >> shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((dig >> b) & 1u);
            m &= ((dig >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = __popcll(m & lt_mask);
        const uint32_t pre = whist[w][dig];
        rank[r] = (uint16_t)(pre + before);
        if (valid && before == 0) whist[w][dig] = pre + __popcll(m);
    }
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t t = whist[k][tid];
        whist[k][tid] = run;
        run += t;
    }
    const uint32_t ex = block_excl_scan(run, lds4, nullptr);
    dig_excl[tid] = ex;
    gbase[tid] = offsets[tid * nb + blockIdx.x] - ex;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t i = tile + w * (kSortTile / 4) + r * 64 + lane;
        if (i < n) {
            const uint32_t dig = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t lp = dig_excl[dig] + whist[w][dig] + rank[r];
            skeys[lp] = key[r];
            svals[lp] = val[r];
        }
    }
    __syncthreads();
    const uint32_t count = min((uint32_t)kSortTile, n - tile);
    for (uint32_t i = tid; i < count; i += kThreads) {
        const uint64_t k = skeys[i];
        const uint32_t dst = gbase[(uint32_t)(k >> shift) & 255u] + i;
        keys_out[dst] = k;
        vals_out[dst] = svals[i];
    }
}

// ---- Karras 2012: one internal node per thread ------------------------------
__device__ __forceinline__ int kdelta(const uint64_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));   // duplicate keys: fall back to the position
    return __clzll((long long)(a ^ b));
}
// Everything hierarchy construction, refit and emission keep about one node of the binary tree, in one
// 64-byte line: those kernels follow parent / child links, so a node's fields are wanted together.
struct alignas(32) BuildNode {
    int l, r;               // children (inner nodes)
    int first, last;        // range of sorted primitives (inner nodes)
    int parent;             // -1 for the root
    uint32_t counter;       // children that have reported during refit
    uint32_t pad[2];
};
static_assert(sizeof(BuildNode) == 32, "two nodes per 64-byte line");
// The boxes live in an array of their own, two 16-byte quads per node: they are handed from one workgroup to another
// INSIDE the refit launch, and only write-through stores and cache-bypassing loads ever touch their lines there.
//   box[2 * node]     = padded box minimum; w = smallest sphere radius below (+inf: none)
//   box[2 * node + 1] = padded box maximum; w = bits of the height: 0 for a leaf or a subtree that becomes one
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_through(float4* p, float4 v) {   // 16-byte write-through store (system scope)
    const f32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void load_through2(const float4* p, float4& a, float4& b) {   // two quads, past L1 and the local L2
    f32x4 ra, rb;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(ra), "=&v"(rb) : "v"(p) : "memory");
    a = make_float4(ra.x, ra.y, ra.z, ra.w);
    b = make_float4(rb.x, rb.y, rb.z, rb.w);
}

// Node ids: internal i in [0, n-1) -> i ; leaf j in [0, n) -> (n-1) + j.
__global__ __launch_bounds__(kThreads) void k_karras(const uint64_t* __restrict__ keys, int n, BuildNode* __restrict__ bn) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (kdelta(keys, n, i, i + 1) - kdelta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = kdelta(keys, n, i, i - d);
    int lmax = 2;
    while (kdelta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (kdelta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = kdelta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (kdelta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = (lo == gamma) ? (n - 1) + gamma : gamma;
    const int right = (hi == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
    bn[i].l = left;
    bn[i].r = right;
    bn[i].first = lo;
    bn[i].last = hi;
    bn[i].counter = 0;
    bn[left].parent = i;
    bn[right].parent = i;
    if (i == 0) bn[0].parent = -1;
}

struct PadParams {
    float rel;        // boxes grow by rel * (largest |coordinate| of the box)
    float cam[3];     // the camera the safe region is laid around (rtx_bvh_build)
};

// Extra padding for a triangle whose two edges at v0 are less than 30 degrees apart. Moeller-Trumbore's t (poly.rs:149-176)
// is -(s.N) / (d.N) with N = e1 x e2 and s = origin - v0, and both dot products are sums of products of magnitude
// |s| |e1| |e2| (|d| |e1| |e2|) with a result of magnitude |s| |N| (|d| |N|): their rounding errors, a few u of the large
// magnitude each, move the reported hit along the ray by about
//     c u dist / sin(theta0),   theta0 = angle between e1 and e2, dist = distance from the ray origin, c ~ 16,
// -- and, for a ray that grazes the plane, out of the plane by the same amount however small the angle of incidence.
// For ordinary triangles that is inside the standard padding (128 u of the coordinates on the box side plus 128 u of the
// origin on the ray side). For a needle it is not: its reported t can lie in front of its own box, and once a bound on t
// exists (another hit, the light's distance) the box is culled although the exhaustive loop (scene/mod.rs:186-199) would
// have replaced the hit -- seen with an exact duplicate of a needle of 1.1e-3 rad, where the lower index must win the tie
// and lost it. The ray origins a tree has to answer for lie in the safe cube (DevScene::safe_r; others take the
// exhaustive loop), so dist <= 2.2 safe_r, and a needle's box grows by 32 u * 2.2 safe_r / sin(theta0) on every side:
// 0.06 % of safe_r at the 1e-3 rad below which a triangle is tested by every ray anyway (DevScene::always).
__device__ float sliver_pad(const float* __restrict__ p, const SceneBounds* __restrict__ sb, const float cam[3]) {
    const V3 v0 = v3(p[0], p[1], p[2]);
    const V3 e1 = v3(p[3], p[4], p[5]) - v0, e2 = v3(p[6], p[7], p[8]) - v0;
    const V3 c = vcross(e1, e2);
    const float cc = vdot(c, c), ll = vdot(e1, e1) * vdot(e2, e2);
    if (!(cc < 0.25f * ll)) return 0.0f;   // sin(theta0) >= 0.5, a degenerate product, or NaN
    const float sn = fmaxf(sqrtf(cc / ll), 5e-4f);
    // the safe cube's half side as rtx_bvh_build computes it from the same bounds
    float radius = 0.0f, dcam = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t lo = sb->bmin[k], hi = sb->bmax[k];
        const float mn = __uint_as_float(lo ^ ((lo >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        const float mx = __uint_as_float(hi ^ ((hi >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        radius = fmaxf(radius, 0.5f * (mx - mn));
        dcam = fmaxf(dcam, fabsf(cam[k] - (0.5f * mn + 0.5f * mx)));
    }
    if (!(radius < __builtin_inff())) radius = 0.0f;
    if (!(dcam < __builtin_inff())) dcam = 0.0f;
    // (kept finite for scenes with coordinates near the end of the f32 range, whose safe cube has no finite side)
    const float safe_r = fminf(fmaxf(fmaxf(4.0f * radius, 2.0f * dcam + 2.0f * radius), 1e-3f), 1e37f);
    return fminf((32.0f * 5.9604645e-8f * 2.2f * 1.01f) * safe_r / sn, 1e36f);
}

// Positive finite f32 -> f16 bits, rounded toward zero (NaN and values <= 0 give 0).
__device__ __forceinline__ uint32_t f16_bits_down(float r) {
    if (!(r > 0.0f)) return 0u;
    if (r >= 65504.0f) return 0x7bffu;
    const uint32_t b = __float_as_uint(r);
    const int e = (int)(b >> 23) - 127;
    if (e >= -14) return ((uint32_t)(e + 15) << 10) | ((b & 0x7fffffu) >> 13);
    return (uint32_t)(r * 16777216.0f);   // subnormal f16: multiples of 2^-24, truncated
}

// Refit, one launch. Every thread starts at a leaf -- the padded primitive box: padding covers primitive tests that
// accept a ray passing marginally outside the exact shape (f32 rounding in Moeller-Trumbore); the sphere discriminant's
// cancellation error grows with the distance between ray origin and centre and is covered at traversal time from the
// smallest sphere radius of each subtree, carried in the box minimum's w (+inf: no sphere) -- and walks towards the root.
// At every inner node the first child to arrive stops; the second one finds its sibling's box complete, forms the
// union (BoundingBox +, shapes/mod.rs:86-104) and goes on.
// A workgroup's 256 leaves are consecutive in Morton order, so nearly all inner nodes above them (about 250 of 255)
// cover leaves of this workgroup only -- a node's index lies inside its own leaf range, so "first and last leaf belong
// to this workgroup" identifies them and `index - first leaf of the workgroup` is a slot in LDS. For those nodes the
// arrival counter and the children's boxes live in LDS (an LDS atomic and two LDS reads per step instead of two
// dependent trips to memory); only the few nodes whose leaves span workgroups use the device-wide protocol:
// the two arrivals may run on different CUs and XCDs, whose L1s and L2s do not see each other's plain stores, so the
// arriving child's box is stored write-through (sc0 sc1), the storing wavefront waits for its stores (vmcnt(0)) before
// it counts its arrival with a device-scope atomic, and the wavefront whose arrival came second loads the sibling's box
// with loads that bypass the caches (sc0 sc1). No cache is flushed or invalidated on the way (agent-scope fences per
// step cost this launch a factor of five). Every box also goes to box[] with a plain store for the launches that follow.
// *root_height receives the height of the binary tree. rtx_set_option("verify_build") recomputes every box afterwards.
__global__ __launch_bounds__(kThreads) void k_refit(DevScene s, uint32_t n, const uint32_t* __restrict__ sorted_ids,
                                                     const SceneBounds* __restrict__ sb,
                                                     const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                                                     BuildNode* __restrict__ bn, float4* __restrict__ box, PadParams pp, int leaf_max,
                                                     uint32_t* __restrict__ root_height) {
    __shared__ uint32_t s_cnt[kThreads];        // arrivals at the inner nodes local to this workgroup
    __shared__ float4 s_leaf[kThreads][2];      // boxes of this workgroup's leaves
    __shared__ float4 s_inner[kThreads][2];     // boxes of its local inner nodes
    const uint32_t bs = blockIdx.x * kThreads, be = min(bs + (uint32_t)kThreads, n);   // this workgroup's leaves
    const uint32_t j = bs + threadIdx.x;
    const bool live = j < n;
    s_cnt[threadIdx.x] = 0;
    int me = -1, cur = -1;
    float4 mn = make_float4(0.0f, 0.0f, 0.0f, 0.0f), mx = mn;
    uint32_t h = 0;
    if (live) {
        const uint32_t id = sorted_ids[j];
        float4 a = bmin[id], b = bmax[id];
        const bool wild = a.w != 0.0f;
        if (wild) {   // tested by every ray (DevScene::always); its leaf is a point at the scene's lower corner
            float c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t lo = sb->bmin[k];
                c[k] = __uint_as_float(lo ^ ((lo >> 31) ? 0x80000000u : 0xFFFFFFFFu));
                if (!(fabsf(c[k]) <= 1e18f)) c[k] = 0.0f;   // a scene of nothing but such primitives has no bounds
            }
            a = make_float4(c[0], c[1], c[2], 0.0f);
            b = a;
        }
        const float m = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(b.x))), fmaxf(fabsf(b.y), fabsf(b.z)));
        float pad = pp.rel * m + 1e-30f;
        const uint32_t ref = s.prim_ref[id];
        float rmin = __builtin_inff();
        if (wild) {
            // no margin, no flag: the leaf is never what finds this primitive
        } else if (ref >> 31) {
            rmin = fabsf(s.sph[ref & 0x7fffffffu].w);
            if (!(rmin >= 0.0f)) rmin = 0.0f;   // NaN radius
        } else {
            pad += sliver_pad(s.tri_pos + (size_t)(ref & 0x7fffffffu) * 9, sb, pp.cam);
        }
        me = (int)((n - 1) + j);
        mn = make_float4(a.x - pad, a.y - pad, a.z - pad, rmin);
        mx = make_float4(b.x + pad, b.y + pad, b.z + pad, __uint_as_float(0u));
        box[2 * (size_t)me] = mn;
        box[2 * (size_t)me + 1] = mx;
        s_leaf[threadIdx.x][0] = mn;
        s_leaf[threadIdx.x][1] = mx;
        cur = (n > 1) ? bn[me].parent : -1;
    }
    __syncthreads();   // counters cleared, leaf boxes in place
    // bounded by the tree height; every lane leaves the loop (first arrival, or the root)
    while (__ballot(cur >= 0)) {
        int l = 0, r = 0, first = 0, last = 0, parent = -1;
        if (cur >= 0) {   // written by the launch before this one
            l = bn[cur].l; r = bn[cur].r; first = bn[cur].first; last = bn[cur].last; parent = bn[cur].parent;
        }
        const bool local = cur >= 0 && (uint32_t)first >= bs && (uint32_t)last < be;
        const bool global = cur >= 0 && !local;
        bool second = false;
        float4 sa = mn, sb = mx;
        const int sib = (l == me) ? r : l;
        if (local) {
            // this lane's box is in LDS already (written below, or above for a leaf); LDS serves one wavefront's accesses in order
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t old = __hip_atomic_fetch_add(&s_cnt[(uint32_t)cur - bs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            second = old == 1u;
            if (second) {
                if (sib >= (int)(n - 1)) {
                    sa = s_leaf[(uint32_t)sib - (n - 1) - bs][0];
                    sb = s_leaf[(uint32_t)sib - (n - 1) - bs][1];
                } else {
                    sa = s_inner[(uint32_t)sib - bs][0];
                    sb = s_inner[(uint32_t)sib - bs][1];
                }
            }
        }
        if (__ballot(global)) {
            if (global) {
                store_through(&box[2 * (size_t)me], mn);
                store_through(&box[2 * (size_t)me + 1], mx);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's boxes have reached memory before its arrivals count
            if (global) {
                const uint32_t old = __hip_atomic_fetch_add(&bn[cur].counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                second = old == 1u;
                if (second) load_through2(&box[2 * (size_t)sib], sa, sb);
            }
        }
        if (!second) cur = -1;
        if (second) {
            mn = make_float4(fminf(mn.x, sa.x), fminf(mn.y, sa.y), fminf(mn.z, sa.z), fminf(mn.w, sa.w));
            const int size = last - first + 1;
            h = (size <= leaf_max) ? 0u : min(1u + max(h, __float_as_uint(sb.w)), 255u);
            mx = make_float4(fmaxf(mx.x, sb.x), fmaxf(mx.y, sb.y), fmaxf(mx.z, sb.z), __uint_as_float(h));
            box[2 * (size_t)cur] = mn;
            box[2 * (size_t)cur + 1] = mx;
            if (local) {
                s_inner[(uint32_t)cur - bs][0] = mn;
                s_inner[(uint32_t)cur - bs][1] = mx;
                bn[cur].counter = 2u;   // both arrivals were counted in LDS; the record says so like any other node's
            }
            me = cur;
            cur = parent;
            if (cur < 0) *root_height = h;   // the root is complete
        }
    }
}

// Optional check of the refit's hand-overs (the "verify_build" option), in a launch of its own, where every box is
// visible: each inner node's box and height are recomputed from its children's and compared bit for bit.
__global__ __launch_bounds__(kThreads) void k_verify_refit(int n, const BuildNode* __restrict__ bn, const float4* __restrict__ box, int leaf_max,
                                                            uint32_t* __restrict__ mismatches) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    const int l = bn[i].l, r = bn[i].r;
    const float4 la = box[2 * (size_t)l], lb = box[2 * (size_t)l + 1], ra = box[2 * (size_t)r], rb = box[2 * (size_t)r + 1];
    const float4 a = box[2 * (size_t)i], b = box[2 * (size_t)i + 1];
    const int size = bn[i].last - bn[i].first + 1;
    const uint32_t h = (size <= leaf_max) ? 0u : min(1u + max(__float_as_uint(lb.w), __float_as_uint(rb.w)), 255u);
    auto same = [](float x, float y) { return __float_as_uint(x) == __float_as_uint(y); };
    const bool ok = same(a.x, fminf(la.x, ra.x)) && same(a.y, fminf(la.y, ra.y)) && same(a.z, fminf(la.z, ra.z)) && same(a.w, fminf(la.w, ra.w)) &&
                    same(b.x, fmaxf(lb.x, rb.x)) && same(b.y, fmaxf(lb.y, rb.y)) && same(b.z, fmaxf(lb.z, rb.z)) && __float_as_uint(b.w) == h &&
                    bn[i].counter == 2u;
    if (!ok) atomicAdd(mismatches, 1u);
}

// Reference to binary node c as a child of a traversal node: a leaf (single primitive
// or a collapsed subtree of at most leaf_max primitives) or an inner node.
__device__ __forceinline__ bool is_expandable(int c, int n, const BuildNode* __restrict__ bn, int leaf_max) {
    return c < n - 1 && (bn[c].last - bn[c].first + 1) > leaf_max;
}
__device__ __forceinline__ int child_ref(int c, int n, const BuildNode* __restrict__ bn, uint32_t* leaf_count, int leaf_max) {
    if (c >= n - 1) {
        ++*leaf_count;
        return (int)~(((uint32_t)(c - (n - 1)) << 3) | 0u);
    }
    const int first = bn[c].first, size = bn[c].last - first + 1;
    if (size <= leaf_max) {
        ++*leaf_count;
        return (int)~(((uint32_t)first << 3) | (uint32_t)(size - 1));
    }
    return c;
}

// Power-of-two grid step (as a biased f32 exponent) such that extent / step <= 254.
__device__ __forceinline__ uint32_t step_exponent(float extent) {
    if (!(extent > 0.0f)) return 1u;
    int ex;
    (void)frexpf(extent / 254.0f, &ex);    // extent/254 = m * 2^ex, m in [0.5, 1)  ->  2^ex >= extent/254
    return (uint32_t)min(max(ex + 127, 1), 254);
}

// Emits the 4-wide traversal nodes, top down, one level of the 4-wide tree per launch. A
// traversal node is rooted at a binary node i; its children start
// as i's two children, and while there are fewer than four, the child with the largest
// surface area that is still an inner node is replaced by its own two children -- the big
// boxes are the ones a ray is most likely to enter, so opening them here saves a step there.
// Children that remain inner nodes root the traversal nodes of the next level.
// Nodes are numbered in level order, densely: the k-th root of a level is node base(level) + k, base(level) = the
// number of nodes of all levels above, so the array has exactly one slot per node and the top of the tree -- what
// every ray reads -- is contiguous. A workgroup reserves the slots of all its children at the next level with one
// atomic (a per-wavefront reservation on one counter serialises at ~90 per microsecond).
// Child boxes are quantised to 8 bits per plane relative to the node's own box, rounded
// outward and verified with the same fma the traversal uses, so a quantised box always
// contains the padded box.
struct LevelInfo {
    uint32_t count;   // roots at this level (reserved by the level above)
    uint32_t leaves;  // leaf references emitted by this level
};
__device__ __forceinline__ float half_area(const float4& mn, const float4& mx) {
    const float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(kThreads) void k_emit4(int n, uint32_t level, const int* __restrict__ qin, int* __restrict__ qout,
                                                     LevelInfo* __restrict__ lvl, const BuildNode* __restrict__ bn, const float4* __restrict__ box,
                                                     BvhNode* __restrict__ nodes, int leaf_max) {
    __shared__ uint32_t s_leaves, s_base, lds4[4];
    if (threadIdx.x == 0) s_leaves = 0;
    __syncthreads();
    uint32_t my_leaves = 0;
    const uint32_t cnt = lvl[level].count;
    // nodes of all levels above: their counts are final (their launches have ended)
    uint32_t node_base = 0;
    for (uint32_t k = 0; k < level; ++k) node_base += lvl[k].count;
    const uint32_t next_base = node_base + cnt;
    for (uint32_t slot0 = blockIdx.x * kThreads; slot0 < cnt; slot0 += gridDim.x * kThreads) {   // block-uniform trip count
        const uint32_t slot = slot0 + threadIdx.x;
        const bool live = slot < cnt;
        int ch[4] = {0, 0, 0, 0};
        int nc = 0;
        uint32_t inner = 0;   // children that root a node of the next level
        int i = 0;
        if (live) {
            i = qin[slot];
            ch[0] = bn[i].l;
            ch[1] = bn[i].r;
            nc = 2;
            while (nc < 4) {
                int best = -1;
                float best_area = -1.0f;
                for (int k = 0; k < nc; ++k) {
                    if (!is_expandable(ch[k], n, bn, leaf_max)) continue;
                    const float a = half_area(box[2 * (size_t)ch[k]], box[2 * (size_t)ch[k] + 1]);
                    if (!(a <= best_area)) {   // also takes a NaN area rather than stopping early
                        best = k;
                        best_area = a;
                    }
                }
                if (best < 0) break;
                const int c = ch[best];
                ch[best] = bn[c].l;
                ch[nc++] = bn[c].r;
            }
            for (int k = 0; k < nc; ++k) inner += is_expandable(ch[k], n, bn, leaf_max) ? 1u : 0u;
        }
        uint32_t total;
        const uint32_t offset = block_excl_scan(inner, lds4, &total);
        if (threadIdx.x == 0) s_base = total ? atomicAdd(&lvl[level + 1].count, total) : 0u;
        __syncthreads();
        uint32_t pos = s_base + offset;   // position of this thread's first inner child in the next level
        if (live) {
            const float4 bmn = box[2 * (size_t)i], bmx = box[2 * (size_t)i + 1];
            const float org[3] = {bmn.x, bmn.y, bmn.z};
            const float ext[3] = {bmx.x - bmn.x, bmx.y - bmn.y, bmx.z - bmn.z};
            uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0}, eb[3];
            float cmn[4][3], cmx[4][3];
            uint32_t rcode[4] = {0x7c00u, 0x7c00u, 0x7c00u, 0x7c00u};   // smallest sphere radius per child, f16; +inf = none
            bool spheres = false;
            uint32_t sphere_bits = 0;   // bit k: a sphere somewhere below child k
            for (int k = 0; k < nc; ++k) {
                const float4 a = box[2 * (size_t)ch[k]], b = box[2 * (size_t)ch[k] + 1];
                if (a.w < __builtin_inff()) {
                    rcode[k] = f16_bits_down(a.w);
                    spheres = true;
                    sphere_bits |= 1u << k;
                }
                cmn[k][0] = a.x; cmn[k][1] = a.y; cmn[k][2] = a.z;
                cmx[k][0] = b.x; cmx[k][1] = b.y; cmx[k][2] = b.z;
            }
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                uint32_t e = step_exponent(ext[ax]);
                for (;;) {   // raise the step until every child plane is representable (at most a few rounds)
                    const float step = __uint_as_float(e << 23);
                    bool ok = true;
                    uint32_t lo_pack = 0, hi_pack = 0;
                    for (int k = 0; k < nc; ++k) {
                        float lo = floorf((cmn[k][ax] - org[ax]) / step);
                        lo = fminf(fmaxf(lo, 0.0f), 255.0f);
                        while (lo > 0.0f && __builtin_fmaf(lo, step, org[ax]) > cmn[k][ax]) lo -= 1.0f;
                        float hi = ceilf((cmx[k][ax] - org[ax]) / step);
                        hi = fminf(fmaxf(hi, 0.0f), 255.0f);
                        while (hi < 255.0f && __builtin_fmaf(hi, step, org[ax]) < cmx[k][ax]) hi += 1.0f;
                        if (__builtin_fmaf(lo, step, org[ax]) > cmn[k][ax] || __builtin_fmaf(hi, step, org[ax]) < cmx[k][ax]) ok = false;
                        lo_pack |= (uint32_t)lo << (8 * k);
                        hi_pack |= (uint32_t)hi << (8 * k);
                    }
                    if (ok || e >= 254u) {
                        qlo[ax] = lo_pack;
                        qhi[ax] = hi_pack;
                        eb[ax] = e;
                        break;
                    }
                    ++e;
                }
            }
            int ref[4] = {0, 0, 0, 0};
            for (int k = 0; k < nc; ++k) {
                ref[k] = child_ref(ch[k], n, bn, &my_leaves, leaf_max);
                if (ref[k] >= 0) {                 // roots a node of the next level
                    qout[pos] = ref[k];
                    ref[k] = (int)(next_base + pos);
                    ++pos;
                }
            }
            BvhNode nd;
            nd.a = make_uint4(__float_as_uint(org[0]), __float_as_uint(org[1]), __float_as_uint(org[2]),
                              eb[0] | (eb[1] << 8) | (eb[2] << 16) | ((uint32_t)nc << 24) | (sphere_bits << 27) | (spheres ? 0x80000000u : 0u));
            nd.b = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
            nd.c = make_uint4(qhi[1], qhi[2], (uint32_t)ref[0], (uint32_t)ref[1]);
            nd.d = make_uint4((uint32_t)ref[2], (uint32_t)ref[3], rcode[0] | (rcode[1] << 16), rcode[2] | (rcode[3] << 16));
            nodes[node_base + slot] = nd;
        }
        __syncthreads();   // s_base and lds4 are reused by the next pass
    }
    if (my_leaves) atomicAdd(&s_leaves, my_leaves);     // LDS atomic
    __syncthreads();
    if (threadIdx.x == 0 && s_leaves) atomicAdd(&lvl[level].leaves, s_leaves);
}

// Traversal records, in `order` (sorted ids) or in id order when order == nullptr.
// flat / n_flat (with a tree): positions of the records that every ray tests (DevScene::always): triangles without an
// area to speak of, |e1 x e2| <= 1e-3 |e1| |e2|, and primitives with non-finite or enormous coordinates (prim_wild);
// the list holds at most s.n_prims entries, n_flat counts them.
__global__ __launch_bounds__(kThreads) void k_make_recs(DevScene s, const uint32_t* __restrict__ order, PrimRec* __restrict__ recs,
                                                         uint32_t* __restrict__ flat, uint32_t* __restrict__ n_flat) {
    for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < s.n_prims; j += gridDim.x * kThreads) {
        const uint32_t id = order ? order[j] : j;
        const uint32_t ref = s.prim_ref[id];
        const uint32_t idx = ref & 0x7fffffffu;
        const bool wild = flat && prim_wild(s, id);
        if (wild) flat[atomicAdd(n_flat, 1u)] = j;
        PrimRec rc;
        if (ref >> 31) {
            const float4 sp = s.sph[idx];
            rc.q0 = make_float4(sp.x, sp.y, sp.z, __uint_as_float(id));
            rc.q1 = make_float4(sp.w, 0.0f, 0.0f, __uint_as_float(1u));
            rc.q2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        } else {
            const float* p = s.tri_pos + (size_t)idx * 9;
            const V3 v0 = v3(p[0], p[1], p[2]);
            const V3 e1 = v3(p[3], p[4], p[5]) - v0;   // poly.rs:145-146
            const V3 e2 = v3(p[6], p[7], p[8]) - v0;
            rc.q0 = make_float4(v0.x, v0.y, v0.z, __uint_as_float(id));
            rc.q1 = make_float4(e1.x, e1.y, e1.z, __uint_as_float(0u));
            rc.q2 = make_float4(e2.x, e2.y, e2.z, 0.0f);
            if (flat) {
                const V3 c = vcross(e1, e2);
                if (!wild && vdot(c, c) <= 1e-6f * vdot(e1, e1) * vdot(e2, e2)) flat[atomicAdd(n_flat, 1u)] = j;   // false for NaN: such a triangle is never hit
            }
        }
        recs[j] = rc;
    }
}

__global__ __launch_bounds__(kThreads) void k_copy_recs(const PrimRec* __restrict__ recs, const uint32_t* __restrict__ which, uint32_t n,
                                                         PrimRec* __restrict__ out) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) out[i] = recs[which[i]];
}

// prim_ref from device-resident index arrays (rtx_scene_set_device): ref[id] = element | kind bit, with the
// permutation check of the host path: status[0] is raised when an id is out of range or taken twice
// (ref starts as all ones; every id must be claimed exactly once, and n claims of n distinct ids fill the array),
// status[1] when some triangle asks for vertex normals.
__global__ __launch_bounds__(kThreads) void k_prim_ref(uint32_t n_tris, uint32_t n_spheres, const uint32_t* __restrict__ tri_index,
                                                        const uint32_t* __restrict__ sph_index, const uint8_t* __restrict__ tri_flags,
                                                        uint32_t* __restrict__ ref, uint32_t* __restrict__ status) {
    const uint32_t n = n_tris + n_spheres;
    bool bad = false, normals = false;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        uint32_t id, val;
        if (i < n_tris) {
            id = tri_index ? tri_index[i] : i;
            val = i;
            normals |= (tri_flags[i] & RTX_TRI_VERTEX_NORMAL) != 0;
        } else {
            const uint32_t k = i - n_tris;
            id = sph_index ? sph_index[k] : n_tris + k;
            val = k | 0x80000000u;
        }
        if (id >= n) bad = true;
        else if (atomicExch(&ref[id], val) != 0xFFFFFFFFu) bad = true;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(&status[0], 1u);
    if (__ballot(normals) && (threadIdx.x & 63) == 0) atomicOr(&status[1], 1u);
}

// DevScene::shade_recs from the scene's arrays (one thread per reference primitive id).
__global__ __launch_bounds__(kThreads) void k_shade_recs(DevScene s, uint4* __restrict__ out) {
    for (uint32_t id = blockIdx.x * kThreads + threadIdx.x; id < s.n_prims; id += gridDim.x * kThreads) {
        const uint32_t ref = s.prim_ref[id];
        const uint32_t idx = ref & 0x7fffffffu;
        uint4 a, b, c = make_uint4(0u, 0u, 0u, 0u), d = make_uint4(0u, 0u, 0u, 0u);
        if (ref >> 31) {
            const float4 sp = s.sph[idx];
            a = make_uint4(__float_as_uint(sp.x), __float_as_uint(sp.y), __float_as_uint(sp.z), ref);
            b = make_uint4(__float_as_uint(sp.w), 0u, 0u, s.sph_mat[idx]);
        } else {
            const float* p = s.tri_pos + (size_t)idx * 9;
            const uint32_t* m = s.tri_mat + (size_t)idx * 3;
            a = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), idx | ((uint32_t)(s.tri_flags[idx] & 3u) << 28));
            b = make_uint4(__float_as_uint(p[3]), __float_as_uint(p[4]), __float_as_uint(p[5]), m[0]);
            c = make_uint4(__float_as_uint(p[6]), __float_as_uint(p[7]), __float_as_uint(p[8]), m[1]);
            d.x = m[2];
        }
        uint4* o = out + (size_t)id * 4;
        o[0] = a; o[1] = b; o[2] = c; o[3] = d;
    }
}

template <typename T>
int dev_alloc(rtx_context* ctx, std::vector<void*>& owner, T** out, size_t count) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return rtx_fail(ctx, RTX_ERR_OUT_OF_MEMORY, std::string("hipMalloc: ") + hipGetErrorString(e));
    owner.push_back(p);
    *out = (T*)p;
    return RTX_OK;
}

// Temporaries of a build: carved out of one arena, hipMalloc for what does not fit.
struct TmpArena {
    rtx_context* ctx;
    std::vector<void*> extra;   // fallbacks, freed at the end
    char* base = nullptr;
    size_t size = 0, used = 0;
    template <typename T>
    int get(T** out, size_t count) {
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes <= size) {
            *out = (T*)(base + used);
            used += bytes;
            return RTX_OK;
        }
        return dev_alloc(ctx, extra, out, count);
    }
};

inline uint32_t grid_for(uint64_t n, uint32_t cap = 4096) {
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + kThreads - 1) / kThreads, cap));
}

}  // namespace

int rtx_build_shade_recs(rtx_context* ctx, const DevScene& s, uint4* d_recs) {
    static_assert((RTX_TRI_VERTEX_NORMAL | RTX_TRI_VERTEX_MATERIAL) == 3, "the record keeps two flag bits");
    if (s.n_prims > 0) hipLaunchKernelGGL(k_shade_recs, dim3(grid_for(s.n_prims)), dim3(kThreads), 0, ctx->stream, s, d_recs);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

int rtx_build_prim_ref(rtx_context* ctx, uint32_t n_tris, uint32_t n_spheres, const uint32_t* d_tri_index, const uint32_t* d_sph_index,
                       const uint8_t* d_tri_flags, uint32_t* d_ref, uint32_t* d_status) {
    const uint64_t n = (uint64_t)n_tris + n_spheres;
    RTX_HIP(ctx, hipMemsetAsync(d_ref, 0xFF, std::max<uint64_t>(n, 1) * sizeof(uint32_t), ctx->stream));
    RTX_HIP(ctx, hipMemsetAsync(d_status, 0, 2 * sizeof(uint32_t), ctx->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_prim_ref, dim3(grid_for(n)), dim3(kThreads), 0, ctx->stream, n_tris, n_spheres, d_tri_index, d_sph_index,
                           d_tri_flags, d_ref, d_status);
    RTX_HIP(ctx, hipGetLastError());
    return RTX_OK;
}

// Overflow area of the per-lane traversal stacks (see TravStack) and, when asked for, the stamp buffer: owned by the
// context, not shared with clones. A 4-wide level pushes at most 3 entries; with transparent materials the top
// 8 * RTX_MAX_DEPTH entries hold the suspended calls of a shadow ray's recursion.
int rtx_alloc_traversal_scratch(rtx_context* ctx) {
    DevScene& s = ctx->ds;
    for (void* p : ctx->trav_allocs) (void)hipFree(p);
    ctx->trav_allocs.clear();
    s.spill_entries = ctx->trav_spill_need + (s.has_transparent ? 8 * RTX_MAX_DEPTH : 1);
    s.spill_stride = ctx->num_cus * 8 * 256;
    int* spill = nullptr;
    int rc = dev_alloc(ctx, ctx->trav_allocs, &spill, (size_t)s.spill_entries * s.spill_stride);
    if (rc != RTX_OK) return rc;
    s.spill = spill;
    s.stamps = nullptr;
    if (ctx->opt_stamps) {
        unsigned long long* stamps = nullptr;
        const size_t count = (size_t)2 * (s.spill_stride / 64) * 4 + ((size_t)2 << 19);   // + per-unit records of diagnostic builds
        rc = dev_alloc(ctx, ctx->trav_allocs, &stamps, count);
        if (rc != RTX_OK) return rc;
        RTX_HIP(ctx, hipMemsetAsync(stamps, 0, count * sizeof(unsigned long long), ctx->stream));
        s.stamps = stamps;
    }
    return RTX_OK;
}

int rtx_bvh_build(rtx_context* ctx) {
    DevScene& s = ctx->ds;
    const uint32_t n = s.n_prims;
    hipStream_t st = ctx->stream;
    ctx->build_store.reset();      // a clone that shares the old tree keeps it alive
    auto store = std::make_shared<DevStore>();
    ctx->built = false;
    s.always = nullptr;
    s.n_always = 0;
    rtx_build_stats bs{};
    bs.n_prims = n;
    bs.upload_ms = ctx->build_stats.upload_ms;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    RTX_HIP(ctx, hipEventCreate(&ev0));
    if (hipEventCreate(&ev1) != hipSuccess) {
        (void)hipEventDestroy(ev0);
        return rtx_fail(ctx, RTX_ERR_HIP, "hipEventCreate failed");
    }
    if (hipEventRecord(ev0, st) != hipSuccess) {
        (void)hipEventDestroy(ev0);
        (void)hipEventDestroy(ev1);
        return rtx_fail(ctx, RTX_ERR_HIP, "hipEventRecord failed");
    }

    // Temporaries come out of one arena (a dozen hipMalloc calls cost more than the kernels of a 1 M-primitive
    // build); it stays with the context for the next build unless it is large. What does not fit falls back to hipMalloc.
    TmpArena tmp{ctx};
    {
        const size_t want = (size_t)n * 272 + (8u << 20);
        if (ctx->build_arena_size < want) {
            if (ctx->build_arena) (void)hipFree(ctx->build_arena);
            ctx->build_arena = nullptr;
            ctx->build_arena_size = 0;
            void* p = nullptr;
            if (hipMalloc(&p, want) == hipSuccess) {
                ctx->build_arena = p;
                ctx->build_arena_size = want;
            } else {
                (void)hipGetLastError();   // piecewise allocation below may still fit
            }
        }
        tmp.base = (char*)ctx->build_arena;
        tmp.size = ctx->build_arena_size;
    }
    auto cleanup = [&]() {   // every way out of the build, failures included
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        ev0 = ev1 = nullptr;
        for (void* p : tmp.extra) (void)hipFree(p);
        tmp.extra.clear();
        if (ctx->build_arena_size > ((size_t)1 << 30)) {   // do not sit on gigabytes between builds
            (void)hipFree(ctx->build_arena);
            ctx->build_arena = nullptr;
            ctx->build_arena_size = 0;
        }
    };
#define TRY(x) do { int rc_ = (x); if (rc_ != RTX_OK) { cleanup(); return rc_; } } while (0)
#define TRY_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cleanup(); \
        return rtx_fail(ctx, RTX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)

    PrimRec* recs = nullptr;
    TRY(dev_alloc(ctx, store->ptrs, &recs, (size_t)n + 1));   // one spare record: leaf fetches read 64 bytes (fetch_ref)
    TRY_HIP(hipMemsetAsync(recs + n, 0, sizeof(PrimRec), st));

    // Scene bounds are needed in both modes (safe region of the padded tests).
    float4 *bmin = nullptr, *bmax = nullptr;
    SceneBounds* sb = nullptr;
    TRY(tmp.get(&bmin, n));
    TRY(tmp.get(&bmax, n));
    TRY(tmp.get(&sb, 1));
    SceneBounds init;
    for (int k = 0; k < 3; ++k) { init.cmin[k] = init.bmin[k] = 0xFFFFFFFFu; init.cmax[k] = init.bmax[k] = 0u; }
    TRY_HIP(hipMemcpyAsync(sb, &init, sizeof(init), hipMemcpyHostToDevice, st));
    if (n > 0) hipLaunchKernelGGL(k_bounds, dim3(grid_for(n, 1024)), dim3(kThreads), 0, st, s, bmin, bmax, sb);
    // The bounds stay on the device for the Morton keys; the host reads them (safe region of the padded tests, needed
    // in both modes) at the end of the build.
    SceneBounds hb;
    s.pad_rel = 1.0f / 131072.0f;   // 2^-17, about 128 ulp of the coordinate magnitude
    PadParams pp;
    pp.rel = s.pad_rel;
    for (int k = 0; k < 3; ++k) pp.cam[k] = ctx->camera.pos[k];

    const int leaf_max = RTX_LEAF_MAX;   // the traversal kernels test exactly the record a leaf reference points at
    uint32_t levels4 = 0;   // height of the 4-wide tree
    const bool want_bvh = (ctx->opt_force_mode == 2) ? (n > (uint32_t)leaf_max)
                        : (ctx->opt_force_mode == 1) ? false : (n > RTX_EXHAUSTIVE_MAX);
    if (!want_bvh) {
        if (n > 0) hipLaunchKernelGGL(k_make_recs, dim3(grid_for(n)), dim3(kThreads), 0, st, s, (const uint32_t*)nullptr, recs, (uint32_t*)nullptr, (uint32_t*)nullptr);
        s.recs = recs;
        s.nodes = nullptr;
        s.n_nodes = 0;
        s.uses_bvh = 0;
    } else {
        // Morton keys: enough bits per axis to separate ~n cells, at most 21.
        int bits = 4;
        while (bits < 21 && (1ull << (3 * bits)) < (uint64_t)n * 4096ull) ++bits;
        bs.key_bits = 3 * bits;
        uint64_t* keys[2];
        uint32_t* vals[2];
        for (int k = 0; k < 2; ++k) { TRY(tmp.get(&keys[k], n)); TRY(tmp.get(&vals[k], n)); }
        const int key_bits = 3 * bits;
        hipLaunchKernelGGL(k_morton, dim3(grid_for(n)), dim3(kThreads), 0, st, n, bmin, bmax, sb, bits, keys[0], vals[0]);

        const uint32_t nb = (n + kSortTile - 1) / kSortTile;
        const uint32_t m = 256u * nb;
        const uint32_t np = (m + kScanChunk - 1) / kScanChunk;
        uint32_t *hist = nullptr, *partial = nullptr;
        TRY(tmp.get(&hist, m));
        TRY(tmp.get(&partial, np));
        int cur = 0;
        for (int shift = 0; shift < key_bits; shift += 8) {
            hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(kThreads), 0, st, keys[cur], n, shift, hist, nb);
            hipLaunchKernelGGL(k_scan_partial, dim3(np), dim3(kThreads), 0, st, hist, m, partial);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kThreads), 0, st, partial, np);
            hipLaunchKernelGGL(k_scan_apply, dim3(np), dim3(kThreads), 0, st, hist, m, partial);
            hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(kThreads), 0, st, keys[cur], vals[cur], keys[cur ^ 1],
                               vals[cur ^ 1], n, shift, hist, nb);
            cur ^= 1;
        }
        const uint64_t* skeys = keys[cur];
        const uint32_t* sids = vals[cur];

        constexpr uint32_t kMaxLevels = 256;   // the binary tree is at most 255 high (k_refit saturates there)
        int* queue[2];
        LevelInfo* lvl;
        uint32_t* root_height;
        BuildNode* bn;
        float4* box;
        BvhNode* nodes_tmp;
        const size_t total = 2 * (size_t)n - 1;
        TRY(tmp.get(&bn, total));
        TRY(tmp.get(&box, 2 * total));
        TRY(tmp.get(&queue[0], n));
        TRY(tmp.get(&queue[1], n));
        TRY(tmp.get(&lvl, kMaxLevels + 2));
        TRY(tmp.get(&root_height, 1));
        TRY_HIP(hipMemsetAsync(lvl, 0, (kMaxLevels + 2) * sizeof(LevelInfo), st));
        TRY_HIP(hipMemsetAsync(root_height, 0, 4, st));
        hipLaunchKernelGGL(k_karras, dim3((n - 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, skeys, (int)n, bn);
        hipLaunchKernelGGL(k_refit, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, st, s, n, sids, sb, bmin, bmax, bn, box, pp, leaf_max,
                           root_height);
        uint32_t* mismatches = nullptr;
        if (ctx->opt_verify_build) {
            TRY(tmp.get(&mismatches, 1));
            TRY_HIP(hipMemsetAsync(mismatches, 0, 4, st));
            hipLaunchKernelGGL(k_verify_refit, dim3((n - 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (int)n, bn, box, leaf_max, mismatches);
        }
        // Emission: one launch per level of the 4-wide tree, from the root down, into a temporary array with one slot per
        // binary inner node (an upper bound); the nodes that exist are then copied to an allocation of their exact size.
        // How many levels there are is not known here (at most the binary tree's height, usually about half of it), and
        // asking the device would stall the stream in mid-build: a first round of launches covers trees of up to 2^36
        // well-spread primitives, a launch for a level that does not exist ends at once, and the build's one wait -- after
        // everything is enqueued -- tells whether a deeper tree needs further rounds.
        TRY(tmp.get(&nodes_tmp, n - 1));
        uint32_t *flat = nullptr, *n_flat = nullptr;
        TRY(tmp.get(&flat, n));
        TRY(tmp.get(&n_flat, 1));
        TRY_HIP(hipMemsetAsync(n_flat, 0, 4, st));
        std::vector<LevelInfo> hl(kMaxLevels + 2);
        uint32_t height = 0, bad_boxes = 0, launched = 0, flat_count = 0;
        {
            const int root = 0;
            const LevelInfo first{1u, 0u};
            TRY_HIP(hipMemcpyAsync(queue[0], &root, 4, hipMemcpyHostToDevice, st));
            TRY_HIP(hipMemcpyAsync(&lvl[0], &first, sizeof(first), hipMemcpyHostToDevice, st));
        }
        uint64_t width = 1;   // a level of the 4-wide tree has at most min(4^level, n / 2) nodes: small grids for the top levels
        bool recs_made = false;
        for (uint32_t round_levels = 18;; round_levels = 8) {
            for (uint32_t k = 0; k < round_levels && launched < kMaxLevels; ++k, ++launched) {
                const uint32_t grid = (uint32_t)std::min<uint64_t>(grid_for(std::min<uint64_t>(width, n)), 2048);
                hipLaunchKernelGGL(k_emit4, dim3(grid), dim3(kThreads), 0, st, (int)n, launched, queue[launched & 1], queue[(launched & 1) ^ 1], lvl, bn,
                                   box, nodes_tmp, leaf_max);
                width = std::min<uint64_t>(width * 4, n);
            }
            if (!recs_made) {
                hipLaunchKernelGGL(k_make_recs, dim3(grid_for(n)), dim3(kThreads), 0, st, s, sids, recs, flat, n_flat);
                recs_made = true;
            }
            TRY_HIP(hipMemcpyAsync(&flat_count, n_flat, 4, hipMemcpyDeviceToHost, st));
            if (mismatches) TRY_HIP(hipMemcpyAsync(&bad_boxes, mismatches, 4, hipMemcpyDeviceToHost, st));
            TRY_HIP(hipMemcpyAsync(&height, root_height, 4, hipMemcpyDeviceToHost, st));
            TRY_HIP(hipMemcpyAsync(hl.data(), lvl, hl.size() * sizeof(LevelInfo), hipMemcpyDeviceToHost, st));
            TRY_HIP(hipStreamSynchronize(st));
            if (bad_boxes) { cleanup(); return rtx_fail(ctx, RTX_ERR_HIP, "verify_build: " + std::to_string(bad_boxes) + " inner nodes whose box is not the union of their children's"); }
            if (height == 0 || height >= 255) { cleanup(); return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "BVH deeper than the traversal stack allows"); }
            if (hl[launched].count == 0 || launched >= height) break;   // no roots were reserved for a further level
        }
        uint64_t n_nodes = 0, n_leaves = 0;
        for (uint32_t level = 0; level < launched && hl[level].count; ++level) {
            n_nodes += hl[level].count;
            n_leaves += hl[level].leaves;
            levels4 = level + 1;
        }
        if (hl[launched].count != 0 || n_nodes == 0 || n_nodes > (uint64_t)n - 1 || n_leaves == 0) { cleanup(); return rtx_fail(ctx, RTX_ERR_HIP, "BVH emission did not terminate"); }
        BvhNode* nodes = nullptr;
        TRY(dev_alloc(ctx, store->ptrs, &nodes, (size_t)n_nodes));
        TRY_HIP(hipMemcpyAsync(nodes, nodes_tmp, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToDevice, st));
        if (flat_count) {   // rare: see DevScene::always
            PrimRec* always = nullptr;
            TRY(dev_alloc(ctx, store->ptrs, &always, (size_t)flat_count));
            hipLaunchKernelGGL(k_copy_recs, dim3((flat_count + kThreads - 1) / kThreads), dim3(kThreads), 0, st, recs, flat, flat_count, always);
            s.always = always;
            s.n_always = flat_count;
            bs.n_always = flat_count;
        }
        bs.n_nodes = (uint32_t)n_nodes;
        bs.n_leaves = (uint32_t)n_leaves;
        bs.max_depth = height;   // height of the binary tree; the 4-wide tree has about half of it
        s.recs = recs;
        s.nodes = nodes;
        s.n_nodes = (uint32_t)n_nodes;
        s.uses_bvh = 1;
    }
    TRY_HIP(hipMemcpyAsync(&hb, sb, sizeof(hb), hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    {
        float smin[3], smax[3];
        for (int k = 0; k < 3; ++k) {
            smin[k] = (n == 0) ? 0.0f : dec_f32(hb.bmin[k]);
            smax[k] = (n == 0) ? 0.0f : dec_f32(hb.bmax[k]);
            ctx->scene_min[k] = smin[k];
            ctx->scene_max[k] = smax[k];
        }
        // Safe region: a cube around the scene that contains the camera with margin.
        float radius = 0.0f, dcam = 0.0f;
        for (int k = 0; k < 3; ++k) {
            s.safe_c[k] = 0.5f * smin[k] + 0.5f * smax[k];
            radius = std::max(radius, 0.5f * (smax[k] - smin[k]));
            dcam = std::max(dcam, std::fabs(ctx->camera.pos[k] - s.safe_c[k]));
        }
        if (!std::isfinite(radius)) radius = 0.0f;
        if (!std::isfinite(dcam)) dcam = 0.0f;
        s.safe_r = std::max(std::max(4.0f * radius, 2.0f * dcam + 2.0f * radius), 1e-3f);
    }
    {
        const uint32_t need = 3 * (levels4 + 1) + 4;   // levels4: height of the 4-wide tree (0 without a BVH)
        ctx->trav_spill_need = need > RTX_STACK_LDS ? need - RTX_STACK_LDS : 0;
        TRY(rtx_alloc_traversal_scratch(ctx));
    }
    bs.uses_bvh = s.uses_bvh;
    TRY_HIP(hipEventRecord(ev1, st));
    TRY_HIP(hipEventSynchronize(ev1));
    TRY_HIP(hipGetLastError());
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    bs.build_ms = ms;
    cleanup();
#undef TRY
#undef TRY_HIP
    if (s.uses_bvh && bs.max_depth > 200)
        return rtx_fail(ctx, RTX_ERR_UNSUPPORTED, "BVH deeper than the traversal stack allows");
    ctx->build_store = store;
    ctx->build_stats = bs;
    ctx->built = true;
    if (bs.n_always > 64)   // correct, but every ray pays for each of them: say so where a caller who wonders will look
        ctx->error = "note: " + std::to_string(bs.n_always) + " triangles without an area are tested by every ray (rtx_build_stats.n_always)";
    return RTX_OK;
}
