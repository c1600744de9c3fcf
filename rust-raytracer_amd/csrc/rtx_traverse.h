// rtx_traverse.h -- closest-hit / any-hit queries against the scene.
//
// Semantics to reproduce: Scene::intersects (src/scene/mod.rs:181-201), the
// exhaustive loop: the hit with the smallest t wins and, among equal t, the
// lowest primitive index (the loop only replaces on `new_t < t`). The update
// rule used here, `t < best || (t == best && id < best_id)`, gives that result
// for any visiting order, so the BVH may visit primitives in any order as long
// as it never skips one whose own test would have produced the winning hit.
// Box tests are therefore conservative: boxes are padded (see rtx_bvh.hip) and
// the ray-side padding below covers rounding in the primitive tests, which
// grows with the magnitude of the ray origin.
//
// Three differences are documented rather than reproduced. A sphere test that returns Hit(-0.0) (a quotient that
// underflows, sphere.rs:82,99-101; the loop's `new_t < t` treats it as 0.0) is reported as Hit(+0.0): the same number,
// ordered the same way against every other hit, but with ONE bit pattern, so that the order-independent rule can be
// evaluated on the bits (the 64-bit minimum of rtx_coop.h) as well as on the values. A primitive test that
// returns Hit(NaN) (only possible for rays with non-finite components,
// sphere.rs:70-102) is ignored here, while the reference would keep it if it
// came first. And a ray that lies IN the plane of a triangle -- direction and
// origin both within f32 rounding of it, about 1e-7 relative -- turns the
// determinant of poly.rs:149-153 into rounding noise, so the reference's test
// may report a hit at an arbitrary t, far from the triangle; no box can promise
// to contain that, and the tree may return the hit the ray really has instead
// (tools/fuzz_bvh.py counts such rays apart: a few per hundred thousand rays aimed
// along edges and through planes on purpose, none among rays in general position).
// What decides is the noise of the barycentric coordinates, u_f32 |s| / (|e| sin(angle at v0) cos(ray, normal)) with
// s = origin - v0: of order one for in-plane rays at an ordinary triangle, and already for rays a milliradian off the
// plane of a needle seen from fifty times its length away (the test then reports hits several lengths beside it).
// Triangles without an area are the same case for EVERY ray; they are kept out of
// the boxes' hands and tested by all rays (DevScene::always). Needles -- edges at v0
// between 1e-3 and 0.5 rad apart -- report a t that is off by far more than an ordinary
// triangle's; their boxes are padded in proportion (sliver_pad, rtx_bvh.hip).
#pragma once
#include "rtx_internal.h"

#define RTX_BLOCK 256
// Workgroups of 256 threads (= waves per SIMD) that a traversal kernel keeps resident on a CU. The traversal loop is
// a chain of dependent steps per ray, so a further resident workgroup is worth more than what it costs in registers
// or LDS -- as long as nothing spills to scratch in the loop (measured per kernel: the figures below; one step up
// spills 28-120 bytes per lane and is slower, except where noted). The kernels are compiled for exactly this many
// waves, their LDS (stacks 16 KiB + shared slots 3 KiB + the copy of the tree's top levels) is sized to fit it, and
// launches ask for no more workgroups per CU than that (more would queue behind the others and end the launch late).
//   closest hit:           triangle-only 74 registers, with spheres 76-77 -> 6 (80 registers is the limit for six)
//   any-hit shadow rays:   without area lights 71-72 (triangle-only / with spheres: the compiler fits the latter's 76-79 into the
//                          72 that seven need, without scratch) -> 7; the seventh workgroup is only used by launches long
//                          enough to feed it, see trav_blocks: +3.4 % on a launch of eight 1080p frames, +3 % on the 4 M-ray
//                          launches of the 64 M-primitive scene, -3 % on one 1080p frame's launch; with area lights 73-79 -> 6
//   closest-hit shadow rays (transparent materials): without area lights 86-88 registers, held to 80 with 32-40 bytes of scratch -> 6
//                          (the exception to the rule above: -2.5 to -3 % on the launch, measured on 16 M triangles with glass spheres
//                          and on the bench scene with four); with area lights 93-96 -> 5
// (tools/kernel_resources.sh prints the compiler's figures: no other kernel uses scratch.)
constexpr int closest_waves(bool sph) { return 6; }
constexpr int shadow_waves(bool any, bool area, bool sph) { return any ? (!area ? 7 : 6) : (area ? 5 : 6); }
#define RTX_WAVES_ATTR(n) __attribute__((amdgpu_waves_per_eu(n, n)))
// Which kernels a scene runs on: the triangle-only ones unless it holds spheres or triangles that every ray has to test
// (DevScene::always; rare enough to share the general kernels, compiled once more with that loop).
RTX_HD bool general_kernels(const DevScene& s) { return s.n_spheres != 0 || s.n_always != 0; }

RTX_D uint32_t f2u(float f) { return __float_as_uint(f); }
RTX_D float u2f(uint32_t u) { return __uint_as_float(u); }

// The primitive test of one record (already in registers); t is valid when true is returned.
// SPH = false: the scene has no sphere, so the record is a triangle (traversal kernels are compiled for both cases:
// a triangle-only scene pays neither the kind test nor, in the node step, the sphere widening).
template <bool SPH = true>
RTX_D bool record_hit(float4 q0, float4 q1, float4 q2, V3 o, V3 d, float& t) {
    if (!SPH || f2u(q1.w) == 0u) return tri_intersect(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), &t);   // t > 1e-7 on a hit
    const bool hit = sphere_intersect(o, d, v3(q0.x, q0.y, q0.z), q1.x, &t);
    t += 0.0f;   // -0.0 -> +0.0 (a sphere hit is >= 0 by value; the sign of a zero must not order it after every positive t)
    return hit;
}
// One candidate primitive against one ray; updates (best_t, best_id).
template <bool SPH = true>
RTX_D void test_quads(float4 q0, float4 q1, float4 q2, V3 o, V3 d, float& best_t, int& best_id) {
    const int id = (int)f2u(q0.w);
    float t;
    if (record_hit<SPH>(q0, q1, q2, o, d, t) && (t < best_t || (t == best_t && id < best_id))) {
        best_t = t;
        best_id = id;
    }
}
template <bool SPH = true>
RTX_D void test_record(const PrimRec* __restrict__ recs, uint32_t idx, V3 o, V3 d, float& best_t, int& best_id) {
    test_quads<SPH>(recs[idx].q0, recs[idx].q1, recs[idx].q2, o, d, best_t, best_id);
}

// The loop of src/scene/mod.rs:186-199 over every record.
RTX_D void trace_exhaustive(const DevScene& s, V3 o, V3 d, float& best_t, int& best_id) {
    best_t = __builtin_inff();
    best_id = 0x7fffffff;
    for (uint32_t i = 0; i < s.n_prims; ++i) test_record(s.recs, i, o, d, best_t, best_id);
    if (best_id == 0x7fffffff) best_id = -1;
}

// Per-ray constants of the slab test: (b - (o +- pad)) * inv is evaluated as
// fma(b, inv, -(o +- pad) * inv), one operation per plane. The fused form differs
// from the unfused one by a rounding of the box coordinate, far inside the padding;
// only box tests are fused, primitive tests keep the reference's operation order.
struct RayBox {
    V3 on, of;   // offsets of the near and far planes: (o +- pad) * inv with the sign that widens the box
    V3 inv;
};
RTX_D float safe_inv(float d) {
    return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : 1.0f / d;
}
// The plane of a box a ray enters first is its low one where the direction is positive and
// its high one where it is negative, so near and far are picked by sign, not by min / max:
// with low <= high and pad >= 0 both give the same numbers (rounding is monotone).
RTX_D RayBox make_raybox(const DevScene& s, V3 o, V3 d) {
    const float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
    const float pad = s.pad_rel * m;
    RayBox rb;
    rb.inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    const V3 om = v3((o.x + pad) * rb.inv.x, (o.y + pad) * rb.inv.y, (o.z + pad) * rb.inv.z);   // goes with a box's low planes
    const V3 op = v3((o.x - pad) * rb.inv.x, (o.y - pad) * rb.inv.y, (o.z - pad) * rb.inv.z);   // goes with its high planes
    rb.on = v3(rb.inv.x < 0.0f ? op.x : om.x, rb.inv.y < 0.0f ? op.y : om.y, rb.inv.z < 0.0f ? op.z : om.z);
    rb.of = v3(rb.inv.x < 0.0f ? om.x : op.x, rb.inv.y < 0.0f ? om.y : op.y, rb.inv.z < 0.0f ? om.z : op.z);
    return rb;
}

// Diagnostics: one record per wavefront slot of a traversal launch (see rtx_debug_stamps).
struct WaveStamp {
    unsigned long long* rec;
    unsigned long long units;
    unsigned long long step_sum = 0, step_max = 0;   // diagnostic build (RTX_COUNT_STEPS) only
    unsigned long long prof = 0;                      // diagnostic build: (t_mem, t_alu, t_don) >> 8 in 21 bits each
    // Adds one unit's per-lane step counts: sum over lanes and the slowest lane.
    RTX_D void add_steps(uint32_t lane_steps) {
        uint32_t sum = lane_steps, mx = lane_steps;
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o);
            mx = max(mx, (uint32_t)__shfl_xor(mx, o));
        }
        step_sum += sum;
        step_max += mx;
    }
    RTX_D void begin(const DevScene& s, int kind) {
        rec = nullptr;
        units = 0;
        if (s.stamps && (threadIdx.x & 63) == 0) {
            rec = s.stamps + ((size_t)kind * (s.spill_stride / 64) + (blockIdx.x * (RTX_BLOCK / 64) + (threadIdx.x >> 6))) * 4;
            rec[0] = wall_clock64();
        }
    }
    RTX_D void end() {
        if (rec) {
            uint32_t xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            rec[1] = wall_clock64();
            rec[2] = units | (step_max << 16) | (step_sum << 40);
            rec[3] = (xcc & 0xfu) | (((hw >> 13) & 0x7u) << 8) | (((hw >> 8) & 0xfu) << 16);
#ifdef RTX_COUNT_STEPS
            rec[3] = prof;   // diagnostic build: cycle split instead of the placement
#endif
        }
    }
};

RTX_D bool in_safe_region(const DevScene& s, V3 o) {
    return fabsf(o.x - s.safe_c[0]) <= s.safe_r && fabsf(o.y - s.safe_c[1]) <= s.safe_r &&
           fabsf(o.z - s.safe_c[2]) <= s.safe_r;  // false for NaN origins
}

// |d|^2 within 2^-40 .. 2^40 (false for NaN): see coop_trace.
RTX_D bool dir_in_range(V3 d) {
    const float a = vdot(d, d);
    return a >= 0x1p-40f && a <= 0x1p40f;
}

// Per-lane traversal stack. The first RTX_STACK_LDS entries live in LDS: entry k of
// lane l at lds[k * RTX_BLOCK + l], so the lanes of a wavefront always touch 64
// consecutive dwords (conflict-free whatever their individual depths are). Deeper
// entries, which near-first traversal rarely needs, spill to a global array laid out
// the same way ([entry][thread slot]); keeping the LDS part at 16 KiB per workgroup is
// what lets 8 workgroups (32 wavefronts) share a CU. The spill area has
// DevScene::spill_stride thread slots, so traversal grids are bounded by it.
#define RTX_STACK_LDS 16
// Pointers into LDS carry their address space, so stack accesses compile to ds_read / ds_write
// (a generic pointer would turn them into flat accesses that also count as vector-memory
// operations and force the traversal to wait for its node fetch before touching the stack).
#define RTX_LDS_PTR(T) __attribute__((address_space(3))) T*
struct TravStack {
    RTX_LDS_PTR(int) base;
    int* spill;
    uint32_t spill_stride;
#ifdef RTX_COUNT_STEPS
    uint32_t steps = 0;   // diagnostic build: node steps of this lane
    uint32_t leaves = 0;  // diagnostic build: primitive tests of this lane
    unsigned long long t_mem = 0, t_alu = 0, t_don = 0;   // diagnostic build: shader cycles waiting for the fetch / computing / giving work
    unsigned long long t_deq = 0, t_setup = 0;            // diagnostic build: shader cycles taking a work unit / setting up its rays
#endif
    RTX_D void set(int* lds, const DevScene& s) {
        base = (RTX_LDS_PTR(int))lds + threadIdx.x;
        spill = s.spill + (blockIdx.x * RTX_BLOCK + threadIdx.x);
        spill_stride = s.spill_stride;
    }
    RTX_D void put(int sp, int v) {
        if (sp < RTX_STACK_LDS) base[sp * RTX_BLOCK] = v;
        else spill[(size_t)(sp - RTX_STACK_LDS) * spill_stride] = v;
    }
    RTX_D int get(int sp) const {
        if (sp < RTX_STACK_LDS) return base[sp * RTX_BLOCK];
        return spill[(size_t)(sp - RTX_STACK_LDS) * spill_stride];
    }
};

// Sphere tests accept rays that miss: the discriminant b*b - 4ac (sphere.rs:62-67) is a difference of two numbers near
// 4|d|^2 D^2 (D = distance from the ray origin to the centre), each carrying a few roundings, so a ray passing within
// sqrt(r^2 + K D^2) of the centre can still report a hit. K follows from the reference's operation order, u = 2^-24,
// o = fl(ori - centre) taken as the (slightly moved, by <= u D per axis) exact offset, D = |o|, a = |d|^2:
//   fl(d.o): three products, two sums: error <= 3u |d| D, so b = 2 fl(d.o) is off by <= 6u |d| D and, with |b| <= 2|d|D,
//            fl(b*b) by <= 2 |b| 6u|d|D + u b^2 <= 28u a D^2;
//   c = fl(fl(o.o) - fl(r*r)): <= 3u D^2 + u r^2 + u (D^2 - r^2) = 4u D^2;  a = fl(d.d): relative 3u;
//   fl(fl(4a) c): <= 4a (4u c + 4u D^2) <= 32u a D^2 (c <= D^2);  the final subtraction keeps the sign.
// Exactly, b*b - 4ac = 4a (r^2 - p^2), p = distance of the centre from the ray's line: a non-negative computed value
// needs p^2 <= r^2 + 15u D^2. K = 15.5 * 2^-24 leaves room for the second-order terms and the moved offset; boxes are
// padded, quantised outwards and the margin below is scaled by 1.002 on top. (|d|^2 and the products must stay in the
// normal range of f32 for this: see dir_in_range.)
// The excess over r depends on the ray, so it is not baked into the boxes: a node with a sphere somewhere below it is
// flagged, and the traversal widens ALL its children by
//     e = sqrt(K) Dn  >=  sqrt(r^2 + K Dn^2) - r   for every r > 0,     Dn = distance to the node's farthest corner,
// one square root per node step (a per-child margin from the smallest radius below each child -- still stored in the node --
// cost a square root and a reciprocal per child, and a margin masked per child by the node's child_spheres bits cost
// twenty registers: both are paid in resident wavefronts; the per-radius margin only tightens for spheres with
// r^2 >~ K D^2, and for those it is small against r either way). Widening a child without spheres is always allowed
// (a box may be larger): in a tree that mixes the kinds it happens to the triangles that share a node with a sphere,
// and to the upper levels, where every node has a sphere somewhere below it; low nodes with triangles only -- most of
// a mesh -- are not flagged and skip the arithmetic.
#ifndef RTX_SPHERE_K
#define RTX_SPHERE_K 9.2387e-7f   // 15.5 * 2^-24
#endif
constexpr float kSphereK = RTX_SPHERE_K;
constexpr float kSphereSqrtK = 9.6119e-4f * 1.002f;   // sqrt(K), rounded up, with the margin

// wn* / wf*: the four children's near and far plane bytes of one axis; ln* / lf*: the matching offsets.
RTX_D void node4_slabs(uint32_t wnx, uint32_t wny, uint32_t wnz, uint32_t wfx, uint32_t wfy, uint32_t wfz, uint32_t nchild,
                       float limit, float sx, float sy, float sz, float lnx, float lny, float lnz, float lfx,
                       float lfy, float lfz, float* t, int* hits = nullptr) {
    const float inf = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sh = 8 * k;
        float nx = __builtin_fmaf((float)((wnx >> sh) & 0xffu), sx, lnx), fx = __builtin_fmaf((float)((wfx >> sh) & 0xffu), sx, lfx);
        float ny = __builtin_fmaf((float)((wny >> sh) & 0xffu), sy, lny), fy = __builtin_fmaf((float)((wfy >> sh) & 0xffu), sy, lfy);
        float nz = __builtin_fmaf((float)((wnz >> sh) & 0xffu), sz, lnz), fz = __builtin_fmaf((float)((wfz >> sh) & 0xffu), sz, lfz);
        // [lo, hi] = the ray's parameter range inside the box, cut to [0, limit]: one comparison decides
        const float lo = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);
        const float hi = fminf(fminf(fminf(fx, fy), fz), limit);
        const bool hit = ((uint32_t)k < nchild) & (lo <= hi);   // bitwise: no branches
        t[k] = hit ? lo : inf;
        if (hits) hits[k] = hit;
    }
}

// One step through a 4-wide node (layout: rtx_internal.h). The child boxes are
// 8-bit offsets from the node's origin in units of a per-axis power of two, so the
// slab test runs directly on the quantised values:
//     t = fma(q, scale * inv, fma(origin, inv, -(o +- pad) * inv)).
// The quantised boxes contain the padded boxes they were built from, so the test
// stays conservative. Children that are hit are ordered near to far; the nearest
// becomes the next node and the others are pushed, farthest first.
// Returns the next reference, or `done` when nothing is hit and the stack is empty.
// The lane's stack holds entries [bot, sp): bot > 0 once the oldest entries were given away (rtx_coop.h).
// SORT = false (any-hit queries): the order of the children does not matter, they are taken as stored.
template <bool SORT = true, bool SPH = true>
RTX_D int node4_step(const uint4 A, const uint4 B, const uint4 C, const uint4 D, const RayBox& rb, V3 o, float limit,
                     TravStack& st, int& sp, int bot, int done) {
#ifdef RTX_COUNT_STEPS
    ++st.steps;
#endif
    const uint32_t meta = A.w;
    const float stx = u2f((meta & 0xffu) << 23), sty = u2f(((meta >> 8) & 0xffu) << 23), stz = u2f(((meta >> 16) & 0xffu) << 23);
    const float sx = stx * rb.inv.x, sy = sty * rb.inv.y, sz = stz * rb.inv.z;
    const uint32_t nchild = (meta >> 24) & 7u;
    const float ox = u2f(A.x), oy = u2f(A.y), oz = u2f(A.z);
    const float lnx = __builtin_fmaf(ox, rb.inv.x, -rb.on.x), lfx = __builtin_fmaf(ox, rb.inv.x, -rb.of.x);
    const float lny = __builtin_fmaf(oy, rb.inv.y, -rb.on.y), lfy = __builtin_fmaf(oy, rb.inv.y, -rb.of.y);
    const float lnz = __builtin_fmaf(oz, rb.inv.z, -rb.on.z), lfz = __builtin_fmaf(oz, rb.inv.z, -rb.of.z);
    // low-plane bytes: B.x B.y B.z, high-plane bytes: B.w C.x C.y
    const bool ngx = rb.inv.x < 0.0f, ngy = rb.inv.y < 0.0f, ngz = rb.inv.z < 0.0f;
    const uint32_t wnx = ngx ? B.w : B.x, wfx = ngx ? B.x : B.w;
    const uint32_t wny = ngy ? C.x : B.y, wfy = ngy ? B.y : C.x;
    const uint32_t wnz = ngz ? C.y : B.z, wfz = ngz ? B.z : C.y;
    const float inf = __builtin_inff();
    float t[4];
    int hk[4] = {0, 0, 0, 0};   // SORT = false: which children the ray enters (a range that starts at +inf counts: visiting more is always safe)
    int r[4] = {(int)C.z, (int)C.w, (int)D.x, (int)D.y};
    // spheres below: widen the children (see kSphereK). The branch is taken by the whole wavefront or not at all
    // (a vote), so that wavefronts away from any sphere skip the widening arithmetic instead of computing it masked.
    float lnx_ = lnx, lny_ = lny, lnz_ = lnz, lfx_ = lfx, lfy_ = lfy, lfz_ = lfz;
    if (SPH && __ballot((meta >> 31) != 0u) != 0ull) {
        const float dx = fmaxf(fabsf(ox - o.x), fabsf(__builtin_fmaf(255.0f, stx, ox) - o.x));
        const float dy = fmaxf(fabsf(oy - o.y), fabsf(__builtin_fmaf(255.0f, sty, oy) - o.y));
        const float dz = fmaxf(fabsf(oz - o.z), fabsf(__builtin_fmaf(255.0f, stz, oz) - o.z));
        const float dn2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
        // v_sqrt_f32 is good to 1 ulp; the 1.002 in kSphereSqrtK covers it
        const float e = ((meta >> 31) != 0u) ? kSphereSqrtK * __builtin_amdgcn_sqrtf(dn2) : 0.0f;
        // planes move outward by e: e * |1/d| in ray parameter
        const float wx = e * fabsf(rb.inv.x), wy = e * fabsf(rb.inv.y), wz = e * fabsf(rb.inv.z);
        lnx_ -= wx; lny_ -= wy; lnz_ -= wz;
        lfx_ += wx; lfy_ += wy; lfz_ += wz;
    }
    node4_slabs(wnx, wny, wnz, wfx, wfy, wfz, nchild, limit, sx, sy, sz, lnx_, lny_, lnz_, lfx_, lfy_, lfz_, t, SORT ? nullptr : hk);
#define RTX_CE(i, j)                                   \
    do {                                               \
        const bool sw = t[j] < t[i];                   \
        const float ta = sw ? t[j] : t[i], tb = sw ? t[i] : t[j]; \
        const int ra = sw ? r[j] : r[i], rb_ = sw ? r[i] : r[j];  \
        t[i] = ta; t[j] = tb; r[i] = ra; r[j] = rb_;   \
    } while (0)
    // Up to three children go to the stack. Their slots follow from the hit flags alone, so the common case -- the
    // entries fit the LDS part of the stack -- is three predicated LDS writes to computed slots, with no test per push.
    if (SORT) {
        RTX_CE(0, 1); RTX_CE(2, 3); RTX_CE(0, 2); RTX_CE(1, 3); RTX_CE(1, 2);
        const int h3 = t[3] < inf, h2 = t[2] < inf, h1 = t[1] < inf;   // sorted: a hit at k implies hits below k
        if (sp + 3 <= RTX_STACK_LDS) {
            if (h3) st.base[sp * RTX_BLOCK] = r[3];
            if (h2) st.base[(sp + h3) * RTX_BLOCK] = r[2];
            if (h1) st.base[(sp + h3 + h2) * RTX_BLOCK] = r[1];
            sp += h3 + h2 + h1;
        } else {
            if (h3) st.put(sp++, r[3]);
            if (h2) st.put(sp++, r[2]);
            if (h1) st.put(sp++, r[1]);
        }
        if (t[0] < inf) return r[0];
    } else {
        // any order will do: the lowest hit child is next, the others are pushed from the highest down
        const int h0 = hk[0], h1 = hk[1], h2 = hk[2], h3 = hk[3];
        const int p3 = h3 & (h0 | h1 | h2), p2 = h2 & (h0 | h1), p1 = h1 & h0;   // pushed: hit, and not the lowest hit
        if (sp + 3 <= RTX_STACK_LDS) {
            if (p3) st.base[sp * RTX_BLOCK] = r[3];
            if (p2) st.base[(sp + p3) * RTX_BLOCK] = r[2];
            if (p1) st.base[(sp + p3 + p2) * RTX_BLOCK] = r[1];
            sp += p3 + p2 + p1;
        } else {
            if (p3) st.put(sp++, r[3]);
            if (p2) st.put(sp++, r[2]);
            if (p1) st.put(sp++, r[1]);
        }
        if (h0 | h1 | h2 | h3) return h0 ? r[0] : h1 ? r[1] : h2 ? r[2] : r[3];
    }
#undef RTX_CE
    if (sp == bot) return done;
    return st.get(--sp);
}


constexpr int kRefDone = 0x7fffffff;   // "no more work" reference (never a node index: fewer than 2^28 nodes exist)

// What a traversal reference points at, fetched with ONE set of loads whatever its kind:
// 64 bytes at nodes[ref] (inner node) or at recs[first] (leaf: the record's 48 bytes plus
// 16 that are ignored; the record array is allocated one record longer than it is used).
// A wavefront whose lanes sit on a mix of inner nodes and leaves then waits for memory once
// per iteration, not once per kind.
struct Fetched {
    uint4 a, b, c, d;
};
// The first top_nodes(waves) nodes of the tree -- its top levels: nodes are numbered in level order -- are copied into
// LDS by every traversal workgroup. Every ray walks through them, and a wavefront's 64 divergent 64-byte fetches
// occupy the CU's vector-memory path for ~125 cycles even when they all hit in L1 (four dwordx4 instructions, ~31
// cycles each); the same fetch from LDS is four ds_read_b128 (~16 cycles) with a fraction of the latency.
// How many: as many as the LDS left by the kernel's resident workgroups holds (160 KiB per CU; a workgroup's stacks
// and shared slots take 19 KiB): 4 workgroups -> 320 nodes (20 KiB, levels 0-4 of a full tree), 5 -> 128, 6 -> 64
// (levels 0-2 and most of level 3), 7 -> 48. The cache is worth 1-2 %, a resident workgroup 3-12 %.
constexpr uint32_t top_nodes(int waves) { return waves <= 4 ? 320u : waves == 5 ? 128u : waves == 6 ? 64u : 48u; }
typedef unsigned int rtx_u32x4 __attribute__((ext_vector_type(4)));
struct TopCache {
    RTX_LDS_PTR(rtx_u32x4) base;
    uint32_t count;   // nodes cached (0: none, e.g. no tree)
    // All threads of the workgroup call this before the first traversal step; `lds` holds `capacity` * 4 quads.
    RTX_D void load(uint4* lds, const DevScene& s, uint32_t capacity) {
        base = (RTX_LDS_PTR(rtx_u32x4))lds;
        count = s.uses_bvh ? min(s.n_nodes, capacity) : 0u;
        const uint4* src = reinterpret_cast<const uint4*>(s.nodes);
        for (uint32_t k = threadIdx.x; k < count * 4u; k += RTX_BLOCK) lds[k] = src[k];
        __syncthreads();
    }
};
RTX_D Fetched fetch_ref(const DevScene& s, const TopCache& tc, int cur) {
    if ((uint32_t)cur < tc.count) {   // an inner node of the cached top levels (leaf references are negative)
        RTX_LDS_PTR(rtx_u32x4) q = tc.base + (uint32_t)cur * 4u;
        const rtx_u32x4 a = q[0], b = q[1], c = q[2], d = q[3];
        Fetched f;
        f.a = make_uint4(a.x, a.y, a.z, a.w);
        f.b = make_uint4(b.x, b.y, b.z, b.w);
        f.c = make_uint4(c.x, c.y, c.z, c.w);
        f.d = make_uint4(d.x, d.y, d.z, d.w);
        return f;
    }
    const uint4* p = (cur >= 0) ? reinterpret_cast<const uint4*>(s.nodes + cur)
                                : reinterpret_cast<const uint4*>(s.recs + (~(uint32_t)cur >> 3));
    // Four 16-byte loads issued back to back and one wait, spelled out: left to itself the
    // compiler splits the 64 bytes by field use and sinks the parts only the node path reads
    // behind the branch, which adds load instructions (each costs the CU's address unit one
    // access per lane) and a second memory wait to every node step.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 a, b, c, d;
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\t"
        "global_load_dwordx4 %1, %4, off offset:16\n\t"
        "global_load_dwordx4 %2, %4, off offset:32\n\t"
        "global_load_dwordx4 %3, %4, off offset:48\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p)
        : "memory");
    Fetched f;
    f.a = make_uint4(a.x, a.y, a.z, a.w);
    f.b = make_uint4(b.x, b.y, b.z, b.w);
    f.c = make_uint4(c.x, c.y, c.z, c.w);
    f.d = make_uint4(d.x, d.y, d.z, d.w);
    return f;
}
RTX_D float4 as_f4(uint4 u) { return make_float4(u2f(u.x), u2f(u.y), u2f(u.z), u2f(u.w)); }

// Shadow-ray occlusion for scenes without transparent materials. The reference
// (src/lib.rs:117-135) takes the closest hit and reports "lit" iff it is farther
// from the ray origin than the light (`ori.distance(hit) > ori.distance(light)`),
// or always "shadowed" for a directional light. vdistance(o, o + d*t) is monotone
// in t under f32 rounding, so "the closest hit is nearer than the light" equals
// "some hit is nearer than the light": the search can stop at the first such hit.
// dist_light < 0 encodes a directional light (every hit occludes).
RTX_D bool occluding_hit(V3 o, V3 d, float t, float dist_light) {
    if (dist_light < 0.0f) return true;
    const V3 p = o + vmult(d, t);                 // Intersection::point, intersection.rs:26-28
    return !(vdistance(o, p) > dist_light);       // lib.rs:123-127
}
template <bool SPH = true>
RTX_D bool any_quads(float4 q0, float4 q1, float4 q2, V3 o, V3 d, float dist_light) {
    float t;
    return record_hit<SPH>(q0, q1, q2, o, d, t) && occluding_hit(o, d, t, dist_light);
}
template <bool SPH = true>
RTX_D bool any_record(const PrimRec* __restrict__ recs, uint32_t idx, V3 o, V3 d, float dist_light) {
    return any_quads<SPH>(recs[idx].q0, recs[idx].q1, recs[idx].q2, o, d, dist_light);
}
