// rtx_coop.h -- wavefront-cooperative traversal: the lanes of a wavefront share the
// work of their slowest rays.
//
// One ray per lane is the fast way through the bulk of a launch, but a launch ends with
// its slowest wavefront, and a wavefront's 64 rays end with their slowest ray: a ray
// that grazes the surface visits ten times the nodes of its neighbours, one dependent
// memory access after the other, while the other lanes (and, at the end of a launch,
// most of the device) wait. A traversal stack, however, is a list of independent
// subtrees. So the query runs in two phases:
//   1. every lane traverses for its own ray, results in registers, until fewer than `threshold` lanes of the wavefront still have work;
//   2. the lanes publish their results to a per-ray slot in LDS and idle lanes take over
//      subtrees from busy lanes: a lane with more than one pending entry gives its OLDEST
//      stack entry (the largest subtree) to an idle lane, together with the ray (ds_bpermute).
//      The helper traverses that subtree for the owner's ray; hits go to the owner's slot
//      with a 64-bit LDS atomic min on (t bits, primitive id), which is exactly the
//      order-independent rule `t < best || (t == best && id < best_id)`; every lane reads
//      its ray's slot before each step, so all helpers cull with the best bound found so far.
// Which lane tests a primitive does not change the primitive's t, and the minimum does not
// depend on the order of its operands, so the result is the one the exhaustive loop gives.
// Any-hit queries share a flag instead of a key: the first occluder found by any lane ends
// the work of all lanes on that ray.
#pragma once
#include "rtx_traverse.h"

#ifdef RTX_DIAG
// Diagnostic build: lane occupancy of the traversal loop, summed over all wavefronts (index 0: closest hit, 1: any-hit).
// [0] phase-1 iterations, [1] lanes with work in them, [2] of those on inner nodes, [3] iterations with both kinds,
// [4] phase-2 iterations, [5] lanes with work in them, [6] of those on inner nodes, [7] iterations with both kinds,
// [8] donation rounds, [9] queries (wavefront calls), [10] subtrees given
// [12..16] queries by phase-2 length (0, 1-4, 5-16, 17-64, more), [17..21] phase-2 iterations in those classes
static __device__ unsigned long long g_diag[2][32];
#define RTX_DIAG_ADD(k, v) do { dg[k] += (uint32_t)(v); } while (0)
#define RTX_DIAG_FLUSH() do { if ((threadIdx.x & 63) == 0) { const int c_ = dg[4] == 0 ? 0 : dg[4] <= 4 ? 1 : dg[4] <= 16 ? 2 : dg[4] <= 64 ? 3 : 4; dg[12 + c_] = 1; dg[17 + c_] = dg[4]; for (int k_ = 0; k_ < 22; ++k_) if (dg[k_]) atomicAdd(&g_diag[ANY ? 1 : 0][k_], (unsigned long long)dg[k_]); } } while (0)
#else
#define RTX_DIAG_ADD(k, v) do { } while (0)
#define RTX_DIAG_FLUSH() do { } while (0)
#endif

struct CoopLds {
    unsigned long long slot[RTX_BLOCK];   // per lane's ray: closest: (t bits << 32) | id; any-hit: 0 = occluded, 1 = not
    int pair[RTX_BLOCK];                  // per wavefront: donor lane by rank
};

#ifndef RTX_DONATE_MIN
#define RTX_DONATE_MIN 16
#endif
constexpr int kDonateMin = RTX_DONATE_MIN;   // idle lanes that make a round of giving worth its cost (coop_trace, phase 2)
constexpr unsigned long long kCoopMiss = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;   // t = +inf, no primitive

// All 64 lanes of a wavefront must call this together (`active` = the lane has a ray).
// Closest (ANY = false): out_id = primitive id or -1, out_t = its t.
// Any-hit (ANY = true):  out_id = 1 when some primitive occludes the light at distance dist_light
//                        (dist_light < 0: any hit occludes), else 0.
// SPH: the scene holds spheres (false: triangle-only code, see record_hit).
// ALWAYS: the scene holds triangles that every ray has to test (DevScene::always); kernels without them do not carry the loop.
template <bool ANY, bool SPH = true, bool ALWAYS = false>
RTX_D void coop_trace(const DevScene& s, const TopCache& tc, bool active, V3 o, V3 d, float dist_light, TravStack& st, CoopLds& L, int threshold,
                      float& out_t, int& out_id) {
    const int lane = threadIdx.x & 63;
    float best_t = __builtin_inff();
    int best_id = ANY ? 0 : 0x7fffffff;
    int cur = kRefDone, sp = 0;
    if (active) {
        // every primitive, as the reference loop does: without a tree, for origins outside the region the sphere margin
        // is proven for, and -- scenes with spheres -- for directions so short or long that |d|^2 or the products of
        // sphere.rs:59-67 leave the normal range of f32 (an underflowing |d|^2 makes the reference's sphere test report
        // hits for spheres nowhere near the ray; no box can promise those)
        if (!s.uses_bvh || !in_safe_region(s, o) || (SPH && !dir_in_range(d))) {
            if (ANY) {
                for (uint32_t i = 0; i < s.n_prims; ++i)
                    if (any_record<SPH>(s.recs, i, o, d, dist_light)) { best_id = 1; break; }
            } else {
                for (uint32_t i = 0; i < s.n_prims; ++i) test_record<SPH>(s.recs, i, o, d, best_t, best_id);
            }
        } else {
            cur = 0;
            // primitives whose test answers to rounding noise or to overflow rather than to where the ray goes (DevScene::always): usually none
            for (uint32_t i = 0; ALWAYS && i < s.n_always; ++i) {
                if (ANY) {
                    if (any_record<SPH>(s.always, i, o, d, dist_light)) { best_id = 1; cur = kRefDone; break; }
                } else {
                    test_record<SPH>(s.always, i, o, d, best_t, best_id);
                }
            }
        }
    }
    // any-hit: boxes beyond the light never hold an occluder (vdistance(o, o + d*t) carries an absolute rounding error of a few ulp of |o|)
    float t_limit = __builtin_inff();
    if (ANY && dist_light >= 0.0f) {
        const float len = vlength(d);
        const float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
        if (len > 0.0f) t_limit = ((dist_light + 4e-6f * m) / len) * 1.0001f + 1e-6f;
    }
    RayBox rb = make_raybox(s, o, d);
#ifdef RTX_DIAG
    uint32_t dg[22] = {};
#endif
    RTX_DIAG_ADD(9, 1);

    // ---- phase 1: one ray per lane ----
    uint64_t busy;
#ifdef RTX_COOP_ONLY
    busy = __ballot(cur != kRefDone);
#else
    for (;;) {
        busy = __ballot(cur != kRefDone);
        if (busy == 0 || (int)__popcll(busy) < threshold) break;
#ifdef RTX_DIAG
        {
            const uint64_t nodes = __ballot(cur != kRefDone && cur >= 0), leaves = __ballot(cur < 0);
            RTX_DIAG_ADD(0, 1); RTX_DIAG_ADD(1, __popcll(busy)); RTX_DIAG_ADD(2, __popcll(nodes)); RTX_DIAG_ADD(3, (nodes && leaves) ? 1 : 0);
        }
#endif
#ifdef RTX_COUNT_STEPS
        const unsigned long long c0 = __builtin_readcyclecounter();
#endif
        if (cur != kRefDone) {
            const Fetched f = fetch_ref(s, tc, cur);
#ifdef RTX_COUNT_STEPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long c1 = __builtin_readcyclecounter();
            st.t_mem += c1 - c0;
#endif
            if (cur >= 0) {
                cur = node4_step<!ANY, SPH>(f.a, f.b, f.c, f.d, rb, o, ANY ? t_limit : best_t, st, sp, 0, kRefDone);
            } else {
                // a leaf is one primitive, and its record is what was fetched
                if (ANY) {
                    if (any_quads<SPH>(as_f4(f.a), as_f4(f.b), as_f4(f.c), o, d, dist_light)) {
                        best_id = 1;
                        sp = 0;
                    }
                } else {
                    test_quads<SPH>(as_f4(f.a), as_f4(f.b), as_f4(f.c), o, d, best_t, best_id);
                }
#ifdef RTX_COUNT_STEPS
                ++st.leaves;
#endif
                cur = (sp == 0) ? kRefDone : st.get(--sp);
            }
#ifdef RTX_COUNT_STEPS
            st.t_alu += __builtin_readcyclecounter() - c1;
#endif
        }
    }
#endif
    if (busy == 0) {   // wave-uniform
        RTX_DIAG_FLUSH();
        out_t = best_t;
        out_id = ANY ? best_id : (best_id == 0x7fffffff ? -1 : best_id);
        return;
    }

    // ---- phase 2: idle lanes take subtrees from busy lanes ----
    RTX_LDS_PTR(unsigned long long) slot = (RTX_LDS_PTR(unsigned long long))L.slot + (threadIdx.x & ~63u);   // this wavefront's 64 slots
    RTX_LDS_PTR(int) pair = (RTX_LDS_PTR(int))L.pair + (threadIdx.x & ~63u);
    // LDS operations of one wavefront execute in order; wavefront-scope atomics keep the compiler from caching them
    __hip_atomic_store(&slot[lane], ANY ? (best_id ? 0ull : 1ull) : (((unsigned long long)f2u(best_t) << 32) | (uint32_t)best_id),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    int owner = lane, bot = 0;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (;;) {
#ifdef RTX_COUNT_STEPS
        const unsigned long long d0 = __builtin_readcyclecounter();
#endif
        const bool has = cur != kRefDone;
        busy = __ballot(has);
        if (busy == 0) break;
        const bool can_give = has && sp > bot;
        const uint64_t don = __ballot(can_give);
#ifdef RTX_DIAG
        {
            const uint64_t nodes = __ballot(has && cur >= 0), leaves = __ballot(cur < 0);
            RTX_DIAG_ADD(4, 1); RTX_DIAG_ADD(5, __popcll(busy)); RTX_DIAG_ADD(6, __popcll(nodes)); RTX_DIAG_ADD(7, (nodes && leaves) ? 1 : 0);
            if (~busy != 0 && don != 0) { RTX_DIAG_ADD(8, 1); RTX_DIAG_ADD(10, min(__popcll(~busy), __popcll(don))); }
        }
#endif
        // A round of giving costs about twenty cross-lane moves, so it waits until a quarter of the lanes is idle (the
        // phase starts with at least that many idle): rounds for one or two lanes that have just finished a small subtree
        // were nine of ten iterations on a scene of 64 M small primitives (+2 % there for 16 against 1, 8 or 24; nothing elsewhere).
        if (__popcll(~busy) >= kDonateMin && don != 0) {   // wave-uniform
            const int np = min(__popcll(~busy), __popcll(don));
            const int rd = __popcll(don & lt), ri = __popcll(~busy & lt);
            const bool giver = can_give && rd < np, taker = !has && ri < np;
            if (giver) __hip_atomic_store(&pair[rd], lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            const int src = taker ? __hip_atomic_load(&pair[ri], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : lane;
            __builtin_amdgcn_wave_barrier();
            const int bref = can_give ? st.get(bot) : 0;
            const int t_ref = __shfl(bref, src), t_owner = __shfl(owner, src);
            const float ox = __shfl(o.x, src), oy = __shfl(o.y, src), oz = __shfl(o.z, src);
            const float dx = __shfl(d.x, src), dy = __shfl(d.y, src), dz = __shfl(d.z, src);
            const float i0 = __shfl(rb.inv.x, src), i1 = __shfl(rb.inv.y, src), i2 = __shfl(rb.inv.z, src);
            const float m0 = __shfl(rb.on.x, src), m1 = __shfl(rb.on.y, src), m2 = __shfl(rb.on.z, src);
            const float p0 = __shfl(rb.of.x, src), p1 = __shfl(rb.of.y, src), p2 = __shfl(rb.of.z, src);
            const float tl = ANY ? __shfl(t_limit, src) : 0.0f, dl = ANY ? __shfl(dist_light, src) : 0.0f;
            if (giver) ++bot;
            if (taker) {
                cur = t_ref;
                owner = t_owner;
                o = v3(ox, oy, oz);
                d = v3(dx, dy, dz);
                rb.inv = v3(i0, i1, i2);
                rb.on = v3(m0, m1, m2);
                rb.of = v3(p0, p1, p2);
                if (ANY) {
                    t_limit = tl;
                    dist_light = dl;
                }
                sp = bot = 0;
            }
        }
#ifdef RTX_COUNT_STEPS
        const unsigned long long d1 = __builtin_readcyclecounter();
        st.t_don += d1 - d0;
#endif
        if (cur != kRefDone) {
            const unsigned long long key = __hip_atomic_load(&slot[owner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            if (ANY && key == 0ull) {   // another lane found an occluder for this ray
                cur = kRefDone;
                sp = bot;
            } else {
                const Fetched f = fetch_ref(s, tc, cur);
#ifdef RTX_COUNT_STEPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long d2 = __builtin_readcyclecounter();
                st.t_mem += d2 - d1;
#endif
                if (cur >= 0) {
                    cur = node4_step<!ANY, SPH>(f.a, f.b, f.c, f.d, rb, o, ANY ? t_limit : u2f((uint32_t)(key >> 32)), st, sp, bot, kRefDone);
                } else {
                    if (ANY) {
                        if (any_quads<SPH>(as_f4(f.a), as_f4(f.b), as_f4(f.c), o, d, dist_light)) {
                            __hip_atomic_store(&slot[owner], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            sp = bot;
                        }
                    } else {
                        float bt = u2f((uint32_t)(key >> 32));
                        int bi = (int)(uint32_t)key;
                        const float bt0 = bt;
                        const int bi0 = bi;
                        test_quads<SPH>(as_f4(f.a), as_f4(f.b), as_f4(f.c), o, d, bt, bi);
                        if (bt != bt0 || bi != bi0)
                            __hip_atomic_fetch_min(&slot[owner], ((unsigned long long)f2u(bt) << 32) | (uint32_t)bi, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
#ifdef RTX_COUNT_STEPS
                    ++st.leaves;
#endif
                    cur = (sp == bot) ? kRefDone : st.get(--sp);
                }
#ifdef RTX_COUNT_STEPS
                st.t_alu += __builtin_readcyclecounter() - d2;
#endif
            }
        }
    }
    RTX_DIAG_FLUSH();
    const unsigned long long key = __hip_atomic_load(&slot[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (ANY) {
        out_t = 0.0f;
        out_id = (key == 0ull) ? 1 : 0;
    } else {
        out_t = u2f((uint32_t)(key >> 32));
        const int id = (int)(uint32_t)key;
        out_id = (id == 0x7fffffff) ? -1 : id;
    }
}
