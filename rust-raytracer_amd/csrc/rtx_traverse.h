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
// One difference is documented rather than reproduced: a primitive test that
// returns Hit(NaN) (only possible for rays with non-finite components,
// sphere.rs:70-102) is ignored here, while the reference would keep it if it
// came first.
#pragma once
#include "rtx_internal.h"

#define RTX_BLOCK 256

RTX_D uint32_t f2u(float f) { return __float_as_uint(f); }
RTX_D float u2f(uint32_t u) { return __uint_as_float(u); }

// One candidate primitive against one ray; updates (best_t, best_id).
RTX_D void test_record(const PrimRec* __restrict__ recs, uint32_t idx, V3 o, V3 d, float& best_t, int& best_id) {
    const float4 q0 = recs[idx].q0;
    const float4 q1 = recs[idx].q1;
    const int id = (int)f2u(q0.w);
    float t;
    bool hit;
    if (f2u(q1.w) == 0u) {
        const float4 q2 = recs[idx].q2;
        hit = tri_intersect(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), &t);
    } else {
        hit = sphere_intersect(o, d, v3(q0.x, q0.y, q0.z), q1.x, &t);
    }
    if (hit && (t < best_t || (t == best_t && id < best_id))) {
        best_t = t;
        best_id = id;
    }
}

// The loop of src/scene/mod.rs:186-199 over every record.
RTX_D void trace_exhaustive(const DevScene& s, V3 o, V3 d, float& best_t, int& best_id) {
    best_t = __builtin_inff();
    best_id = 0x7fffffff;
    for (uint32_t i = 0; i < s.n_prims; ++i) test_record(s.recs, i, o, d, best_t, best_id);
    if (best_id == 0x7fffffff) best_id = -1;
}

struct RayBox {  // per-ray constants of the slab test
    V3 om, op;   // origin shifted by +pad / -pad
    V3 inv;
};
RTX_D float safe_inv(float d) {
    return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : 1.0f / d;
}
RTX_D RayBox make_raybox(const DevScene& s, V3 o, V3 d) {
    const float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
    const float pad = s.pad_rel * m;
    RayBox rb;
    rb.om = v3(o.x + pad, o.y + pad, o.z + pad);
    rb.op = v3(o.x - pad, o.y - pad, o.z - pad);
    rb.inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    return rb;
}
// Conservative slab test; t_in is the entry parameter (can be negative).
RTX_D bool box_hit(const RayBox& rb, float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                   float best_t, float& t_in) {
    const float ax = (bx0 - rb.om.x) * rb.inv.x, bx = (bx1 - rb.op.x) * rb.inv.x;
    const float ay = (by0 - rb.om.y) * rb.inv.y, by = (by1 - rb.op.y) * rb.inv.y;
    const float az = (bz0 - rb.om.z) * rb.inv.z, bz = (bz1 - rb.op.z) * rb.inv.z;
    const float lo = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t_in = lo;
    return (lo <= hi) && (hi >= 0.0f) && (lo <= best_t);
}

RTX_D bool in_safe_region(const DevScene& s, V3 o) {
    return fabsf(o.x - s.safe_c[0]) <= s.safe_r && fabsf(o.y - s.safe_c[1]) <= s.safe_r &&
           fabsf(o.z - s.safe_c[2]) <= s.safe_r;  // false for NaN origins
}

// Per-lane traversal stack in LDS: entry k of lane l lives at stack[k * RTX_BLOCK + l],
// so the lanes of a wavefront always touch 64 consecutive dwords (conflict-free
// whatever their individual depths are).
template <int STACK>
struct LaneStack {
    int* base;
    RTX_D void set(int* lds) { base = lds + threadIdx.x; }
    RTX_D void put(int sp, int v) { base[sp * RTX_BLOCK] = v; }
    RTX_D int get(int sp) const { return base[sp * RTX_BLOCK]; }
};

// Closest hit through the BVH. Returns best_id = -1 on a miss.
// `t_limit` bounds the search (inf for closest-hit queries).
template <int STACK>
RTX_D void trace_bvh_closest(const DevScene& s, V3 o, V3 d, LaneStack<STACK>& st, float& best_t, int& best_id) {
    if (!in_safe_region(s, o)) {
        trace_exhaustive(s, o, d, best_t, best_id);
        return;
    }
    best_t = __builtin_inff();
    best_id = 0x7fffffff;
    const RayBox rb = make_raybox(s, o, d);
    const BvhNode* __restrict__ nodes = s.nodes;
    int cur = 0, sp = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 q0 = nodes[cur].q0, q1 = nodes[cur].q1, q2 = nodes[cur].q2, q3 = nodes[cur].q3;
            float tl, tr;
            const bool hl = box_hit(rb, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best_t, tl);
            const bool hr = box_hit(rb, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best_t, tr);
            const int left = (int)f2u(q3.x), right = (int)f2u(q3.y);
            if (hl && hr) {
                const bool lnear = tl <= tr;
                st.put(sp, lnear ? right : left);
                sp = (sp + 1 < STACK) ? sp + 1 : sp;  // depth is checked on the host; never overflows
                cur = lnear ? left : right;
                continue;
            }
            if (hl) { cur = left; continue; }
            if (hr) { cur = right; continue; }
        } else {
            const uint32_t leaf = ~(uint32_t)cur;
            const uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k) test_record(s.recs, first + k, o, d, best_t, best_id);
        }
        if (sp == 0) break;
        cur = st.get(--sp);
    }
    if (best_id == 0x7fffffff) best_id = -1;
}

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
RTX_D bool any_record(const PrimRec* __restrict__ recs, uint32_t idx, V3 o, V3 d, float dist_light) {
    const float4 q0 = recs[idx].q0;
    const float4 q1 = recs[idx].q1;
    float t;
    bool hit;
    if (f2u(q1.w) == 0u) {
        const float4 q2 = recs[idx].q2;
        hit = tri_intersect(o, d, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), &t);
    } else {
        hit = sphere_intersect(o, d, v3(q0.x, q0.y, q0.z), q1.x, &t);
    }
    return hit && occluding_hit(o, d, t, dist_light);
}
template <int STACK>
RTX_D bool trace_any(const DevScene& s, V3 o, V3 d, float dist_light, LaneStack<STACK>& st) {
    if (!s.uses_bvh || !in_safe_region(s, o)) {
        for (uint32_t i = 0; i < s.n_prims; ++i)
            if (any_record(s.recs, i, o, d, dist_light)) return true;
        return false;
    }
    // Box culling bound: hits beyond the light never occlude. |d| is ~1 for point
    // lights (Light::get_dir normalises) but is kept general.
    float t_limit = __builtin_inff();
    if (dist_light >= 0.0f) {
        // vdistance(o, o + d*t) carries an absolute rounding error of a few ulp of |o|.
        const float len = vlength(d);
        const float m = fmaxf(fabsf(o.x), fmaxf(fabsf(o.y), fabsf(o.z)));
        t_limit = (len > 0.0f) ? ((dist_light + 4e-6f * m) / len) * 1.0001f + 1e-6f : __builtin_inff();
    }
    const RayBox rb = make_raybox(s, o, d);
    const BvhNode* __restrict__ nodes = s.nodes;
    int cur = 0, sp = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 q0 = nodes[cur].q0, q1 = nodes[cur].q1, q2 = nodes[cur].q2, q3 = nodes[cur].q3;
            float tl, tr;
            const bool hl = box_hit(rb, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, t_limit, tl);
            const bool hr = box_hit(rb, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, t_limit, tr);
            const int left = (int)f2u(q3.x), right = (int)f2u(q3.y);
            if (hl && hr) {
                st.put(sp, right);
                sp = (sp + 1 < STACK) ? sp + 1 : sp;
                cur = left;
                continue;
            }
            if (hl) { cur = left; continue; }
            if (hr) { cur = right; continue; }
        } else {
            const uint32_t leaf = ~(uint32_t)cur;
            const uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; ++k)
                if (any_record(s.recs, first + k, o, d, dist_light)) return true;
        }
        if (sp == 0) break;
        cur = st.get(--sp);
    }
    return false;
}

// Closest hit in either mode.
template <int STACK>
RTX_D void trace_closest(const DevScene& s, V3 o, V3 d, LaneStack<STACK>& st, float& best_t, int& best_id) {
    if (s.uses_bvh) trace_bvh_closest<STACK>(s, o, d, st, best_t, best_id);
    else trace_exhaustive(s, o, d, best_t, best_id);
}
