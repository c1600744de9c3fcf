// rtx_shading.h -- per-hit geometry, secondary rays and the local lighting terms.
// Every function restates the cited reference function with the same operation
// order; the recursion of src/lib.rs:201-255 itself is flattened by the kernels
// in rtx_shade.hip.
#pragma once
#include "rtx_internal.h"

struct HitGeom {
    V3 p;           // Intersection::point            (intersection.rs:26-28)
    V3 n;           // Intersection::surface_normal   (intersection.rs:38-40)
    Col cd;         // Intersection::color            (intersection.rs:30-32)
    uint32_t mat0;  // Intersection::material = materials[0] (poly.rs:181-183, sphere.rs:105-107)
    bool panic;     // the reference would have panicked (poly.rs:84-86 or an index past the material list)
    bool blended;   // cd is a blend of three vertex materials (poly.rs:91-94); otherwise it is materials[mat0].diffuse, or black without that material
};

RTX_D V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

RTX_D const rtx_material* material_or_null(const DevScene& s, uint32_t id) {
    return (id < s.n_materials) ? &s.mats[id] : nullptr;
}

// Poly::weighted_areas (poly.rs:78-89)
RTX_D void weighted_areas(V3 p0, V3 p1, V3 p2, V3 point, float& a0, float& a1, float& a2, bool& panic) {
    const float area = varea(p0, p1, p2);
    a0 = varea(p0, p1, point) / area;
    a1 = varea(p2, p0, point) / area;
    a2 = varea(p1, p2, point) / area;
    if (a0 > 1.0f || a1 > 1.0f || a2 > 1.0f) panic = true;
}

// Point, normal (and optionally colour) of a hit. `want_color` is false for
// shadow-ray continuation, which only needs the normal and the material.
RTX_D HitGeom hit_geometry(const DevScene& s, int prim_id, V3 o, V3 d, float t, bool want_color) {
    HitGeom g;
    g.panic = false;
    g.blended = false;
    g.p = o + vmult(d, t);
    g.cd = col_new();
    const uint4* rec = s.shade_recs + (size_t)prim_id * 4;   // one aligned 64-byte record per primitive (DevScene::shade_recs)
    const uint4 ra = rec[0], rb = rec[1];
    const uint32_t ref = ra.w;
    if (ref >> 31) {  // sphere
        g.n = vnormalize(g.p - v3(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z)));  // sphere.rs:109-113 (never flipped)
        g.mat0 = rb.w;
        if (want_color) {
            const rtx_material* m = material_or_null(s, g.mat0);
            if (m) g.cd = col_raw(m->diffuse); else g.panic = true;
        }
        return g;
    }
    const uint4 rc = rec[2];
    const uint32_t idx = ref & 0x0fffffffu;
    const V3 p0 = v3(__uint_as_float(ra.x), __uint_as_float(ra.y), __uint_as_float(ra.z));
    const V3 p1 = v3(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z));
    const V3 p2 = v3(__uint_as_float(rc.x), __uint_as_float(rc.y), __uint_as_float(rc.z));
    const uint32_t flags = (ref >> 28) & 3u;
    const uint32_t m0 = rb.w;
    g.mat0 = m0;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    if (flags & (RTX_TRI_VERTEX_NORMAL | RTX_TRI_VERTEX_MATERIAL)) weighted_areas(p0, p1, p2, g.p, a0, a1, a2, g.panic);
    V3 normal;
    if (flags & RTX_TRI_VERTEX_NORMAL) {  // poly.rs:102-105
        const float* tn = s.tri_nrm + (size_t)idx * 9;
        normal = vmult(ld3(tn), a2) + vmult(ld3(tn + 3), a1) + vmult(ld3(tn + 6), a0);
    } else {                              // poly.rs:96-100
        normal = vcross(p1 - p0, p2 - p0);
    }
    normal = vnormalize(normal);          // poly.rs:185-196
    if (vdot(normal, d) > 0.0f) normal = vinvert(normal);
    g.n = normal;
    if (want_color) {
        if (flags & RTX_TRI_VERTEX_MATERIAL) {  // poly.rs:91-94
            const rtx_material* ma = material_or_null(s, m0);
            const rtx_material* mb = material_or_null(s, rc.w);
            const rtx_material* mc = material_or_null(s, rec[3].x);
            if (!ma || !mb || !mc) g.panic = true;
            const Col da = ma ? col_raw(ma->diffuse) : col_new();
            const Col db = mb ? col_raw(mb->diffuse) : col_new();
            const Col dc = mc ? col_raw(mc->diffuse) : col_new();
            g.cd = col_mult(da, a2) + col_mult(db, a1) + col_mult(dc, a0);
            g.blended = true;
        } else {                                // poly.rs:201
            const rtx_material* m = material_or_null(s, m0);
            if (m) g.cd = col_raw(m->diffuse); else g.panic = true;
        }
    }
    return g;
}

// Intersection::reflective_ray (intersection.rs:42-48)
RTX_D void reflective_ray(V3 dir, V3 point, V3 normal, V3& ro, V3& rd) {
    const V3 d0 = vinvert(dir);
    ro = point + vmult(normal, 0.0001f);
    rd = vmult(normal, vdot(d0, normal) * 2.0f) - d0;
}
// Intersection::refractive_ray (intersection.rs:50-81); false = total internal reflection.
RTX_D bool refractive_ray(V3 in_dir, V3 point, V3 normal, bool in_vacuum, V3& ro, V3& rd) {
    const float n = in_vacuum ? 1.0f / 1.5f : 1.5f / 1.0f;
    const float cos_in = vdot(normal, in_dir);
    if (cos_in > 0.0f) normal = vinvert(normal);
    const float c = vdot(in_dir, normal);
    const float cos_phi_2 = 1.0f - n * n * (1.0f - c * c);
    if (cos_phi_2 < 0.0f) return false;
    const float cos_phi = sqrtf(cos_phi_2);
    V3 term1 = in_dir - vmult(normal, c);
    term1 = vmult(term1, n);
    rd = term1 - vmult(normal, cos_phi);
    ro = point - vmult(normal, 0.01f);
    return true;
}

// Area lights. The reference draws AreaLight::sample_point from an OS-seeded generator
// (scene/mod.rs:94-96), so its images are not reproducible; here every draw is a pure
// function of a 64-bit key that names the draw: (pixel, path through the ray tree, light,
// call site, sample number). Keys are derived with rng_push along that path.
RTX_D uint64_t rng_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RTX_D uint64_t rng_push(uint64_t key, uint64_t v) { return rng_mix64(key ^ rng_mix64(v)); }
RTX_D float rng_open01(uint64_t h) {   // 23 random bits -> a float strictly inside (0, 1)
    return ((float)(uint32_t)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);
}
// Key offsets of the draws (one namespace per call site of the reference).
constexpr uint64_t kKeyLight = 0x1000;     // + light index, shade_intersection's loop (lib.rs:216)
constexpr uint64_t kKeyShadowSample = 0x100;   // + sample, shadow_scalar's loop (lib.rs:113)
constexpr uint64_t kKeyDirectSample = 0x200;   // + sample, direct_lightning's loop (lib.rs:181)
constexpr uint64_t kKeyReflect = 0x2000, kKeyRefract = 0x3000;   // children of a node (lib.rs:234,244)

// AreaLight::sample_point (scene/mod.rs:93-104): a = min corner, b = max corner.
RTX_D V3 area_sample_point(const DevLight& l, uint64_t key) {
    const float rx = rng_open01(rng_mix64(key ^ 1)), ry = rng_open01(rng_mix64(key ^ 2)), rz = rng_open01(rng_mix64(key ^ 3));
    float dx = fabsf(l.b[0] - l.a[0]) * 0.5f;
    float dy = fabsf(l.b[1] - l.a[1]) * 0.5f;
    float dz = fabsf(l.b[2] - l.a[2]) * 0.5f;
    dx = dx - rx * (dx * 2.0f);
    dy = dy - ry * (dy * 2.0f);
    dz = dz - rz * (dz * 2.0f);
    return v3(l.b[0] + dx, l.b[1] + dy, l.b[2] + dz);
}
// Light::position / Light::get_dir (scene/mod.rs:35-59); `key` is only read for area lights.
RTX_D V3 light_pos(const DevLight& l, uint64_t key) {
    if (l.type == RTX_LIGHT_AREA) return area_sample_point(l, key);
    return (l.type == RTX_LIGHT_POINT) ? v3(l.a[0], l.a[1], l.a[2]) : v3(0.0f, 0.0f, 0.0f);
}
RTX_D V3 light_dir(const DevLight& l, V3 point, uint64_t key) {
    if (l.type == RTX_LIGHT_DIRECTIONAL) return vinvert(v3(l.a[0], l.a[1], l.a[2]));
    if (l.type == RTX_LIGHT_AREA) return vnormalize(area_sample_point(l, key) - point);
    return vnormalize(v3(l.a[0], l.a[1], l.a[2]) - point);
}
// RayTracer::calculate_fattj (lib.rs:147-155)
RTX_D float calculate_fattj(const DevLight& l, V3 point, uint64_t key) {
    if (l.type == RTX_LIGHT_DIRECTIONAL) return 1.0f;
    const float dist = vdistance(point, light_pos(l, key));
    return fminf(1.0f, 1.0f / (0.25f + 0.1f * dist + 0.01f * dist * dist));
}
// RayTracer::direct_lightning (lib.rs:171-199): n = num_samples for an area light, else 1;
// `key` names this call, sample s draws with key + kKeyDirectSample + s.
RTX_D Col direct_lightning(const DevLight& l, V3 point, V3 normal, V3 ray_dir, Col cd, Col ks, float kt,
                           float shininess, Col sj, float fattj, uint32_t n, uint64_t key) {
    const float q = shininess * 128.0f;
    const Col direct_light = col_mult(col_raw(l.color) * sj, fattj);
    Col lightning = col_new();
    const float nf = (float)n;
    for (uint32_t smp = 0; smp < n; ++smp) {
        const V3 dir = light_dir(l, point, (l.type == RTX_LIGHT_AREA) ? rng_push(key, kKeyDirectSample + smp) : 0ull);
        // diffuse_lightning (lib.rs:157-161)
        const float a = 1.0f - kt;
        const float b = fmaxf(0.0f, vdot(normal, dir));
        const Col diffuse_light = col_mult(cd, a * b);
        // specular_lightning (lib.rs:163-169)
        const V3 v = vinvert(ray_dir);
        float t = vdot(normal, dir);
        V3 rj = vmult(vmult(normal, t), 2.0f);
        rj = rj - dir;
        t = fmaxf(vdot(rj, v), 0.0f);
        // ks == 0 and a base in [0, 1] with a non-negative exponent: powf is finite, so ks * powf is
        // exactly zero and the call (the most expensive operation of the shading pass) can be skipped
        const bool matte = ks.r == 0.0f && ks.g == 0.0f && ks.b == 0.0f && t <= 1.0f && q >= 0.0f;
        const Col specular_light = matte ? col_new() : col_mult(ks, powf(t, q));
        Col sample = direct_light * (diffuse_light + specular_light);
        if (n != 1u) sample = col_init(sample.r / nf, sample.g / nf, sample.b / nf);   // x / 1.0f is x: three divisions less per point light
        else sample = col_init(sample.r, sample.g, sample.b);
        lightning = lightning + sample;
    }
    return lightning;
}
