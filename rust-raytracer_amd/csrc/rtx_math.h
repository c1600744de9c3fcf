// rtx_math.h -- f32 vector / colour arithmetic shared by every kernel.
//
// Each function restates one reference function with the SAME operation order
// (left-to-right adds, no fused multiply-add: the library is compiled with
// -ffp-contract=off, correctly rounded divide and sqrt, f32 denormals kept), so
// that a hit's `t`, its point and its normal are bit-identical to what the
// reference computes on the CPU for the same inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RTX_HD __host__ __device__ __forceinline__
#define RTX_D __device__ __forceinline__

struct V3 {
    float x, y, z;
};
RTX_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RTX_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }  // src/vec.rs:41-47
RTX_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }  // src/vec.rs:49-55
RTX_HD V3 vmult(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }         // src/vec.rs:95-97
RTX_HD float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // src/vec.rs:115-117
RTX_HD V3 vcross(V3 a, V3 b) {                                                   // src/vec.rs:108-113
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RTX_HD float vlength(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  // src/vec.rs:91-93
RTX_HD V3 vnormalize(V3 a) {                                                     // src/vec.rs:99-106
    float len = vlength(a);
    if (len != 0.0f) {
        a.x = a.x / len;
        a.y = a.y / len;
        a.z = a.z / len;
    }
    return a;
}
RTX_HD V3 vinvert(V3 a) { return vmult(a, -1.0f); }                              // src/vec.rs:119-121
RTX_HD float vdistance(V3 a, V3 o) {                                             // src/vec.rs:123-128
    float p = a.x - o.x, q = a.y - o.y, r = a.z - o.z;
    return sqrtf(p * p + q * q + r * r);
}
RTX_HD float varea(V3 a, V3 b, V3 c) {                                           // src/vec.rs:130-134
    return vlength(vcross(b - a, c - a)) * 0.5f;
}

// Color: every constructor clamps to [0,1] (src/scene/material.rs:17-27); a NaN
// passes through both comparisons unchanged.
RTX_HD float clamp01(float v) {
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}
struct Col {
    float r, g, b;
};
RTX_HD Col col_new() { return Col{0.0f, 0.0f, 0.0f}; }
RTX_HD Col col_init(float r, float g, float b) { return Col{clamp01(r), clamp01(g), clamp01(b)}; }
RTX_HD Col col_raw(const float* p) { return Col{p[0], p[1], p[2]}; }  // already-clamped stored colour
RTX_HD float col_scalar(Col c) { return sqrtf(c.r * c.r + c.g * c.g + c.b * c.b); }   // material.rs:53-55
RTX_HD Col col_mult(Col c, float s) { return col_init(c.r * s, c.g * s, c.b * s); }   // material.rs:65-67
RTX_HD Col operator*(Col a, Col b) { return col_init(a.r * b.r, a.g * b.g, a.b * b.b); }  // material.rs:70-76
RTX_HD Col operator+(Col a, Col b) { return col_init(a.r + b.r, a.g + b.g, a.b + b.b); }  // material.rs:78-84
RTX_HD uint8_t col_u8(float c) {  // `(c * 255.0) as u8` (material.rs:57-63): truncation; NaN -> 0
    float v = c * 255.0f;
    if (!(v >= 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

// Poly::intersects (src/scene/shapes/poly.rs:138-179) with e1 = v1 - v0 and
// e2 = v2 - v0 taken from the primitive record (they are the same f32 values
// the reference recomputes per call). Returns true and sets t on Hit.
//
// Written without early returns: a GPU lane gains nothing from leaving early
// (its wavefront keeps going) and every `return` costs an exec-mask branch. The
// arithmetic that produces t is the reference's, in the reference's order, and the
// accept condition is the conjunction of the negated reject tests, which is what
// the early-return chain evaluates (a NaN fails no reject test and then fails
// `t > 1e-7`, exactly as in the chain).
RTX_HD bool tri_intersect(V3 p, V3 d, V3 v0, V3 e1, V3 e2, float* t_out) {
    const V3 h = vcross(d, e2);
    const float a0 = vdot(e1, h);
    const float f = 1.0f / a0;
    const V3 s = p - v0;
    const float u = f * vdot(s, h);
    const V3 q = vcross(s, e1);
    const float v = f * vdot(d, q);
    const float t = f * vdot(e2, q);
    const bool parallel = a0 > -0.0000001f && a0 < 0.0000001f;
    const bool out_u = u < 0.0f || u > 1.0f;
    const bool out_v = v < 0.0f || u + v > 1.0f;
    *t_out = t;
    return !parallel && !out_u && !out_v && t > 0.0000001f;
}

// Sphere::intersects (src/scene/shapes/sphere.rs:54-103).
RTX_HD bool sphere_intersect(V3 ori, V3 dest, V3 center, float radius, float* t_out) {
    V3 orig = ori - center;
    float a = vdot(dest, dest);
    float b = 2.0f * vdot(dest, orig);
    float c = vdot(orig, orig) - (radius * radius);
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float dist_sqrt = sqrtf(disc);
    float q = (b < 0.0f) ? (-b - dist_sqrt) / 2.0f : (-b + dist_sqrt) / 2.0f;
    float t0 = q / a;
    float t1 = c / q;
    if (t0 > t1) {
        float tmp = t0;
        t0 = t1;
        t1 = tmp;
    }
    if (t1 < 0.0f) return false;
    *t_out = (t0 < 0.0f) ? t1 : t0;
    return true;
}
