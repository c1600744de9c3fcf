// rt_oracle.cpp -- CPU oracle: a C++ restatement of sondrele/rust-raytracer.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h). Every function cites the
// reference file:line whose arithmetic it follows, operation by operation and
// in the same order; all arithmetic is IEEE f32 (build with -ffp-contract=off,
// no fast-math, SSE scalar maths) so results are comparable bit for bit with a
// device kernel that uses the same operation order.
//
// Deviations that are deliberate and documented:
//  * Rust `panic!`s become either an error string (parser) or a per-pixel
//    "would have panicked" flag (Poly::weighted_areas, material indexing); the
//    oracle keeps computing so a full image exists.
//  * AreaLight::sample_point draws from an OS-seeded RNG in the reference
//    (src/scene/mod.rs:94-96) and is therefore not reproducible; here the three
//    uniforms come from a counter-based hash of (pixel, ray-tree path, light,
//    call site, sample index).
//  * Number parsing uses correctly rounded strtof.
#include "rt_oracle.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_error;

// ----------------------------------------------------------------- vec.rs --
struct Vec3 {
    float x, y, z;
};
inline Vec3 V(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 operator+(Vec3 a, Vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }  // vec.rs:41-47
inline Vec3 operator-(Vec3 a, Vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }  // vec.rs:49-55
inline Vec3 operator*(Vec3 a, Vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }  // vec.rs:57-63
inline float vidx(Vec3 a, unsigned i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); } // vec.rs:65-76
inline float length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }  // vec.rs:91-93
inline Vec3 mult(Vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }            // vec.rs:95-97
inline Vec3 normalize(Vec3 a) {                                                       // vec.rs:99-106
    float len = length(a);
    if (len != 0.0f) {
        a.x = a.x / len;
        a.y = a.y / len;
        a.z = a.z / len;
    }
    return a;
}
inline Vec3 cross(Vec3 a, Vec3 b) {                                                   // vec.rs:108-113
    float x = a.y * b.z - a.z * b.y;
    float y = a.z * b.x - a.x * b.z;
    float z = a.x * b.y - a.y * b.x;
    return V(x, y, z);
}
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        // vec.rs:115-117
inline Vec3 invert(Vec3 a) { return mult(a, -1.0f); }                                 // vec.rs:119-121
inline float distance(Vec3 a, Vec3 o) {                                               // vec.rs:123-128
    float p = a.x - o.x;
    float q = a.y - o.y;
    float r = a.z - o.z;
    return std::sqrt(p * p + q * q + r * r);
}
inline float get_area(Vec3 a, Vec3 b, Vec3 c) {                                       // vec.rs:130-134
    Vec3 ab = b - a;
    Vec3 ac = c - a;
    return length(cross(ab, ac)) * 0.5f;
}
inline bool vec_lt(Vec3 a, Vec3 b) {                                                  // vec.rs:26-38
    if (a.x > b.x) return false;
    else if (a.x < b.x) return true;
    else if (a.y < b.y) return true;
    else if (a.y == b.y && a.z < b.z) return true;
    else return false;
}

// ----------------------------------------------------------------- ray.rs --
struct Ray {
    Vec3 ori, dir;
    bool vacuum;  // Cell<bool>, true after Ray::init (ray.rs:13-26)
};
inline Ray ray_init(Vec3 o, Vec3 d) { return Ray{o, d, true}; }

// ------------------------------------------------------------ material.rs --
inline float clamp01(float v) {  // material.rs:23-27 (NaN passes through both tests)
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}
struct Color {
    float r, g, b;
};
inline Color color_new() { return Color{0.0f, 0.0f, 0.0f}; }                           // material.rs:13-15
inline Color color_init(float r, float g, float b) { return Color{clamp01(r), clamp01(g), clamp01(b)}; }  // :17-21
inline float scalar(Color c) { return std::sqrt(c.r * c.r + c.g * c.g + c.b * c.b); }  // :53-55
inline Color cmult(Color c, float s) { return color_init(c.r * s, c.g * s, c.b * s); } // :65-67
inline Color operator*(Color a, Color b) { return color_init(a.r * b.r, a.g * b.g, a.b * b.b); }  // :70-76
inline Color operator+(Color a, Color b) { return color_init(a.r + b.r, a.g + b.g, a.b + b.b); }  // :78-84
inline uint8_t to_u8(float c) {  // `(c * 255.0) as u8`, material.rs:57-63 (truncation; NaN defined as 0 here)
    float v = c * 255.0f;
    if (!(v >= 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
struct Material {  // material.rs:86-94
    Color diffuse, ambient, specular, emissive;
    float shininess, transparency;
};

// ------------------------------------------------------------ scene/mod.rs --
struct Light {  // scene/mod.rs:19-24,62-120
    uint32_t type;
    Vec3 a, b;
    Color intensity;
};
struct Camera {
    Vec3 pos, view_dir;
    float focal_dist;
    Vec3 ortho_up;
    float vertical_fov;
};

// Deterministic stand-in for rand::random::<Open01<f32>>() (scene/mod.rs:94-96).
struct RngKey {
    uint64_t h;
};
inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline RngKey key_push(RngKey k, uint64_t v) { return RngKey{mix64(k.h ^ mix64(v))}; }
inline float open01(uint64_t h) {  // 23 random bits -> (0,1)
    return ((float)(uint32_t)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);
}
inline Vec3 area_sample_point(const Light& l, RngKey k) {  // scene/mod.rs:93-104
    float rx = open01(mix64(k.h ^ 1)), ry = open01(mix64(k.h ^ 2)), rz = open01(mix64(k.h ^ 3));
    float dx = std::fabs(l.b.x - l.a.x) * 0.5f;
    float dy = std::fabs(l.b.y - l.a.y) * 0.5f;
    float dz = std::fabs(l.b.z - l.a.z) * 0.5f;
    dx = dx - rx * (dx * 2.0f);
    dy = dy - ry * (dy * 2.0f);
    dz = dz - rz * (dz * 2.0f);
    return V(l.b.x + dx, l.b.y + dy, l.b.z + dz);
}
inline Vec3 light_position(const Light& l, RngKey k) {  // scene/mod.rs:35-41
    switch (l.type) {
        case RTX_LIGHT_POINT: return l.a;
        case RTX_LIGHT_AREA: return area_sample_point(l, k);
        default: return V(0.0f, 0.0f, 0.0f);
    }
}
inline Vec3 light_get_dir(const Light& l, Vec3 point, RngKey k) {  // scene/mod.rs:43-59
    switch (l.type) {
        case RTX_LIGHT_DIRECTIONAL: return invert(l.a);
        case RTX_LIGHT_POINT: return normalize(l.a - point);
        default: return normalize(area_sample_point(l, k) - point);
    }
}

// ------------------------------------------------------------- shapes/*.rs --
struct BBox {
    Vec3 min, max;
};
inline Vec3 bbox_centroid(const BBox& b) { return mult(b.min, 0.5f) + mult(b.max, 0.5f); }  // shapes/mod.rs:40-42
inline bool bbox_intersects(const BBox& b, const Ray& ray) {                                // shapes/mod.rs:44-83
    Vec3 ori = ray.ori, dir = ray.dir;
    float tmin = (b.min.x - ori.x) / dir.x;
    float tmax = (b.max.x - ori.x) / dir.x;
    if (tmin > tmax) std::swap(tmin, tmax);
    float tymin = (b.min.y - ori.y) / dir.y;
    float tymax = (b.max.y - ori.y) / dir.y;
    if (tymin > tymax) std::swap(tymin, tymax);
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (b.min.z - ori.z) / dir.z;
    float tzmax = (b.max.z - ori.z) / dir.z;
    if (tzmin > tzmax) std::swap(tzmin, tzmax);
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}
inline BBox bbox_union(const BBox& a, const BBox& o) {  // shapes/mod.rs:86-104 (f32::min/max = fminf/fmaxf)
    BBox r;
    r.min = V(std::fmin(a.min.x, o.min.x), std::fmin(a.min.y, o.min.y), std::fmin(a.min.z, o.min.z));
    r.max = V(std::fmax(a.max.x, o.max.x), std::fmax(a.max.y, o.max.y), std::fmax(a.max.z, o.max.z));
    return r;
}

enum { PRIM_POLY = 0, PRIM_SPHERE = 1 };
struct Prim {
    uint8_t kind;
    uint8_t vertex_normal, vertex_material;  // poly.rs:54-55
    uint8_t nmat;                            // length of the poly's de-duplicated material list
    uint32_t mat[3];                         // global ids of materials[0..2]
    Vec3 p[3];                               // vertex positions (poly) ; p[0] = origin (sphere)
    Vec3 n[3];                               // vertex normals (poly)
    float radius;                            // sphere
};

struct Hit {
    bool hit;
    float t;
};
inline Hit poly_intersects(const Prim& q, const Ray& ray) {  // poly.rs:138-179
    Vec3 p = ray.ori, d = ray.dir;
    Vec3 v0 = q.p[0], v1 = q.p[1], v2 = q.p[2];
    Vec3 e1 = v1 - v0;
    Vec3 e2 = v2 - v0;
    Vec3 h = cross(d, e2);
    float a0 = dot(e1, h);
    if (a0 > -0.0000001f && a0 < 0.0000001f) return Hit{false, 0.0f};
    float f = 1.0f / a0;
    Vec3 s = p - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return Hit{false, 0.0f};
    Vec3 qq = cross(s, e1);
    float v = f * dot(d, qq);
    if (v < 0.0f || u + v > 1.0f) return Hit{false, 0.0f};
    float t = f * dot(e2, qq);
    if (t > 0.0000001f) return Hit{true, t};
    return Hit{false, 0.0f};
}
inline Hit sphere_intersects(const Prim& s, const Ray& ray) {  // sphere.rs:54-103
    Vec3 orig = ray.ori - s.p[0];
    Vec3 dest = ray.dir;
    float a = dot(dest, dest);
    float b = 2.0f * dot(dest, orig);
    float c = dot(orig, orig) - (s.radius * s.radius);
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return Hit{false, 0.0f};
    float dist_sqrt = std::sqrt(disc);
    float q = (b < 0.0f) ? (-b - dist_sqrt) / 2.0f : (-b + dist_sqrt) / 2.0f;
    float t0 = q / a;
    float t1 = c / q;
    if (t0 > t1) std::swap(t0, t1);
    if (t1 < 0.0f) return Hit{false, 0.0f};
    return (t0 < 0.0f) ? Hit{true, t1} : Hit{true, t0};
}
inline Hit prim_intersects(const Prim& p, const Ray& r) {  // shapes/mod.rs:145-150
    return p.kind == PRIM_POLY ? poly_intersects(p, r) : sphere_intersects(p, r);
}
inline BBox prim_bbox(const Prim& q) {
    if (q.kind == PRIM_POLY) {  // poly.rs:122-136
        BBox b;
        b.min = V(std::fmin(q.p[0].x, std::fmin(q.p[1].x, q.p[2].x)), std::fmin(q.p[0].y, std::fmin(q.p[1].y, q.p[2].y)),
                  std::fmin(q.p[0].z, std::fmin(q.p[1].z, q.p[2].z)));
        b.max = V(std::fmax(q.p[0].x, std::fmax(q.p[1].x, q.p[2].x)), std::fmax(q.p[0].y, std::fmax(q.p[1].y, q.p[2].y)),
                  std::fmax(q.p[0].z, std::fmax(q.p[1].z, q.p[2].z)));
        return b;
    }
    BBox b;  // sphere.rs:45-52
    b.min = mult(V(-1.0f, -1.0f, -1.0f), q.radius) + q.p[0];
    b.max = mult(V(1.0f, 1.0f, 1.0f), q.radius) + q.p[0];
    return b;
}

// --------------------------------------------------------------- bvh.rs ----
struct BvhNode {
    int kind = 0;  // 0 Empty, 1 Member, 2 Leaf  (bvh.rs:8-13)
    std::unique_ptr<BvhNode> left, right;
    int32_t prim = -1;  // index into Scene::prims
    BBox bbox{V(0, 0, 0), V(0, 0, 0)};
};

struct Scene {
    Camera camera{};
    std::vector<Light> lights;
    std::vector<Prim> prims;
    std::vector<Material> materials;
    std::unique_ptr<BvhNode> bvh;  // built lazily (Tree::init, bvh.rs:82-86)
};

std::unique_ptr<BvhNode> bvh_build(const Scene& sc, std::vector<int32_t>& ids, size_t lo, size_t hi, size_t depth) {
    // bvh.rs:88-113. The reference sorts clones of the primitives; sorting ids is equivalent.
    auto node = std::make_unique<BvhNode>();
    size_t len = hi - lo;
    if (len == 0) return node;  // Empty
    if (len == 1) {
        node->kind = 2;
        node->prim = ids[lo];
        node->bbox = prim_bbox(sc.prims[ids[lo]]);
        return node;
    }
    unsigned axis = (unsigned)(depth % 3);
    // Rust's sort_by is a stable merge sort; the comparator never returns Equal (bvh.rs:98-103).
    std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](int32_t a, int32_t b) {
        return vidx(bbox_centroid(prim_bbox(sc.prims[a])), axis) < vidx(bbox_centroid(prim_bbox(sc.prims[b])), axis);
    });
    size_t half = len / 2;
    node->kind = 1;
    node->left = bvh_build(sc, ids, lo, lo + half, depth + 1);
    node->right = bvh_build(sc, ids, lo + half, hi, depth + 1);
    node->bbox = bbox_union(node->left->bbox, node->right->bbox);  // TreeNode::init, bvh.rs:47-56
    return node;
}

struct NodeHit {
    const BvhNode* node;
    float t;
};
NodeHit bvh_intersect(const Scene& sc, const BvhNode* node, const Ray& ray, uint64_t* tests) {  // bvh.rs:119-143
    if (!node || node->kind == 0) return NodeHit{nullptr, 0.0f};
    if (node->kind == 2) {
        ++*tests;
        Hit h = prim_intersects(sc.prims[node->prim], ray);
        return h.hit ? NodeHit{node, h.t} : NodeHit{nullptr, 0.0f};
    }
    if (!bbox_intersects(node->bbox, ray)) return NodeHit{nullptr, 0.0f};
    NodeHit l = bvh_intersect(sc, node->left.get(), ray, tests);
    NodeHit r = bvh_intersect(sc, node->right.get(), ray, tests);
    if (l.node && r.node) return (l.t < r.t) ? l : r;  // bvh.rs:134 (right wins ties)
    if (l.node) return l;
    if (r.node) return r;
    return NodeHit{nullptr, 0.0f};
}

// -------------------------------------------------------- intersection.rs --
struct Isect {
    bool hit;
    float t;           // `point: f32`
    Ray ray;           // clone of the ray (scene/mod.rs:191,195)
    const Prim* prim;
    int32_t prim_id;
};

struct Ctx {  // per-thread render state
    const Scene* sc;
    bool use_bvh;
    orc_stats st{};
    bool panic = false;
};

Isect scene_intersects(Ctx& cx, const Ray& ray) {
    const Scene& sc = *cx.sc;
    Isect out{false, 0.0f, ray, nullptr, -1};
    if (cx.use_bvh) {  // BvhScene::intersects, scene/mod.rs:237-244
        NodeHit h = bvh_intersect(sc, sc.bvh.get(), ray, &cx.st.prim_tests);
        if (h.node) out = Isect{true, h.t, ray, &sc.prims[h.node->prim], h.node->prim};
        return out;
    }
    // Scene::intersects, scene/mod.rs:181-201
    float point = 0.0f;
    bool has_intersected = false;
    const size_t n = sc.prims.size();
    cx.st.prim_tests += n;
    for (size_t i = 0; i < n; ++i) {
        Hit h = prim_intersects(sc.prims[i], ray);
        if (!h.hit) continue;
        if (!has_intersected) {
            has_intersected = true;
            point = h.t;
            out = Isect{true, point, ray, &sc.prims[i], (int32_t)i};
        } else if (h.t < point) {
            point = h.t;
            out = Isect{true, point, ray, &sc.prims[i], (int32_t)i};
        }
    }
    return out;
}

inline Vec3 isect_point(const Isect& i) { return i.ray.ori + mult(i.ray.dir, i.t); }  // intersection.rs:26-28

struct Areas {
    float a0, a1, a2;
};
Areas weighted_areas(Ctx& cx, const Prim& q, Vec3 point) {  // poly.rs:78-89
    float area = get_area(q.p[0], q.p[1], q.p[2]);
    float area0 = get_area(q.p[0], q.p[1], point) / area;
    float area1 = get_area(q.p[2], q.p[0], point) / area;
    float area2 = get_area(q.p[1], q.p[2], point) / area;
    if (area0 > 1.0f || area1 > 1.0f || area2 > 1.0f) cx.panic = true;  // reference: panic!
    return Areas{area0, area1, area2};
}
Vec3 surface_normal(Ctx& cx, const Prim& q, Vec3 direction, Vec3 point) {
    if (q.kind == PRIM_SPHERE) return normalize(point - q.p[0]);  // sphere.rs:109-113
    Vec3 normal;                                                  // poly.rs:185-196
    if (q.vertex_normal) {                                        // poly.rs:102-105
        Areas w = weighted_areas(cx, q, point);
        normal = mult(q.n[0], w.a2) + mult(q.n[1], w.a1) + mult(q.n[2], w.a0);
    } else {                                                      // poly.rs:96-100
        Vec3 v = q.p[1] - q.p[0];
        Vec3 w = q.p[2] - q.p[0];
        normal = cross(v, w);
    }
    normal = normalize(normal);
    if (dot(normal, direction) > 0.0f) normal = invert(normal);
    return normal;
}
const Material& mat_at(Ctx& cx, const Prim& q, int k) {
    static const Material black{};
    if (k >= q.nmat || q.mat[k] >= cx.sc->materials.size()) {  // reference: index out of bounds panic (poly.rs:93); RTX_MATERIAL_NONE is past any list
        cx.panic = true;
        return black;
    }
    return cx.sc->materials[q.mat[k]];
}
Color diffuse_color(Ctx& cx, const Prim& q, Vec3 point) {
    if (q.kind == PRIM_POLY && q.vertex_material) {  // poly.rs:91-94,198-203
        Areas w = weighted_areas(cx, q, point);
        return cmult(mat_at(cx, q, 0).diffuse, w.a2) + cmult(mat_at(cx, q, 1).diffuse, w.a1) +
               cmult(mat_at(cx, q, 2).diffuse, w.a0);
    }
    return mat_at(cx, q, 0).diffuse;  // poly.rs:201, sphere.rs:115-117
}
inline const Material& isect_material(Ctx& cx, const Isect& i) { return mat_at(cx, *i.prim, 0); }  // poly.rs:181-183
inline Vec3 isect_normal(Ctx& cx, const Isect& i) { return surface_normal(cx, *i.prim, i.ray.dir, isect_point(i)); }
inline Color isect_color(Ctx& cx, const Isect& i) { return diffuse_color(cx, *i.prim, isect_point(i)); }

Ray reflective_ray(Ctx& cx, const Isect& i) {  // intersection.rs:42-48
    Vec3 normal = isect_normal(cx, i);
    Vec3 d0 = invert(i.ray.dir);
    Vec3 origin = isect_point(i) + mult(normal, 0.0001f);
    Vec3 direction = mult(normal, dot(d0, normal) * 2.0f) - d0;
    return ray_init(origin, direction);
}
bool refractive_ray(Ctx& cx, const Isect& i, Ray* out) {  // intersection.rs:50-81
    Vec3 in_dir = i.ray.dir;
    Vec3 normal = isect_normal(cx, i);
    float n = i.ray.vacuum ? 1.0f / 1.5f : 1.5f / 1.0f;
    float cos_in = dot(normal, in_dir);
    if (cos_in > 0.0f) normal = invert(normal);
    float c = dot(in_dir, normal);
    float cos_phi_2 = 1.0f - n * n * (1.0f - c * c);
    if (cos_phi_2 < 0.0f) return false;  // total internal reflection
    float cos_phi = std::sqrt(cos_phi_2);
    Vec3 term1 = in_dir - mult(normal, c);
    term1 = mult(term1, n);
    Vec3 direction = term1 - mult(normal, cos_phi);
    Vec3 origin = isect_point(i) - mult(normal, 0.01f);
    *out = ray_init(origin, direction);
    out->vacuum = !out->vacuum;  // switch_medium (ray.rs:28-33): always false afterwards
    return true;
}

// ------------------------------------------------------------------ lib.rs --
const float SCALE = 10000.0f;  // lib.rs:22

struct Tracer {
    uint32_t width, height;
    Vec3 center, camera_pos, parallel_up, parallel_right;
    float vertical_fov, horizontal_fov;
};
Tracer setup_camera(const Camera& cam, uint32_t width, uint32_t height) {  // lib.rs:69-84
    Tracer t;
    t.width = width;
    t.height = height;
    t.parallel_right = cross(cam.view_dir, cam.ortho_up);
    t.parallel_up = cross(t.parallel_right, cam.view_dir);
    t.parallel_right = normalize(t.parallel_right);
    t.parallel_up = normalize(t.parallel_up);
    t.vertical_fov = cam.vertical_fov;
    t.horizontal_fov = cam.vertical_fov * ((float)width / (float)height);
    t.camera_pos = cam.pos;
    t.center = cam.pos + mult(cam.view_dir, SCALE);
    return t;
}
inline Vec3 vertical_plane(const Tracer& t) {  // lib.rs:86-89
    float f = std::tan(t.vertical_fov / 2.0f) * SCALE;
    return mult(t.parallel_up, f);
}
inline Vec3 horizontal_plane(const Tracer& t) {  // lib.rs:91-94
    float f = std::tan(t.horizontal_fov / 2.0f) * SCALE;
    return mult(t.parallel_right, f);
}
Ray compute_ray(const Tracer& t, float x, float y) {  // lib.rs:96-103
    x = x * (1.0f / (float)t.width);
    y = y * (1.0f / (float)t.height);
    Vec3 dx = mult(horizontal_plane(t), 2.0f * x - 1.0f);
    Vec3 dy = mult(vertical_plane(t), 2.0f * y - 1.0f);
    Vec3 dir = t.center + dx + dy;
    dir = normalize(dir);
    return ray_init(t.camera_pos, dir);
}

Color shadow_scalar(Ctx& cx, const Light& light, const Isect& isect, size_t n, size_t depth, RngKey key) {  // lib.rs:105-141
    if (depth <= 0) return color_new();
    Vec3 ori = isect_point(isect) + mult(isect_normal(cx, isect), 0.0001f);
    float shade = 0.0f;
    for (size_t s = 0; s < n; ++s) {
        RngKey ks = key_push(key, 0x100 + s);
        Vec3 dir = light_get_dir(light, ori, key_push(ks, 1));
        Ray shadow = ray_init(ori, dir);
        ++cx.st.shadow_rays;
        Isect h = scene_intersects(cx, shadow);
        float term;
        if (h.hit) {
            const Material& m = isect_material(cx, h);
            if (m.transparency == 0.0f) {
                if (light.type == RTX_LIGHT_DIRECTIONAL) term = 0.0f;
                else term = (distance(ori, isect_point(h)) > distance(ori, light_position(light, key_push(ks, 2)))) ? 1.0f : 0.0f;
            } else {
                term = m.transparency * shadow_scalar(cx, light, h, n, depth - 1, key_push(ks, 3)).r;
            }
        } else {
            term = 1.0f;
        }
        shade += term;
    }
    shade = shade / (float)n;
    return color_init(shade, shade, shade);
}
inline Color ambient_lightning(float kt, Color ka, Color cd) { return cmult(cd * ka, 1.0f - kt); }  // lib.rs:143-145
float calculate_fattj(const Light& light, Vec3 point, RngKey key) {                                // lib.rs:147-155
    if (light.type == RTX_LIGHT_DIRECTIONAL) return 1.0f;
    float d = distance(point, light_position(light, key));
    return std::fmin(1.0f, 1.0f / (0.25f + 0.1f * d + 0.01f * d * d));
}
inline Color diffuse_lightning(float kt, Color cd, Vec3 normal, Vec3 dj) {  // lib.rs:157-161
    float a = 1.0f - kt;
    float b = std::fmax(0.0f, dot(normal, dj));
    return cmult(cd, a * b);
}
inline Color specular_lightning(float q, Color ks, Vec3 normal, Vec3 dj, Vec3 v) {  // lib.rs:163-169
    float t = dot(normal, dj);
    Vec3 rj = mult(mult(normal, t), 2.0f);
    rj = rj - dj;
    t = std::fmax(dot(rj, v), 0.0f);
    return cmult(ks, std::pow(t, q));  // f32 powf
}
Color direct_lightning(Ctx& cx, const Light& light, const Isect& isect, Color sj, float fattj, size_t n, RngKey key) {  // lib.rs:171-199
    Vec3 point = isect_point(isect);
    const Material& material = isect_material(cx, isect);
    float kt = material.transparency;
    Color cd = isect_color(cx, isect);
    Color ks = material.specular;
    float q = material.shininess * 128.0f;
    Color direct_light = cmult(light.intensity * sj, fattj);
    Color lightning = color_new();
    for (size_t s = 0; s < n; ++s) {
        float nf = (float)n;
        Vec3 dir = light_get_dir(light, point, key_push(key, 0x200 + s));
        Vec3 normal = isect_normal(cx, isect);
        Color diffuse_light = diffuse_lightning(kt, cd, normal, dir);
        Vec3 v = invert(isect.ray.dir);
        Color specular_light = specular_lightning(q, ks, normal, dir, v);
        Color sample = direct_light * (diffuse_light + specular_light);
        sample = color_init(sample.r / nf, sample.g / nf, sample.b / nf);
        lightning = lightning + sample;
    }
    return lightning;
}

Color shade_intersection(Ctx& cx, const Isect& isect, size_t num_samples, size_t depth, RngKey key) {  // lib.rs:201-255
    if (depth <= 0) return color_new();
    ++cx.st.shaded_nodes;
    const Scene& sc = *cx.sc;
    const Material& material = isect_material(cx, isect);
    float kt = material.transparency;
    Color ks = material.specular;
    Color ka = material.ambient;
    Color cd = isect_color(cx, isect);
    Color ambient_light = ambient_lightning(kt, ka, cd);
    Color direct_light = color_new();
    for (size_t li = 0; li < sc.lights.size(); ++li) {
        const Light& light = sc.lights[li];
        RngKey kl = key_push(key, 0x1000 + li);
        float fattj = calculate_fattj(light, isect_point(isect), key_push(kl, 0));
        if (fattj > 0.0f) {
            size_t n = (light.type == RTX_LIGHT_AREA) ? num_samples : 1;
            Color ss = shadow_scalar(cx, light, isect, n, depth, key_push(kl, 1));
            direct_light = direct_light + direct_lightning(cx, light, isect, ss, fattj, n, key_push(kl, 2));
        }
    }
    Color reflective_light = color_new();
    if (scalar(ks) > 0.0f) {
        Ray ray = reflective_ray(cx, isect);
        ++cx.st.reflect_rays;
        if (depth - 1 == 0) ++cx.st.last_level_rays;
        Isect h = scene_intersects(cx, ray);
        if (h.hit) reflective_light = ks * shade_intersection(cx, h, num_samples, depth - 1, key_push(key, 0x2000));
    }
    Color refractive_light = color_new();
    if (kt > 0.0f) {
        Ray ray;
        if (refractive_ray(cx, isect, &ray)) {
            ++cx.st.refract_rays;
            if (depth - 1 == 0) ++cx.st.last_level_rays;
            Isect h = scene_intersects(cx, ray);
            if (h.hit) refractive_light = cmult(shade_intersection(cx, h, num_samples, depth - 1, key_push(key, 0x3000)), kt);
        }
    }
    return direct_light + ambient_light + reflective_light + refractive_light;
}

// --------------------------------------------------------- parser/mod.rs ---
struct ParseError {
    std::string msg;
};
struct Parser {
    std::string buf;
    size_t pos = 0;
    bool finished = false, peaked = false;
    std::string last;

    static bool is_ws(unsigned char c) {  // char::is_whitespace on a byte-as-char
        return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0;
    }
    std::string next_token() {  // parser/mod.rs:53-82
        if (peaked) {
            peaked = false;
            return last;
        }
        std::string tok;
        for (;;) {
            if (pos >= buf.size()) {
                finished = true;
                return tok;
            }
            unsigned char c = (unsigned char)buf[pos++];
            if (!is_ws(c)) tok.push_back((char)c);
            else if (!tok.empty()) return tok;
        }
    }
    std::string peak() {  // parser/mod.rs:40-51
        if (peaked) return last;
        last = next_token();
        peaked = true;
        return last;
    }
    void expect(const char* token) {  // check_and_consume, parser/mod.rs:96-99
        std::string t = next_token();
        if (t != token) throw ParseError{"expected '" + std::string(token) + "', got '" + t + "'"};
    }
    float next_f32() {  // next_num::<f32>, parser/mod.rs:84-90
        std::string t = next_token();
        char* end = nullptr;
        float v = std::strtof(t.c_str(), &end);
        if (t.empty() || end != t.c_str() + t.size()) throw ParseError{"Could not represent token as num: '" + t + "'"};
        return v;
    }
    long next_int() {
        std::string t = next_token();
        char* end = nullptr;
        long v = std::strtol(t.c_str(), &end, 10);
        if (t.empty() || end != t.c_str() + t.size()) throw ParseError{"Could not represent token as num: '" + t + "'"};
        return v;
    }
    float parse_f32(const char* name) { expect(name); return next_f32(); }  // :101-104
    Vec3 parse_vec3(const char* name) {                                     // :106-109
        expect(name);
        float x = next_f32(), y = next_f32(), z = next_f32();
        return V(x, y, z);
    }
    Color parse_color(const char* name) {                                   // :111-114 (clamps)
        expect(name);
        float r = next_f32(), g = next_f32(), b = next_f32();
        return color_init(r, g, b);
    }
    bool parse_bool(const char* name, const char* flag) {                   // :116-122
        expect(name);
        return next_token() == flag;
    }
    Light parse_light() {                                                   // :124-148
        std::string kw = next_token();
        expect("{");
        Light l{};
        if (kw == "point_light") {
            l.type = RTX_LIGHT_POINT;
            l.a = parse_vec3("position");
            l.intensity = parse_color("color");
        } else if (kw == "area_light") {
            l.type = RTX_LIGHT_AREA;
            l.a = parse_vec3("position");
            l.b = parse_vec3("position");
            l.intensity = parse_color("color");
        } else if (kw == "directional_light") {
            l.type = RTX_LIGHT_DIRECTIONAL;
            l.a = parse_vec3("direction");
            l.intensity = parse_color("color");
        } else {
            throw ParseError{"LightType is not valid: " + kw};
        }
        expect("}");
        return l;
    }
    Material parse_material() {                                             // :150-165
        expect("material");
        expect("{");
        Material m;
        m.diffuse = parse_color("diffColor");
        m.ambient = parse_color("ambColor");
        m.specular = parse_color("specColor");
        m.emissive = parse_color("emisColor");
        m.shininess = parse_f32("shininess");
        m.transparency = parse_f32("ktran");
        expect("}");
        return m;
    }
    Prim parse_sphere(Scene& sc) {                                          // :167-193
        expect("sphere");
        expect("{");
        expect("name");
        next_token();
        expect("numMaterials");
        long nm = next_int();
        Prim s{};
        s.kind = PRIM_SPHERE;
        s.mat[0] = s.mat[1] = s.mat[2] = RTX_MATERIAL_NONE;
        long k = 0;
        while (nm > 0) {
            Material m = parse_material();
            if (k == 0) s.mat[0] = (uint32_t)sc.materials.size();
            sc.materials.push_back(m);
            ++k;
            --nm;
        }
        s.nmat = (uint8_t)std::min<long>(k, 3);
        s.p[0] = parse_vec3("origin");
        s.radius = parse_f32("radius");
        parse_vec3("xaxis"); parse_f32("xlength");
        parse_vec3("yaxis"); parse_f32("ylength");
        parse_vec3("zaxis"); parse_f32("zlength");
        expect("}");
        return s;
    }
    void parse_vertex(Prim& q, int i, bool has_normal, bool has_material, uint32_t* mat_index) {  // :195-214
        q.p[i] = parse_vec3("pos");
        if (has_normal) q.n[i] = parse_vec3("norm");
        if (has_material) {
            expect("materialIndex");
            mat_index[i] = (uint32_t)next_int();
        }
    }
    void parse_polyset(Scene& sc, std::vector<Prim>& out) {                 // :216-298
        expect("poly_set");
        expect("{");
        expect("name");
        next_token();
        expect("numMaterials");
        long nm = next_int();
        uint32_t base = (uint32_t)sc.materials.size();
        uint32_t count = 0;
        while (nm > 0) {
            sc.materials.push_back(parse_material());
            ++count;
            --nm;
        }
        expect("type");
        next_token();
        bool per_vertex_normal = parse_bool("normType", "PER_VERTEX_NORMAL");
        bool material_binding = parse_bool("materialBinding", "PER_VERTEX_MATERIAL");
        expect("hasTextureCoords");
        next_token();
        expect("rowSize");
        next_token();
        expect("numPolys");
        long np = next_int();
        while (np > 0) {
            expect("poly");
            expect("{");
            expect("numVertices");
            next_token();  // always 3
            Prim q{};
            q.kind = PRIM_POLY;
            q.vertex_normal = per_vertex_normal;
            q.vertex_material = material_binding;
            q.mat[0] = q.mat[1] = q.mat[2] = RTX_MATERIAL_NONE;
            uint32_t mi[3] = {0, 0, 0};
            for (int i = 0; i < 3; ++i) parse_vertex(q, i, per_vertex_normal, material_binding, mi);
            expect("}");
            auto global = [&](uint32_t local) -> uint32_t {
                if (local >= count) throw ParseError{"materialIndex out of range"};  // reference: index panic
                return base + local;
            };
            if (material_binding) {  // de-duplicated list, :266-287
                q.mat[0] = global(mi[0]);
                q.nmat = 1;
                if (mi[1] != mi[0]) q.mat[q.nmat++] = global(mi[1]);
                if (mi[2] != mi[1] && mi[2] != mi[0]) q.mat[q.nmat++] = global(mi[2]);
            } else {                 // :288-290
                q.mat[0] = global(0);
                q.nmat = 1;
            }
            out.push_back(q);
            --np;
        }
        expect("}");
    }
    Camera parse_camera() {                                                 // :300-312
        expect("camera");
        expect("{");
        Camera c;
        c.pos = parse_vec3("position");
        c.view_dir = parse_vec3("viewDirection");
        c.focal_dist = parse_f32("focalDistance");
        c.ortho_up = parse_vec3("orthoUp");
        c.vertical_fov = parse_f32("verticalFOV");
        expect("}");
        return c;
    }
    void parse_scene(Scene& sc) {                                           // :314-346
        expect("Composer");
        expect("format");
        expect("2.1");
        expect("ascii");
        std::string tkn = peak();
        while (!finished) {
            if (tkn == "camera") {
                sc.camera = parse_camera();
            } else if (tkn == "sphere") {
                sc.prims.push_back(parse_sphere(sc));
            } else if (tkn == "poly_set") {
                std::vector<Prim> set;
                parse_polyset(sc, set);
                while (!set.empty()) {  // polyset.pop(): reversed order, :333-338
                    sc.prims.push_back(set.back());
                    set.pop_back();
                }
            } else if (tkn.size() >= 5 && tkn.compare(tkn.size() - 5, 5, "light") == 0) {
                sc.lights.push_back(parse_light());
            } else {
                throw ParseError{"Unexpected token: " + tkn};
            }
            tkn = peak();
        }
    }
};

bool read_file(const char* path, std::string* out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize((size_t)n);
    size_t got = n ? std::fread(&(*out)[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == (size_t)n;
}

void ensure_bvh(Scene& sc) {
    if (sc.bvh) return;
    std::vector<int32_t> ids(sc.prims.size());
    for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int32_t)i;
    sc.bvh = bvh_build(sc, ids, 0, ids.size(), 0);
}

inline Vec3 v3(const float* p) { return V(p[0], p[1], p[2]); }
inline void st3(float* p, Vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
inline Material from_rtx(const rtx_material& m) {
    Material o;
    o.diffuse = color_init(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    o.ambient = color_init(m.ambient[0], m.ambient[1], m.ambient[2]);
    o.specular = color_init(m.specular[0], m.specular[1], m.specular[2]);
    o.emissive = color_init(m.emissive[0], m.emissive[1], m.emissive[2]);
    o.shininess = m.shininess;
    o.transparency = m.transparency;
    return o;
}
inline rtx_material to_rtx(const Material& m) {
    rtx_material o;
    o.diffuse[0] = m.diffuse.r; o.diffuse[1] = m.diffuse.g; o.diffuse[2] = m.diffuse.b;
    o.ambient[0] = m.ambient.r; o.ambient[1] = m.ambient.g; o.ambient[2] = m.ambient.b;
    o.specular[0] = m.specular.r; o.specular[1] = m.specular.g; o.specular[2] = m.specular.b;
    o.emissive[0] = m.emissive.r; o.emissive[1] = m.emissive.g; o.emissive[2] = m.emissive.b;
    o.shininess = m.shininess;
    o.transparency = m.transparency;
    return o;
}

}  // namespace

struct orc_scene {
    Scene sc;
};

extern "C" {

const char* orc_last_error(void) { return g_error.c_str(); }

orc_scene* orc_scene_parse(const char* path) {
    auto s = std::make_unique<orc_scene>();
    Parser p;
    if (!path || !read_file(path, &p.buf)) {
        g_error = std::string("file error, path: ") + (path ? path : "(null)");
        return nullptr;
    }
    try {
        p.parse_scene(s->sc);
    } catch (const ParseError& e) {
        g_error = e.msg;
        return nullptr;
    }
    return s.release();
}

orc_scene* orc_scene_from_desc(const rtx_scene_desc* d) {
    if (!d) {
        g_error = "null descriptor";
        return nullptr;
    }
    auto s = std::make_unique<orc_scene>();
    Scene& sc = s->sc;
    sc.camera.pos = v3(d->camera.pos);
    sc.camera.view_dir = v3(d->camera.view_dir);
    sc.camera.ortho_up = v3(d->camera.ortho_up);
    sc.camera.focal_dist = d->camera.focal_dist;
    sc.camera.vertical_fov = d->camera.vertical_fov;
    for (uint32_t i = 0; i < d->n_lights; ++i) {
        Light l{};
        l.type = d->lights[i].type;
        l.a = v3(d->lights[i].a);
        l.b = v3(d->lights[i].b);
        l.intensity = color_init(d->lights[i].color[0], d->lights[i].color[1], d->lights[i].color[2]);
        sc.lights.push_back(l);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) sc.materials.push_back(from_rtx(d->materials[i]));
    size_t n = (size_t)d->n_tris + d->n_spheres;
    sc.prims.resize(n);
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t i = 0; i < d->n_tris; ++i) {
        size_t id = d->tri_prim_index ? d->tri_prim_index[i] : i;
        if (id >= n || seen[id]) {
            g_error = "prim_index is not a permutation";
            return nullptr;
        }
        seen[id] = 1;
        Prim q{};
        q.kind = PRIM_POLY;
        uint8_t fl = d->tri_flags ? d->tri_flags[i] : 0;
        q.vertex_normal = (fl & RTX_TRI_VERTEX_NORMAL) != 0;
        q.vertex_material = (fl & RTX_TRI_VERTEX_MATERIAL) != 0;
        for (int k = 0; k < 3; ++k) {
            q.p[k] = v3(d->tri_pos + (size_t)i * 9 + k * 3);
            q.n[k] = d->tri_normals ? v3(d->tri_normals + (size_t)i * 9 + k * 3) : V(0, 0, 0);
            q.mat[k] = d->tri_mat[(size_t)i * 3 + k];
        }
        q.nmat = 0;
        while (q.nmat < 3 && q.mat[q.nmat] != RTX_MATERIAL_NONE) ++q.nmat;
        sc.prims[id] = q;
    }
    for (uint32_t i = 0; i < d->n_spheres; ++i) {
        size_t id = d->sph_prim_index ? d->sph_prim_index[i] : (size_t)d->n_tris + i;
        if (id >= n || seen[id]) {
            g_error = "prim_index is not a permutation";
            return nullptr;
        }
        seen[id] = 1;
        Prim q{};
        q.kind = PRIM_SPHERE;
        q.p[0] = v3(d->sph_center_radius + (size_t)i * 4);
        q.radius = d->sph_center_radius[(size_t)i * 4 + 3];
        q.mat[0] = d->sph_mat[i];
        q.mat[1] = q.mat[2] = RTX_MATERIAL_NONE;
        q.nmat = 1;
        sc.prims[id] = q;
    }
    return s.release();
}

void orc_scene_free(orc_scene* s) { delete s; }

void orc_scene_counts(const orc_scene* s, uint32_t* n_prims, uint32_t* n_tris, uint32_t* n_spheres,
                      uint32_t* n_materials, uint32_t* n_lights) {
    uint32_t t = 0, sp = 0;
    for (const Prim& p : s->sc.prims) (p.kind == PRIM_POLY ? t : sp)++;
    if (n_prims) *n_prims = (uint32_t)s->sc.prims.size();
    if (n_tris) *n_tris = t;
    if (n_spheres) *n_spheres = sp;
    if (n_materials) *n_materials = (uint32_t)s->sc.materials.size();
    if (n_lights) *n_lights = (uint32_t)s->sc.lights.size();
}

void orc_scene_flatten(const orc_scene* s, float* tri_pos, float* tri_normals, uint32_t* tri_mat,
                       uint8_t* tri_flags, uint32_t* tri_prim_index, float* sph_center_radius,
                       uint32_t* sph_mat, uint32_t* sph_prim_index, rtx_material* materials,
                       rtx_light* lights, rtx_camera* camera) {
    const Scene& sc = s->sc;
    size_t ti = 0, si = 0;
    for (size_t id = 0; id < sc.prims.size(); ++id) {
        const Prim& p = sc.prims[id];
        if (p.kind == PRIM_POLY) {
            for (int k = 0; k < 3; ++k) {
                if (tri_pos) st3(tri_pos + ti * 9 + k * 3, p.p[k]);
                if (tri_normals) st3(tri_normals + ti * 9 + k * 3, p.n[k]);
                if (tri_mat) tri_mat[ti * 3 + k] = (k < p.nmat) ? p.mat[k] : RTX_MATERIAL_NONE;
            }
            if (tri_flags) tri_flags[ti] = (uint8_t)((p.vertex_normal ? RTX_TRI_VERTEX_NORMAL : 0) | (p.vertex_material ? RTX_TRI_VERTEX_MATERIAL : 0));
            if (tri_prim_index) tri_prim_index[ti] = (uint32_t)id;
            ++ti;
        } else {
            if (sph_center_radius) {
                st3(sph_center_radius + si * 4, p.p[0]);
                sph_center_radius[si * 4 + 3] = p.radius;
            }
            if (sph_mat) sph_mat[si] = p.mat[0];
            if (sph_prim_index) sph_prim_index[si] = (uint32_t)id;
            ++si;
        }
    }
    if (materials) for (size_t i = 0; i < sc.materials.size(); ++i) materials[i] = to_rtx(sc.materials[i]);
    if (lights) for (size_t i = 0; i < sc.lights.size(); ++i) {
        rtx_light l{};
        l.type = sc.lights[i].type;
        st3(l.a, sc.lights[i].a);
        st3(l.b, sc.lights[i].b);
        l.color[0] = sc.lights[i].intensity.r; l.color[1] = sc.lights[i].intensity.g; l.color[2] = sc.lights[i].intensity.b;
        lights[i] = l;
    }
    if (camera) {
        st3(camera->pos, sc.camera.pos);
        st3(camera->view_dir, sc.camera.view_dir);
        st3(camera->ortho_up, sc.camera.ortho_up);
        camera->focal_dist = sc.camera.focal_dist;
        camera->vertical_fov = sc.camera.vertical_fov;
    }
}

void orc_intersect_batch(orc_scene* s, uint64_t n, const float* ori3, const float* dir3,
                         int32_t* prim_id, float* t, int use_bvh) {
    if (use_bvh) ensure_bvh(s->sc);
    Ctx cx;
    cx.sc = &s->sc;
    cx.use_bvh = use_bvh != 0;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r = ray_init(v3(ori3 + i * 3), v3(dir3 + i * 3));
        Isect h = scene_intersects(cx, r);
        prim_id[i] = h.hit ? h.prim_id : -1;
        if (t) t[i] = h.hit ? h.t : 0.0f;
    }
}

void orc_render(orc_scene* s, uint32_t width, uint32_t height, uint32_t depth, uint32_t samples,
                uint32_t stride_x, uint32_t stride_y, int threads, int use_bvh,
                uint8_t* rgb, int32_t* prim_id, float* t, uint8_t* panic, float* ray_dir,
                orc_stats* stats) {
    if (use_bvh) ensure_bvh(s->sc);
    if (stride_x == 0) stride_x = 1;
    if (stride_y == 0) stride_y = 1;
    if (threads < 1) threads = 1;
    const Tracer tr = setup_camera(s->sc.camera, width, height);
    std::vector<orc_stats> partial((size_t)threads);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&](int tid) {
        Ctx cx;
        cx.sc = &s->sc;
        cx.use_bvh = use_bvh != 0;
        uint32_t row_no = 0;
        for (uint32_t y = 0; y < height; y += stride_y, ++row_no) {  // lib.rs:262 (row-major coordinates)
            if ((int)(row_no % (uint32_t)threads) != tid) continue;
            for (uint32_t x = 0; x < width; x += stride_x) {
                size_t pix = (size_t)y * width + x;
                Ray ray = compute_ray(tr, (float)x, (float)(height - y - 1));  // lib.rs:263
                ++cx.st.primary_rays;
                cx.panic = false;
                Isect h = scene_intersects(cx, ray);
                Color c = color_new();  // Image::new default pixel
                if (h.hit) c = shade_intersection(cx, h, samples, depth, RngKey{mix64(pix)});
                if (rgb) {
                    rgb[pix * 3 + 0] = to_u8(c.r);
                    rgb[pix * 3 + 1] = to_u8(c.g);
                    rgb[pix * 3 + 2] = to_u8(c.b);
                }
                if (prim_id) prim_id[pix] = h.hit ? h.prim_id : -1;
                if (t) t[pix] = h.hit ? h.t : 0.0f;
                if (panic) panic[pix] = cx.panic ? 1 : 0;
                if (ray_dir) st3(ray_dir + pix * 3, ray.dir);
                if (cx.panic) ++cx.st.panic_pixels;
            }
        }
        partial[(size_t)tid] = cx.st;
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(work, i);
        for (auto& th : pool) th.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        orc_stats o{};
        for (const orc_stats& p : partial) {
            o.primary_rays += p.primary_rays; o.shadow_rays += p.shadow_rays; o.reflect_rays += p.reflect_rays;
            o.refract_rays += p.refract_rays; o.shaded_nodes += p.shaded_nodes; o.last_level_rays += p.last_level_rays;
            o.panic_pixels += p.panic_pixels; o.prim_tests += p.prim_tests;
        }
        o.seconds = std::chrono::duration<double>(t1 - t0).count();
        *stats = o;
    }
}

// ------------------------------------------------------------ KAT hooks ----
void orc_kat_vec_ops(const float a[3], const float b[3], float add[3], float sub[3], float mul[3],
                     float cr[3], float* d, float* length_a, float normalized_a[3], float* dist) {
    Vec3 A = v3(a), B = v3(b);
    st3(add, A + B); st3(sub, A - B); st3(mul, A * B); st3(cr, cross(A, B));
    *d = dot(A, B); *length_a = length(A); st3(normalized_a, normalize(A)); *dist = distance(A, B);
}
float orc_kat_vec_area(const float a[3], const float b[3], const float c[3]) { return get_area(v3(a), v3(b), v3(c)); }
int orc_kat_sphere_intersect(const float center[3], float radius, const float ori[3], const float dir[3], float* t) {
    Prim s{};
    s.kind = PRIM_SPHERE; s.p[0] = v3(center); s.radius = radius;
    Hit h = sphere_intersects(s, ray_init(v3(ori), v3(dir)));
    *t = h.t;
    return h.hit;
}
int orc_kat_poly_intersect(const float v[9], const float ori[3], const float dir[3], float* t) {
    Prim q{};
    q.kind = PRIM_POLY;
    for (int k = 0; k < 3; ++k) q.p[k] = v3(v + 3 * k);
    Hit h = poly_intersects(q, ray_init(v3(ori), v3(dir)));
    *t = h.t;
    return h.hit;
}
void orc_kat_sphere_bbox(const float center[3], float radius, float bmin[3], float bmax[3]) {
    Prim s{};
    s.kind = PRIM_SPHERE; s.p[0] = v3(center); s.radius = radius;
    BBox b = prim_bbox(s);
    st3(bmin, b.min); st3(bmax, b.max);
}
void orc_kat_poly_bbox(const float v[9], float bmin[3], float bmax[3]) {
    Prim q{};
    q.kind = PRIM_POLY;
    for (int k = 0; k < 3; ++k) q.p[k] = v3(v + 3 * k);
    BBox b = prim_bbox(q);
    st3(bmin, b.min); st3(bmax, b.max);
}
int orc_kat_bbox_intersect(const float bmin[3], const float bmax[3], const float ori[3], const float dir[3]) {
    return bbox_intersects(BBox{v3(bmin), v3(bmax)}, ray_init(v3(ori), v3(dir)));
}
void orc_kat_bbox_centroid(const float bmin[3], const float bmax[3], float c[3]) { st3(c, bbox_centroid(BBox{v3(bmin), v3(bmax)})); }
int orc_kat_bbox_less(const float amin[3], const float amax[3], const float bmin[3], const float bmax[3]) {
    return vec_lt(bbox_centroid(BBox{v3(amin), v3(amax)}), bbox_centroid(BBox{v3(bmin), v3(bmax)}));
}
void orc_kat_color_init(float r, float g, float b, float out[3]) {
    Color c = color_init(r, g, b);
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
void orc_kat_compute_ray(const rtx_camera* cam, uint32_t width, uint32_t height, float x, float y, float ori[3], float dir[3]) {
    Camera c;
    c.pos = v3(cam->pos); c.view_dir = v3(cam->view_dir); c.ortho_up = v3(cam->ortho_up);
    c.focal_dist = cam->focal_dist; c.vertical_fov = cam->vertical_fov;
    Ray r = compute_ray(setup_camera(c, width, height), x, y);
    st3(ori, r.ori); st3(dir, r.dir);
}
static void kat_spheres(Scene& sc, const float* cr, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        Prim s{};
        s.kind = PRIM_SPHERE; s.p[0] = v3(cr + 4 * i); s.radius = cr[4 * i + 3];
        s.mat[0] = 0; s.nmat = 1;
        sc.prims.push_back(s);
    }
    sc.materials.push_back(Material{});
    ensure_bvh(sc);
}
int orc_kat_bvh_node(const float* cr, uint32_t n, const char* path, float bmin[3], float bmax[3], int32_t* leaf_index) {
    Scene sc;
    kat_spheres(sc, cr, n);
    const BvhNode* node = sc.bvh.get();
    for (const char* p = path; *p && node; ++p) node = (*p == 'L') ? node->left.get() : node->right.get();
    if (!node) return 0;
    st3(bmin, node->bbox.min); st3(bmax, node->bbox.max);
    *leaf_index = node->prim;
    return node->kind;
}
int orc_kat_bvh_intersect(const float* cr, uint32_t n, const float ori[3], const float dir[3], float* t) {
    Scene sc;
    kat_spheres(sc, cr, n);
    uint64_t tests = 0;
    NodeHit h = bvh_intersect(sc, sc.bvh.get(), ray_init(v3(ori), v3(dir)), &tests);
    if (!h.node) return -1;
    *t = h.t;
    return h.node->prim;
}
int orc_kat_secondary_rays(const orc_scene* s, const float ori[3], const float dir[3], int in_vacuum,
                           float refl_ori[3], float refl_dir[3], int* has_refr, float refr_ori[3], float refr_dir[3]) {
    Ctx cx;
    cx.sc = &s->sc;
    cx.use_bvh = false;
    Ray r = ray_init(v3(ori), v3(dir));
    r.vacuum = in_vacuum != 0;
    Isect h = scene_intersects(cx, r);
    if (!h.hit) return -1;
    Ray rr = reflective_ray(cx, h);
    st3(refl_ori, rr.ori); st3(refl_dir, rr.dir);
    Ray fr;
    *has_refr = refractive_ray(cx, h, &fr) ? 1 : 0;
    if (*has_refr) { st3(refr_ori, fr.ori); st3(refr_dir, fr.dir); }
    return h.prim_id;
}

}  // extern "C"
