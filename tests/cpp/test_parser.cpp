// Mirrors the reference's parser unit tests (src/scene/parser/test/mod.rs:15-211)
// against the C++ host-side SceneParser, reading the same fixture files
// (tests/golden/parser/testdata-*.txt). One function per reference #[test].
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rstracer.hpp"

using namespace rstracer;
static std::string g_prefix;
static int g_passed = 0;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)
static SceneParser scene_parser(const char* name) { return SceneParser(g_prefix + name + ".txt"); }

static void can_parse_tokens() {            // test/mod.rs:15-36
    SceneParser p = scene_parser("light");
    CHECK(p.next_token() == "point_light");
    CHECK(p.next_token() == "{");
    CHECK(p.next_token() == "position");
    CHECK(p.next_token() == "-1");
    std::string tkn;
    while (p.has_next_token()) tkn = p.next_token();
    CHECK(tkn == "}");
    ++g_passed;
}
static void can_peak_at_next_token() {      // test/mod.rs:38-56
    SceneParser p = scene_parser("light");
    CHECK(p.next_token() == "point_light");
    CHECK(p.peak() == "{");
    CHECK(p.peak() == "{");
    CHECK(p.next_token() == "{");
    CHECK(p.next_token() == "position");
    ++g_passed;
}
static void can_parse_f32() {               // test/mod.rs:58-66
    SceneParser p = scene_parser("f32");
    CHECK(p.next_f32() == 1.5f);
    CHECK(p.next_f32() == -0.5f);
    ++g_passed;
}
static void can_parse_position() {          // test/mod.rs:68-75
    SceneParser p = scene_parser("position");
    Vec3 v = p.parse_vec3("position");
    CHECK(v.x == -1.0f && v.y == 0.0f && v.z == 2.0f);
    ++g_passed;
}
static void can_parse_color() {             // test/mod.rs:77-84
    SceneParser p = scene_parser("color");
    Color c = p.parse_color("color");
    CHECK(c.r == 1.0f && c.g == 0.0f && c.b == 0.5f);
    ++g_passed;
}
static void can_parse_light() {             // test/mod.rs:86-114
    SceneParser p = scene_parser("light");
    Light a = p.parse_light();
    CHECK(a.kind == Light::Point && a.pos.x == -1.0f && a.intensity.r == 1.0f);
    Light b = p.parse_light();
    CHECK(b.kind == Light::Area && b.pos.x == 0.0f && b.max.x == 200.0f && b.intensity.r == 0.0f);
    Light c = p.parse_light();
    CHECK(c.kind == Light::Directional && c.pos.x == 0.5f && c.intensity.r == 0.5f);
    ++g_passed;
}
static void can_parse_material() {          // test/mod.rs:116-124
    SceneParser p = scene_parser("material");
    Material m = p.parse_material();
    CHECK(m.diffuse.r == 0.56f && m.ambient.r == 0.2f && m.shininess == 0.2f && m.transparency == 0.5f);
    ++g_passed;
}
static void can_parse_sphere() {            // test/mod.rs:126-133
    SceneParser p = scene_parser("sphere");
    Sphere s = p.parse_sphere();
    CHECK(s.materials.size() == 1 && s.origin.y == -0.5f && s.radius == 1.5f);
    ++g_passed;
}
static void can_parse_poly() {              // test/mod.rs:135-142
    SceneParser p = scene_parser("polygon");
    Poly q = p.parse_poly(false, false);
    CHECK(q.vertices[0].position.x == 0.0f && q.vertices[1].position.x == 0.5f && q.vertices[2].position.x == 10.0f);
    ++g_passed;
}
static void can_parse_polyset() {           // test/mod.rs:144-154
    SceneParser p = scene_parser("polyset");
    std::vector<Poly> set = p.parse_polyset();
    CHECK(set.size() == 12);
    CHECK(!set[0].vertex_material && !set[0].vertex_normal && set[0].materials.size() == 1);
    ++g_passed;
}
static bool col_eq(const Color& c, float r, float g, float b) { return c.r == r && c.g == g && c.b == b; }
static void can_parse_per_vertex_polyset() {   // test/mod.rs:156-192
    SceneParser p = scene_parser("per-vertex-polyset");
    std::vector<Poly> set = p.parse_polyset();
    CHECK(set.size() == 3);
    const Poly& a = set[0];
    CHECK(a.vertex_material && a.vertex_normal && a.materials.size() == 3);
    CHECK(col_eq(a.materials[0].diffuse, 0, 0, 0) && a.vertices[0].mat_index == 0);
    CHECK(col_eq(a.materials[1].diffuse, 0, 0, 1) && a.vertices[1].mat_index == 1);
    CHECK(col_eq(a.materials[2].diffuse, 0, 1, 0) && a.vertices[2].mat_index == 2);
    const Poly& b = set[1];
    CHECK(b.vertex_material && b.vertex_normal && b.materials.size() == 2);
    CHECK(col_eq(b.materials[0].diffuse, 0, 1, 1) && b.vertices[0].mat_index == 0 && b.vertices[1].mat_index == 0);
    CHECK(col_eq(b.materials[1].diffuse, 1, 0, 0) && b.vertices[2].mat_index == 1);
    const Poly& c = set[2];
    CHECK(c.vertex_material && c.vertex_normal && c.materials.size() == 2);
    CHECK(col_eq(c.materials[0].diffuse, 1, 0, 1) && c.vertices[0].mat_index == 0);
    CHECK(col_eq(c.materials[1].diffuse, 1, 0, 0) && c.vertices[1].mat_index == 1 && c.vertices[2].mat_index == 0);
    ++g_passed;
}
static void can_parse_camera() {            // test/mod.rs:194-203
    SceneParser p = scene_parser("camera");
    Camera c = p.parse_camera();
    CHECK(c.pos.x == 1.0f && c.view_dir.x == -1.0f && c.focal_dist == 12.0f && c.ortho_up.x == 2.0f && c.vertical_fov == 0.5f);
    ++g_passed;
}
static void can_parse_scene() {             // test/mod.rs:205-211
    SceneParser p = scene_parser("scene");
    Scene s = p.parse_scene();
    CHECK(s.lights.size() == 3 && s.primitives.size() == 13);
    ++g_passed;
}

int main(int argc, char** argv) {
    g_prefix = argc > 1 ? argv[1] : "tests/golden/parser/testdata-";
    can_parse_tokens();
    can_peak_at_next_token();
    can_parse_f32();
    can_parse_position();
    can_parse_color();
    can_parse_light();
    can_parse_material();
    can_parse_sphere();
    can_parse_poly();
    can_parse_polyset();
    can_parse_per_vertex_polyset();
    can_parse_camera();
    can_parse_scene();
    std::printf("%d tests passed\n", g_passed);
    return 0;
}
