"""Pins the CPU oracle with every known-answer value held by the reference's own
unit tests (one test here per #[test] there; the citation is the reference test)."""
import ctypes as C
import math

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def f3(*v):
    return (C.c_float * 3)(*v)


def vec_ops(L, a, b):
    out = {k: f3(0, 0, 0) for k in ("add", "sub", "mul", "cross", "norm")}
    dot, length, dist = C.c_float(), C.c_float(), C.c_float()
    L.orc_kat_vec_ops(f3(*a), f3(*b), out["add"], out["sub"], out["mul"], out["cross"], C.byref(dot), C.byref(length),
                      out["norm"], C.byref(dist))
    r = {k: tuple(v) for k, v in out.items()}
    r.update(dot=dot.value, length=length.value, dist=dist.value)
    return r


@pytest.fixture(scope="module")
def L(orc):
    return orc.lib()


def test_vec3_add_sub_mul_mult(L):   # src/vec.rs:141-178
    r = vec_ops(L, (0, 1, 2), (0, 1, 2))
    assert r["add"] == (0, 2, 4)
    assert r["sub"] == (0, 0, 0)
    assert r["mul"] == (0, 1, 4)


def test_vec3_length(L):             # src/vec.rs:190-196
    x = vec_ops(L, (1.2, 2.2, 3.2), (0, 0, 0))["length"]
    assert x - 4.06448 > 0 and x - 4.06449 < 0


def test_vec3_normalize(L):          # src/vec.rs:198-204
    x = vec_ops(L, (3, 4, 5), (0, 0, 0))["norm"][0]
    assert x - 0.424264 > 0 and x - 0.424265 < 0


def test_vec3_cross(L):              # src/vec.rs:206-215
    assert vec_ops(L, (1, 2, 3), (3, 4, 5))["cross"] == (-2, 4, -2)


def test_sphere_intersect_t_is_4(L):  # src/scene/shapes/sphere.rs:133-143
    t = C.c_float()
    assert L.orc_kat_sphere_intersect(f3(0, 0, -5), C.c_float(1.0), f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 1
    assert t.value == 4.0


def test_poly_intersect(L):          # src/scene/shapes/poly.rs:232-244
    v = (C.c_float * 9)(2, 0, -3, -2, 0, -3, 0, 2, -1)
    t = C.c_float()
    assert L.orc_kat_poly_intersect(v, f3(0, 0.7071067812, 0), f3(0, 0, -1), C.byref(t)) == 1
    assert abs(t.value - 2.292893) < 1e-6


def test_bbox_from_sphere(L):        # src/scene/shapes/mod.rs:182-188
    lo, hi = f3(0, 0, 0), f3(0, 0, 0)
    L.orc_kat_sphere_bbox(f3(1, 1, 1), C.c_float(2.0), lo, hi)
    assert tuple(lo) == (-1, -1, -1) and tuple(hi) == (3, 3, 3)


def test_bbox_intersect_behind_ray_is_true(L):   # src/scene/shapes/mod.rs:190-197
    assert L.orc_kat_bbox_intersect(f3(-1, -1, -1), f3(3, 3, 3), f3(0, 0, -2), f3(0, 0, -1)) == 1


def test_bbox_compare_by_centroid(L):  # src/scene/shapes/mod.rs:199-205
    assert L.orc_kat_bbox_less(f3(-1, 0, 0), f3(0, 1, 1), f3(0, 0, 0), f3(0, 1, 1)) == 1


def test_color_clamps(L):            # src/scene/material.rs:127-144
    out = f3(9, 9, 9)
    L.orc_kat_color_init(C.c_float(0), C.c_float(0), C.c_float(0), out)
    assert tuple(out) == (0, 0, 0)
    L.orc_kat_color_init(C.c_float(2.0), C.c_float(0.5), C.c_float(-1.0), out)
    assert tuple(out) == (1.0, 0.5, 0.0)


def test_compute_ray(L, orc):        # src/lib.rs:312-324
    cam = orc.RtxCamera()
    cam.view_dir[:] = [0, 0, -1]
    cam.ortho_up[:] = [0, 1, 0]
    cam.vertical_fov = np.float32(np.float32(math.pi) / np.float32(2.0))
    ori, d = f3(1, 1, 1), f3(0, 0, 0)
    L.orc_kat_compute_ray(C.byref(cam), 2, 2, C.c_float(0), C.c_float(0), ori, d)
    assert all(abs(x) < 1e-6 for x in ori)
    assert all(abs(x + 0.57735) < 1e-6 for x in d)


SPHERES4 = np.array([[0, 0, 0, 1], [-1, 2, 1, 1], [-2, -2, 2, 1], [2, 2, -1, 1]], np.float32)


def bvh_node(L, centers, path):
    lo, hi, leaf = f3(0, 0, 0), f3(0, 0, 0), C.c_int32(-7)
    kind = L.orc_kat_bvh_node(centers.ctypes.data_as(C.c_void_p), len(centers), path.encode(), lo, hi, C.byref(leaf))
    return kind, tuple(lo), tuple(hi), leaf.value


def test_bvh_tree_of_size_1(L):      # src/scene/bvh.rs:161-187
    one = np.array([[0, 0, -5, 1]], np.float32)
    kind, lo, hi, leaf = bvh_node(L, one, "")
    assert kind == 2 and leaf == 0
    c = f3(0, 0, 0)
    L.orc_kat_bbox_centroid(f3(*lo), f3(*hi), c)
    assert tuple(c) == (0, 0, -5)
    t = C.c_float()
    assert L.orc_kat_bvh_intersect(one.ctypes.data_as(C.c_void_p), 1, f3(0, 0, 0), f3(0, 0, -1), C.byref(t)) == 0
    assert t.value == 4.0


def test_bvh_build_tree_of_size_4(L):   # src/scene/bvh.rs:189-233
    assert bvh_node(L, SPHERES4, "")[:3] == (1, (-3, -3, -2), (3, 3, 3))
    assert bvh_node(L, SPHERES4, "L")[:3] == (1, (-3, -3, 0), (0, 3, 3))
    assert bvh_node(L, SPHERES4, "R")[:3] == (1, (-1, -1, -2), (3, 3, 1))
    # leaves: ll = (-2,-2,2) lr = (-1,2,1) rl = (0,0,0) rr = (2,2,-1)
    assert [bvh_node(L, SPHERES4, p)[3] for p in ("LL", "LR", "RL", "RR")] == [2, 1, 0, 3]
    assert all(bvh_node(L, SPHERES4, p)[0] == 2 for p in ("LL", "LR", "RL", "RR"))


def test_bvh_intersect_tree_of_size_4(L):   # src/scene/bvh.rs:235-269
    t = C.c_float()
    ptr = SPHERES4.ctypes.data_as(C.c_void_p)
    assert L.orc_kat_bvh_intersect(ptr, 4, f3(2, 2, 2), f3(0, 0, -1), C.byref(t)) == 3
    assert L.orc_kat_bvh_intersect(ptr, 4, f3(-1, -1, 1), f3(-1, -1, 1), C.byref(t)) == 2
    assert L.orc_kat_bvh_intersect(ptr, 4, f3(-1, -1, 1), f3(0, 0, 1), C.byref(t)) == -1


def test_scene_intersect_colour(orc):   # src/scene/mod.rs:270-281: one red sphere, hit colour (1,0,0)
    f = orc.FlatScene()
    f.sph_center_radius = np.array([[0, 0, -5, 1]], np.float32)
    f.sph_mat = np.array([0], np.uint32)
    f.materials = np.zeros((1,), orc.MATERIAL_DTYPE)
    f.materials["diffuse"][0] = (1, 0, 0)
    f.materials["ambient"][0] = (1, 1, 1)   # ambient term = diffuse colour, so the frame shows it
    f.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 0.5)
    s = orc.OracleScene.from_flat(f)
    prim, t = s.intersect_batch([[0, 0, 0]], [[0, 0, -1]])
    assert prim[0] == 0 and t[0] == 4.0
    out = s.render(3, 3, 1, 1)
    assert tuple(out["rgb"][1, 1]) == (255, 0, 0)


def test_scene_intersect_keeps_the_first_hit_even_if_it_is_nan(orc):
    """src/scene/mod.rs:188-192 takes the first Hit unconditionally and replaces it only on `new_point < point`. A ray with a
    NaN direction -- Light::get_dir of a point that IS the light normalises a zero vector, scene/mod.rs:35-59 -- misses every
    triangle (`t > 1e-7` is false, poly.rs:176) and gets Hit(NaN) from every sphere (no comparison of sphere.rs:70-102 is
    true), so the loop returns the scene's first sphere, whatever lies where. The oracle restates that; the library documents
    that it ignores Hit(NaN) instead (a shadow ray that starts at a point light sees the light there, not the first sphere)."""
    f = orc.FlatScene()
    f.tri_pos = np.array([[[-1, -1, -3], [1, -1, -3], [0, 1, -3]]], np.float32)
    f.tri_mat = np.array([[0, 0xFFFFFFFF, 0xFFFFFFFF]], np.uint32)      # materials[0] only (RTX_MATERIAL_NONE elsewhere)
    f.sph_center_radius = np.array([[50, 50, 50, 1], [0, 0, -5, 1]], np.float32)
    f.sph_mat = np.array([0, 0], np.uint32)
    f.materials = np.zeros((1,), orc.MATERIAL_DTYPE)
    f.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 0.5)
    s = orc.OracleScene.from_flat(f)
    nan = float("nan")
    prim, t = s.intersect_batch([[0, 0, 0], [0, 0, 0]], [[nan, nan, nan], [0, 0, -1]])
    assert prim[0] == 1 and np.isnan(t[0])        # primitive 1 = the first sphere, nowhere near the origin
    assert prim[1] == 0 and t[1] == 3.0           # the same scene with a real ray: the triangle in front
    g = orc.FlatScene()
    g.tri_pos, g.tri_mat, g.materials = f.tri_pos, f.tri_mat, f.materials
    g.set_camera((0, 0, 0), (0, 0, -1), (0, 1, 0), 0.5)
    prim, _ = orc.OracleScene.from_flat(g).intersect_batch([[0, 0, 0]], [[nan, nan, nan]])
    assert prim[0] == -1                          # without a sphere such a ray misses


def test_parser_fixture_scene(orc):  # src/scene/parser/test/mod.rs:205-211
    s = orc.OracleScene.parse(GOLDEN_DIR + "/parser/testdata-scene.txt")
    c = s.counts()
    assert c["n_lights"] == 3 and c["n_prims"] == 13


def test_parser_bundled_scene_inventory(orc):
    """Primitive / light counts of the bundled scenes (counted independently from the files)."""
    from conftest import scene_path
    expect = {"sphere": (1, 0, 3), "test00": (4, 0, 0), "test01": (1, 12, 3), "test02": (7, 0, 1), "test03": (0, 96, 3),
              "test04": (11, 36, 2), "test05": (4, 24, 2), "test06": (0, 356, 2), "test07": (0, 5951, 2),
              "test08": (1, 5168, 3), "test09": (0, 10365, 4), "test10": (1, 5338, 10), "test11": (1, 12, 0)}
    for name, (ns, nt, nl) in expect.items():
        c = orc.OracleScene.parse(scene_path(name)).counts()
        assert (c["n_spheres"], c["n_tris"], c["n_lights"]) == (ns, nt, nl), name


def test_poly_set_is_reversed(orc):  # src/scene/parser/mod.rs:333-338
    from conftest import scene_path
    f = orc.OracleScene.parse(scene_path("test01")).flatten()
    # test01: sphere is primitive 0; the poly_set's LAST poly in the file becomes primitive 1
    assert list(f.sph_prim_index) == [0]
    assert np.allclose(f.tri_pos[0], [[-6.19481, -2.52771, 0.67082], [3.11895, -2.52771, -8.92172], [3.11895, -2.52771, 0.67082]])
    assert np.allclose(f.tri_pos[-1], [[-6.19481, -2.18127, 0.67082], [-6.19481, -2.52771, 0.67082], [3.11895, -2.52771, 0.67082]])
