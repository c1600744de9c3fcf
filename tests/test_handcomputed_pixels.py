"""Known-answer pixels worked out here, in numpy float32, straight from the reference's formulas -- not through the
oracle's code: compute_ray (src/lib.rs:96-103), Sphere::intersects (sphere.rs:54-103), shadow_scalar / fattj /
diffuse / specular / direct_lightning / ambient (lib.rs:105-199), the clamped Color arithmetic (material.rs:17-84)
and as_pixel (material.rs:57-63). The oracle must reproduce them (CPU), and so must the device (GPU)."""
import importlib

import numpy as np
import pytest

f32 = np.float32


def _clamp(c):
    return np.minimum(np.maximum(c, f32(0.0)), f32(1.0)).astype(np.float32)


def _scene(rtx, spec, shininess):
    flat = rtx.FlatScene()
    flat.sph_center_radius = np.array([[0.0, 0.0, -5.0, 1.0]], np.float32)
    m = flat.add_material((1.0, 0.25, 0.0), (0.1, 0.1, 0.1), (spec, spec, spec), shininess, 0.0)
    flat.sph_mat = np.array([m], np.uint32)
    flat.add_light(rtx.LIGHT_POINT, (2.0, 2.0, 0.0), (1.0, 0.5, 1.0))
    flat.set_camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), np.pi / 2)
    return flat


def _by_hand(spec, shininess):
    """The pixel whose primary ray is compute_ray(1, 1) of a 2x2 image: straight down -z."""
    # compute_ray: x' = 1 * (1/2), 2x' - 1 = 0 -> dir = normalize(center) = (0, 0, -1); ori = (0, 0, 0)
    o = np.zeros(3, np.float32)
    d = np.array([0, 0, -1], np.float32)
    # Sphere::intersects with c = (0,0,-5), r = 1
    c, r = np.array([0, 0, -5], np.float32), f32(1.0)
    oc = o - c
    a = f32(np.dot(d, d))
    b = f32(2.0) * f32(d[0] * oc[0] + d[1] * oc[1] + d[2] * oc[2])
    cc = f32(oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2]) - r * r
    disc = b * b - f32(4.0) * a * cc
    q = (-b - np.sqrt(disc)) / f32(2.0) if b < 0 else (-b + np.sqrt(disc)) / f32(2.0)
    t0, t1 = q / a, cc / q
    if t0 > t1:
        t0, t1 = t1, t0
    t = t1 if t0 < 0 else t0
    assert t == f32(4.0)                                   # the reference's own test pins this (sphere.rs:133-143)
    p = (o + d * t).astype(np.float32)                     # (0, 0, -4)
    n = (p - c).astype(np.float32)
    n = (n / np.sqrt(f32(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]))).astype(np.float32)    # (0, 0, 1)
    light, inten = np.array([2, 2, 0], np.float32), np.array([1.0, 0.5, 1.0], np.float32)
    cd, ka = np.array([1.0, 0.25, 0.0], np.float32), np.array([0.1, 0.1, 0.1], np.float32)
    ks, kt = np.array([spec] * 3, np.float32), f32(0.0)
    # fattj (lib.rs:147-155)
    dv = (p - light).astype(np.float32)
    dist = np.sqrt(f32(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]))
    fattj = min(f32(1.0), f32(1.0) / (f32(0.25) + f32(0.1) * dist + f32(0.01) * dist * dist))
    # shadow ray from p + n * 1e-4 towards the light leaves the sphere: shadow_scalar = 1 (lib.rs:133)
    sj = np.array([1, 1, 1], np.float32)
    lv = (light - p).astype(np.float32)
    L = (lv / np.sqrt(f32(lv[0] * lv[0] + lv[1] * lv[1] + lv[2] * lv[2]))).astype(np.float32)
    ndl = f32(n[0] * L[0] + n[1] * L[1] + n[2] * L[2])
    diffuse = _clamp(cd * ((f32(1.0) - kt) * max(f32(0.0), ndl)))                         # lib.rs:157-161
    rj = (n * ndl * f32(2.0) - L).astype(np.float32)                                      # lib.rs:163-169
    v = (-d).astype(np.float32)
    rv = max(f32(rj[0] * v[0] + rj[1] * v[1] + rj[2] * v[2]), f32(0.0))
    specular = _clamp(ks * f32(np.power(rv, f32(shininess) * f32(128.0), dtype=np.float32)))
    direct_light = _clamp(_clamp(inten * sj) * fattj)                                     # lib.rs:171-199, one sample
    direct = _clamp(_clamp(direct_light * _clamp(diffuse + specular)) / f32(1.0))
    ambient = _clamp(_clamp(cd * ka) * (f32(1.0) - kt))                                   # lib.rs:143-145
    colour = _clamp(_clamp(np.zeros(3, np.float32) + direct) + ambient)                   # lib.rs:225,254 (no children: + black)
    return (colour * f32(255.0)).astype(np.uint8), float(rv)                              # as_pixel truncates


@pytest.mark.parametrize("spec,shininess", [(0.0, 0.2), (0.5, 0.03125)])
def test_oracle_reproduces_hand_computed_pixel(rtx, orc, spec, shininess):
    want, rv = _by_hand(spec, shininess)
    assert want[0] > 100 and want[2] == (0 if spec == 0.0 else want[2])
    ref = orc.OracleScene.from_flat(_scene(rtx, spec, shininess)).render(2, 2, 10, 10)
    # pixel (x, y) is the ray through (x, height - y - 1) (lib.rs:263): compute_ray(1, 1) is pixel (1, 0)
    assert ref["prim_id"][0, 1] == 0 and ref["t"][0, 1] == np.float32(4.0)
    assert np.abs(ref["rgb"][0, 1].astype(int) - want.astype(int)).max() <= (0 if spec == 0.0 else 1)   # powf: libm vs numpy


@pytest.mark.gpu
@pytest.mark.parametrize("spec,shininess", [(0.0, 0.2), (0.5, 0.03125)])
def test_device_reproduces_hand_computed_pixel(rtx, gpu_ctx_factory, spec, shininess):
    want, _ = _by_hand(spec, shininess)
    ctx = gpu_ctx_factory(_scene(rtx, spec, shininess))
    rgb, _ = ctx.render(2, 2, 10, 10)
    prim, t, _ = ctx.primary_hits(2, 2)
    assert prim[0, 1] == 0 and t[0, 1] == np.float32(4.0)
    assert np.abs(rgb[0, 1].astype(int) - want.astype(int)).max() <= (0 if spec == 0.0 else 1)
