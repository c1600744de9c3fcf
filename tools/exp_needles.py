#!/usr/bin/env python3
"""How far from its own plane does the reference's triangle test (poly.rs:138-179) report the hit of a needle -- a triangle
whose edges at v0 are 1-2 mrad apart -- and is that inside the padding of its box? Needles in planes y = const (so the
error normal to the plane is one subtraction), each with an exact copy at a higher index, rays from all over the safe
cube. Prints the error against the standard padding (2^-17 of the coordinates, box side plus ray side) and against the
needle padding of the build (32 u * 2.2 safe_r / sin(theta0), rtx_bvh.hip), and the tree-against-loop mismatches with
the work sharing of the traversal off, default and on from the first step.   usage: python tools/exp_needles.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
rng = np.random.default_rng(1)
flat = synth.heightfield_scene(48)
k = 3000
v0 = rng.uniform(-1.0, 1.0, (k, 3)) * np.array([1.0, 0.2, 1.0]) + np.array([0.0, 0.45, 0.0])
u = rng.normal(size=(k, 3)); u[:, 1] = 0; u /= np.linalg.norm(u, axis=1, keepdims=True)
w = np.cross(u, np.array([0.0, 1.0, 0.0]))
theta = 10 ** rng.uniform(np.log10(1.02e-3), np.log10(2e-3), (k, 1))
length = 10 ** rng.uniform(-2.0, -1.0, (k, 1))
e1 = u * length
e2 = -(u * np.cos(theta) + w * np.sin(theta)) * length * rng.uniform(0.8, 1.0, (k, 1))
needles = np.stack([v0, v0 + e1, v0 + e2], axis=1).astype(np.float32)
n0 = flat.n_tris
flat.tri_pos = np.concatenate([flat.tri_pos, needles, needles])
flat.tri_mat = np.concatenate([flat.tri_mat, np.tile(flat.tri_mat[:1], (2 * k, 1))])
flat.tri_flags = np.concatenate([flat.tri_flags, np.zeros((2 * k,), np.uint8)])
n = 400000
which = rng.integers(0, k, n)
bary = rng.dirichlet((1, 1, 1), n)
target = (needles[which].astype(np.float64) * bary[:, :, None]).sum(axis=1)
ori = (rng.uniform(-3.0, 3.0, (n, 3)) + np.array([0.0, 1.0, 0.0])).astype(np.float32)
d = (target - ori).astype(np.float32)
d *= (10 ** rng.uniform(-1, 1, (n, 1))).astype(np.float32)
with rtx.Context(0) as ctx:
    ctx.set_scene(flat)
    st = ctx.build()
    print(st)
    px, tx = ctx.intersect_batch(ori, d, exhaustive=True)
    on = (px >= n0) & (px < n0 + k)
    print("hits on needles", on.sum(), "on copies", (px >= n0 + k).sum())
    # error of the reported hit normal to the needle's plane (all needles lie in planes y = const)
    i = np.flatnonzero(on)
    y_plane = needles[px[i] - n0, 0, 1].astype(np.float64)
    y_hit = ori[i, 1].astype(np.float64) + tx[i].astype(np.float64) * d[i, 1].astype(np.float64)
    err = np.abs(y_hit - y_plane)
    pad = 2.0 ** -17 * (np.abs(ori[i]).max(axis=1) + np.abs(needles[px[i] - n0]).max(axis=(1, 2)))
    print("normal error: median %.3g  p99 %.3g  max %.3g;  beyond the standard padding: %d of %d" % (np.median(err), np.percentile(err, 99), err.max(), (err > pad).sum(), len(i)))
    allp = flat.tri_pos.reshape(-1, 3).astype(np.float64)
    mn, mx = allp.min(axis=0), allp.max(axis=0)
    radius = 0.5 * (mx - mn).max()
    dcam = np.abs(np.array([flat.camera.pos[q] for q in range(3)]) - 0.5 * (mn + mx)).max()
    safe_r = max(4 * radius, 2 * dcam + 2 * radius)
    nd = needles[px[i] - n0].astype(np.float64)
    a, b = nd[:, 1] - nd[:, 0], nd[:, 2] - nd[:, 0]
    sin0 = np.linalg.norm(np.cross(a, b), axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    bound = 32 * 2.0 ** -24 * 2.2 * safe_r / np.maximum(sin0, 5e-4)
    print("needle padding of the build: safe_r %.3g, smallest %.3g; largest error / padding %.3g" % (safe_r, bound.min(), (err / bound).max()))
    for sharing in (48, 0, 64):
        ctx.set_option("tune0", sharing)
        pb, tb = ctx.intersect_batch(ori, d)
        print("tune0", sharing, "mismatches", (pb != px).sum())
    ctx.set_option("tune0", 48)
    bad = 0
    worst = i[np.argsort(-(err - pad))][:200]
    for j in worst:
        p1, t1 = ctx.intersect_batch(ori[j:j + 1], d[j:j + 1])
        bad += int(p1[0]) != int(px[j])
    print("single-ray calls on the 200 worst: mismatches", bad)
