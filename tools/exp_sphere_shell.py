#!/usr/bin/env python3
"""Experiment (GPU box): how far outside a small sphere does the reference's f32 sphere test (sphere.rs:54-103) still report
a hit? Rays aimed at distance sqrt(r^2 + c u D^2) from the centre, u = 2^-24, for a ladder of c; the exhaustive device
kernel (the reference's arithmetic) says which of them hit. The analysis behind the traversal's margin gives c <= 15."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rtx = importlib.import_module("rust-raytracer_amd")
f = rtx.FlatScene()
m = f.add_material((0.5, 0.5, 0.5))
f.add_light(rtx.LIGHT_POINT, (0, 5, 0))
f.set_camera((0, 0, 5), (0, 0, -1), (0, 1, 0), 0.8)
rng = np.random.default_rng(7)
ns = 20
f.sph_center_radius = np.concatenate([rng.uniform(-1, 1, (ns, 3)), 10.0 ** rng.uniform(-4, -3, (ns, 1))], axis=1).astype(np.float32)
f.sph_mat = np.full((ns,), m, np.uint32)
with rtx.Context(0) as ctx:
    ctx.set_option("force_mode", 2)
    ctx.set_scene(f)
    ctx.build()
    for c in (0.0625, 0.125, 0.25, 0.5, 1, 1.5, 2, 3, 4, 6, 8, 12, 15, 16):
        n = 4000000
        which = rng.integers(0, ns, n)
        sph = f.sph_center_radius[which].astype(np.float64)
        c0, r = sph[:, :3], sph[:, 3]
        dist = 10.0 ** rng.uniform(-0.5, 1.5, n)
        u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
        ori = c0 + u * dist[:, None]
        perp = np.cross(-u, rng.normal(size=(n, 3))); perp /= np.linalg.norm(perp, axis=1)[:, None]
        p = np.sqrt(r * r + c * 2.0 ** -24 * dist * dist)
        d = c0 + perp * p[:, None] - ori
        d /= np.linalg.norm(d, axis=1)[:, None]
        d[::2] *= 10.0 ** rng.uniform(-2, 2, (len(d[::2]), 1))
        o32, d32 = ori.astype(np.float32), d.astype(np.float32)
        px, tx = ctx.intersect_batch(o32, d32, exhaustive=True)
        pb, tb = ctx.intersect_batch(o32, d32)
        # the ray's true distance from the centre it was aimed at, in f64 from the f32 inputs
        oc = o32.astype(np.float64) - c0
        dd = d32.astype(np.float64)
        pr2 = (oc * oc).sum(1) - (oc * dd).sum(1) ** 2 / (dd * dd).sum(1)
        c_true = (pr2 - r * r) / (2.0 ** -24 * (oc * oc).sum(1))
        hit = px == which          # the sphere the ray was aimed past (another one may be hit for real)
        print("aimed c = %-7g rays %d  hits on the aimed sphere %8d (%.5f)  largest true c among them %.3f   tree != loop: %d" %
              (c, n, hit.sum(), hit.mean(), c_true[hit].max() if hit.any() else float("nan"), (pb != px).sum()), flush=True)
