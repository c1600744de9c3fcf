#!/usr/bin/env python3
"""Differential fuzzing of the BVH query against the exhaustive loop on the GPU (both behind rtx_intersect_batch*):
random triangle soups and spheres of mixed scales, degenerate, duplicated and coplanar primitives (ties on t: the lowest
index must win, src/scene/mod.rs:186-199), rays from random points, from surface points and along the axes.
ids and the bits of t must agree.   usage: python tools/fuzz_bvh.py [--scenes N] [--seconds S] [--first-seed K]"""
import argparse, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=200)
ap.add_argument("--seconds", type=float, default=400.0)
ap.add_argument("--first-seed", type=int, default=0)
ap.add_argument("--rays", type=int, default=200000)
args = ap.parse_args()
rtx = importlib.import_module("rust-raytracer_amd")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = open(os.path.join(ROOT, "gpurun_out", "fuzz_bvh.log"), "a")


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    log.write(line + "\n")
    log.flush()


def make_scene(rng):
    f = rtx.FlatScene()
    f.add_material((0.7, 0.7, 0.7))
    f.add_light(rtx.LIGHT_POINT, (0.0, 5.0, 0.0))
    f.set_camera((0, 0, 5), (0, 0, -1), (0, 1, 0), 0.8)
    kind = int(rng.integers(0, 6))
    n_t = int(10 ** rng.uniform(0, 4.7)) if kind != 5 else 0
    n_s = int(10 ** rng.uniform(0, 3.3)) if kind in (1, 3, 5) else 0
    extent = float(10 ** rng.uniform(-2, 2))
    tris = np.zeros((n_t, 3, 3), np.float32)
    if n_t:
        n_cl = int(rng.integers(1, 8))
        centres = rng.uniform(-extent, extent, (n_cl, 3))
        which = rng.integers(0, n_cl, n_t)
        size = extent * 10 ** rng.uniform(-3.5, -0.3, (n_t, 1, 1))
        tris = (centres[which][:, None, :] + rng.normal(0, extent * 0.15, (n_t, 1, 3)) + rng.normal(0, 1, (n_t, 3, 3)) * size).astype(np.float32)
        if kind in (2, 3) and n_t > 8:       # coplanar, axis-aligned, duplicated and degenerate primitives
            k = n_t // 4
            tris[:k, :, 1] = np.float32(0.25 * extent)                    # all in one plane: overlapping, equal t for vertical rays
            tris[k:k + k // 2] = tris[:k // 2]                            # exact duplicates
            tris[k + k // 2:k + k // 2 + 4, 2] = tris[k + k // 2:k + k // 2 + 4, 1]   # zero-area triangles
        if kind == 4 and n_t > 2:            # a regular grid: shared edges and vertices
            g = int(np.sqrt(n_t // 2)) or 1
            xs = np.linspace(-extent, extent, g + 1, dtype=np.float32)
            cells = []
            for i in range(g):
                for j in range(g):
                    a = (xs[i], 0.0, xs[j]); b = (xs[i + 1], 0.0, xs[j]); c = (xs[i], 0.0, xs[j + 1]); d = (xs[i + 1], 0.0, xs[j + 1])
                    cells.append((a, b, c)); cells.append((b, d, c))
            tris = np.array(cells, np.float32)
    f.tri_pos = tris
    f.tri_mat = np.zeros((tris.shape[0], 3), np.uint32)
    if n_s:
        c = rng.uniform(-extent, extent, (n_s, 3))
        r = extent * 10 ** rng.uniform(-3.3, -0.5, (n_s, 1))
        f.sph_center_radius = np.concatenate([c, r], axis=1).astype(np.float32)
        f.sph_mat = np.zeros((n_s,), np.uint32)
    return f, extent, kind


def make_rays(rng, f, extent, n):
    o = rng.uniform(-2 * extent, 2 * extent, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    k = n // 5
    if f.n_tris:                               # towards and from points on triangles
        t = f.tri_pos[rng.integers(0, f.n_tris, k)]
        w = rng.dirichlet((1, 1, 1), k).astype(np.float32)
        p = (t * w[:, :, None]).sum(axis=1)
        d[:k] = p - o[:k]
        o[k:2 * k] = p + rng.normal(0, 1e-4 * extent, (k, 3)).astype(np.float32)
        o[2 * k:3 * k] = t[:, 0]               # from vertices
        d[2 * k:3 * k] = t[:, 1] - t[:, 0] + rng.normal(0, 1e-3, (k, 3)).astype(np.float32) * (rng.random((k, 1)) < 0.5)
    ax = rng.integers(0, 3, k)                 # axis-parallel rays: zero direction components
    d[3 * k:4 * k] = 0
    d[np.arange(3 * k, 4 * k), ax] = rng.choice([-1.0, 1.0], k)
    if f.n_spheres:
        c = f.sph_center_radius[rng.integers(0, f.n_spheres, k)]
        u = rng.normal(0, 1, (k, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        graze = c[:, :3] + u * c[:, 3:4] * rng.uniform(0.98, 1.02, (k, 1))    # silhouette points
        d[4 * k:5 * k] = (graze - o[4 * k:5 * k]).astype(np.float32)
    d *= (10 ** rng.uniform(-2, 2, (n, 1))).astype(np.float32)                # non-unit directions
    return o, d


t_end = time.time() + args.seconds
bad = 0
total_rays = 0
for seed in range(args.first_seed, args.first_seed + args.scenes):
    if time.time() > t_end:
        break
    rng = np.random.default_rng(seed)
    f, extent, kind = make_scene(rng)
    if f.n_prims == 0:
        continue
    o, d = make_rays(rng, f, extent, args.rays)
    with rtx.Context(0) as ctx:
        ctx.set_option("verify_build", 1)
        ctx.set_scene(f)
        ctx.build()
        p1, t1 = ctx.intersect_batch(o, d)
        p0, t0 = ctx.intersect_batch(o, d, exhaustive=True)
    same = (p0 == p1) & (t0.view(np.uint32) == t1.view(np.uint32))
    total_rays += len(p0)
    n_bad = int((~same).sum())
    say("seed", seed, "kind", kind, "tris", f.n_tris, "spheres", f.n_spheres, "extent %.3g" % extent, "hits", int((p0 >= 0).sum()), "mismatches", n_bad)
    if n_bad:
        bad += 1
        i = int(np.flatnonzero(~same)[0])
        say("  first:", i, "o", o[i].tolist(), "d", d[i].tolist(), "exhaustive", int(p0[i]), float(t0[i]), "bvh", int(p1[i]), float(t1[i]))
say("scenes with mismatches:", bad, "rays:", total_rays)
sys.exit(1 if bad else 0)
