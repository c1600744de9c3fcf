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
ap.add_argument("--show-noise", action="store_true", help="list the differing rays that are counted apart")
ap.add_argument("--family", type=int, default=0, help="0: soups, planes, duplicates, grids, spheres; 1: slivers, grazing rays, far offsets, tangent rays")
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


def make_scene1(rng):
    """Adversarial geometry: slivers, a large floor for grazing rays, scenes far from the origin, spheres for tangent rays."""
    f = rtx.FlatScene()
    f.add_material((0.7, 0.7, 0.7))
    f.add_light(rtx.LIGHT_POINT, (0.0, 5.0, 0.0))
    f.set_camera((0, 0, 5), (0, 0, -1), (0, 1, 0), 0.8)
    kind = 6 + int(rng.integers(0, 4))
    extent = float(10 ** rng.uniform(-1, 1.5))
    offset = np.zeros(3)
    tris = np.zeros((0, 3, 3))
    n_s = 0
    if kind == 6:      # slivers of every aspect ratio among ordinary triangles
        n_t = int(10 ** rng.uniform(1, 4))
        a = rng.uniform(-extent, extent, (n_t, 3))
        u = rng.normal(0, 1, (n_t, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        w = rng.normal(0, 1, (n_t, 3)); w -= (w * u).sum(axis=1, keepdims=True) * u; w /= np.linalg.norm(w, axis=1, keepdims=True)
        L = extent * 10 ** rng.uniform(-2, 0, (n_t, 1))
        hgt = L * 10 ** rng.uniform(-7, 0, (n_t, 1))
        frac = rng.uniform(0, 1, (n_t, 1))
        tris = np.stack([a, a + u * L, a + u * L * frac + w * hgt], axis=1)
    elif kind == 7:    # a floor of few large triangles plus clutter: rays nearly in its plane
        g = int(rng.integers(1, 6))
        xs = np.linspace(-extent, extent, g + 1)
        y0 = float(rng.uniform(-0.5, 0.5) * extent)
        cells = []
        for i in range(g):
            for j in range(g):
                a = (xs[i], y0, xs[j]); b = (xs[i + 1], y0, xs[j]); c = (xs[i], y0, xs[j + 1]); d = (xs[i + 1], y0, xs[j + 1])
                cells.append((a, b, c)); cells.append((b, d, c))
        n_c = int(10 ** rng.uniform(0, 3))
        clutter = rng.uniform(-extent, extent, (n_c, 1, 3)) + rng.normal(0, 0.05 * extent, (n_c, 3, 3))
        tris = np.concatenate([np.array(cells), clutter])
    elif kind == 8:    # an ordinary soup far from the origin
        n_t = int(10 ** rng.uniform(1, 4))
        offset = rng.choice([-1.0, 1.0], 3) * 10 ** rng.uniform(1, 4.5, 3)
        tris = rng.uniform(-extent, extent, (n_t, 1, 3)) + rng.normal(0, 1, (n_t, 3, 3)) * extent * 10 ** rng.uniform(-3, -0.5, (n_t, 1, 1))
        n_s = int(rng.integers(0, 50))
    else:              # spheres of every size, some triangles
        n_s = int(10 ** rng.uniform(0.5, 3.3))
        n_t = int(rng.integers(0, 200))
        tris = rng.uniform(-extent, extent, (n_t, 1, 3)) + rng.normal(0, 0.05 * extent, (n_t, 3, 3))
    f.tri_pos = (tris + offset).astype(np.float32)
    f.tri_mat = np.zeros((f.tri_pos.shape[0], 3), np.uint32)
    if n_s:
        c = rng.uniform(-extent, extent, (n_s, 3)) + offset
        r = extent * 10 ** rng.uniform(-4.5, -0.3, (n_s, 1))
        f.sph_center_radius = np.concatenate([c, r], axis=1).astype(np.float32)
        f.sph_mat = np.zeros((n_s,), np.uint32)
    return f, extent, kind, offset


def make_rays1(rng, f, extent, offset, n, kind):
    o = (rng.uniform(-2 * extent, 2 * extent, (n, 3)) + offset).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    k = n // 4
    if f.n_tris:
        t = f.tri_pos[rng.integers(0, f.n_tris, k)].astype(np.float64)
        w = rng.dirichlet((1, 1, 1), k)
        p = (t * w[:, :, None]).sum(axis=1)
        d[:k] = (p - o[:k]).astype(np.float32)                       # at points of triangles
        # nearly in the plane of a triangle: from a point of the plane (often outside the triangle) along an in-plane
        # direction tilted by 1e-8 .. 1e-2 rad, towards the triangle
        e1 = t[:, 1] - t[:, 0]; e2 = t[:, 2] - t[:, 0]
        nrm = np.cross(e1, e2); ln = np.linalg.norm(nrm, axis=1, keepdims=True); nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-300), 0.0)
        start = p + (e1 * rng.uniform(-3, 3, (k, 1)) + e2 * rng.uniform(-3, 3, (k, 1)))
        tilt = rng.choice([-1.0, 1.0], (k, 1)) * 10 ** rng.uniform(-8, -2, (k, 1))
        dirp = p - start
        dl = np.linalg.norm(dirp, axis=1, keepdims=True)
        start = start - nrm * tilt * dl * rng.uniform(0, 2, (k, 1))
        o[k:2 * k] = start.astype(np.float32)
        d[k:2 * k] = (dirp + nrm * tilt * dl).astype(np.float32)
        # along edges and through vertices
        o[2 * k:2 * k + k // 2] = (t[:k // 2, 0] - e1[:k // 2] * rng.uniform(0.1, 2, (k // 2, 1))).astype(np.float32)
        d[2 * k:2 * k + k // 2] = e1[:k // 2].astype(np.float32)
    if f.n_spheres:
        m = n - 3 * k
        c = f.sph_center_radius[rng.integers(0, f.n_spheres, m)].astype(np.float64)
        oo = o[3 * k:].astype(np.float64)
        inside = rng.random(m) < 0.2
        oo[inside] = c[inside, :3] + rng.normal(0, 0.3, (int(inside.sum()), 3)) * c[inside, 3:4]
        to_c = c[:, :3] - oo
        dist = np.linalg.norm(to_c, axis=1, keepdims=True)
        u = rng.normal(0, 1, (m, 3)); u -= (u * to_c).sum(axis=1, keepdims=True) * to_c / np.maximum(dist ** 2, 1e-300)
        u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-300)
        target = c[:, :3] + u * c[:, 3:4] * (1 + rng.choice([-1.0, 1.0], (m, 1)) * 10 ** rng.uniform(-8, -1, (m, 1)))   # tangent, just in or out
        o[3 * k:] = oo.astype(np.float32)
        d[3 * k:] = (target - oo).astype(np.float32)
    scale = np.where(rng.random((n, 1)) < 0.5, 1.0, 10 ** rng.uniform(-2, 2, (n, 1)))
    nz = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(rng.random((n, 1)) < 0.5, d / np.maximum(nz, 1e-30), d).astype(np.float32)   # half of them unit vectors, as camera rays are
    return o, (d * scale).astype(np.float32)


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
in_plane_total = 0
for seed in range(args.first_seed, args.first_seed + args.scenes):
    if time.time() > t_end:
        break
    rng = np.random.default_rng(seed)
    if args.family == 0:
        f, extent, kind = make_scene(rng)
    else:
        f, extent, kind, offset = make_scene1(rng)
    if f.n_prims == 0:
        continue
    o, d = make_rays(rng, f, extent, args.rays) if args.family == 0 else make_rays1(rng, f, extent, offset, args.rays, kind)
    with rtx.Context(0) as ctx:
        ctx.set_option("verify_build", 1)
        ctx.set_scene(f)
        ctx.build()
        p1, t1 = ctx.intersect_batch(o, d)
        p0, t0 = ctx.intersect_batch(o, d, exhaustive=True)
    same = (p0 == p1) & (t0.view(np.uint32) == t1.view(np.uint32))
    total_rays += len(p0)
    # A ray for which the determinant of poly.rs:149-153 is rounding noise makes the reference's test report hits at
    # arbitrary t, which no bounding box can promise to contain. The barycentric coordinates are quotients
    # [s, d, e2] / det and [d, s, e1] / det (s = origin - v0): the numerators carry a rounding error of about
    # u |s| |d| max(|e1|, |e2|), u = 2^-23 (and the determinant itself one of u |d| |e1| |e2|: |s| below counts as at least
    # an edge's length), so the coordinates are uncertain by
    #     noise = u |s| |d| max(|e1|, |e2|) / |det|  =  u |s| / (|e| sin(angle at v0) cos(angle between ray and normal)).
    # For an ordinary triangle seen from nearby that takes a ray IN its plane (direction and origin within ~1e-6 of it);
    # a needle (small angle at v0) or a far origin widens the set in proportion. A mismatch is counted apart -- the
    # documented difference -- when the noise reaches 3 % of the coordinates' size; everything else fails.
    in_plane = 0
    real = []
    for i in np.flatnonzero(~same):
        lies_in = False
        for pid in (int(p0[i]), int(p1[i])):
            if 0 <= pid < f.n_tris:
                tv = f.tri_pos[pid].astype(np.float64)
                e1, e2 = tv[1] - tv[0], tv[2] - tv[0]
                nr = np.cross(e1, e2)
                ln = np.linalg.norm(nr)
                dd = d[i].astype(np.float64)
                sv = o[i].astype(np.float64) - tv[0]
                le1, le2 = np.linalg.norm(e1), np.linalg.norm(e2)
                has_area = ln > 1e-3 * le1 * le2   # others are tested by every ray: exact
                det = e1 @ np.cross(dd, e2)
                if has_area and det != 0.0:
                    bu = (sv @ np.cross(dd, e2)) / det
                    bv = (dd @ np.cross(sv, e1)) / det
                    noise = 2.0 ** -23 * max(np.linalg.norm(sv), le1, le2) * np.linalg.norm(dd) * max(le1, le2) / abs(det)
                    if noise > 0.03 * max(1.0, abs(bu), abs(bv)):
                        lies_in = True
                        if args.show_noise:
                            say("  noise ray", int(i), "prim", pid, "noise %.3g" % noise, "barycentrics %.4g %.4g" % (bu, bv),
                                "sin at v0 %.3g" % (ln / (le1 * le2)), "|s| / max|e| %.3g" % (np.linalg.norm(sv) / max(le1, le2)),
                                "exhaustive", int(p0[i]), float(t0[i]), "bvh", int(p1[i]), float(t1[i]))
                elif has_area:
                    lies_in = True
        if lies_in:
            in_plane += 1
        else:
            real.append(int(i))
    in_plane_total += in_plane
    say("seed", seed, "kind", kind, "tris", f.n_tris, "spheres", f.n_spheres, "extent %.3g" % extent, "hits", int((p0 >= 0).sum()),
        "mismatches", len(real), "rays with a determinant of rounding noise that differ", in_plane)
    if real:
        bad += 1
        for i in real[:4]:
            say("  ray", i, "o", o[i].tolist(), "d", d[i].tolist(), "exhaustive", int(p0[i]), float(t0[i]), "bvh", int(p1[i]), float(t1[i]))
say("scenes with mismatches:", bad, "rays:", total_rays, "rays with a determinant of rounding noise that differ:", in_plane_total)
sys.exit(1 if bad else 0)
