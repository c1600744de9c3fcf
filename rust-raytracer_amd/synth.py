"""Seeded synthetic scenes for the throughput configurations (tessellated height
fields of a stated triangle count, optionally with spheres). Pure numpy integer
hashing, so the same arrays come out on every machine."""
import importlib
import math

import numpy as np

_pkg = importlib.import_module(__name__.rsplit(".", 1)[0])

SEED = 0x5EED


def _hash01(ix, iz, seed):
    """Integer lattice hash -> [0, 1) as float64 (exact in every numpy version)."""
    h = (ix.astype(np.uint64) * np.uint64(0x8DA6B343) + iz.astype(np.uint64) * np.uint64(0xD8163841)
         + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h.astype(np.float64) / 4294967296.0


def value_noise(x, z, octaves=4, seed=SEED):
    """Smooth value noise in [0, 1): `octaves` octaves on an integer-hash lattice."""
    total = np.zeros_like(x, dtype=np.float64)
    amp, freq, norm = 1.0, 4.0, 0.0
    for o in range(octaves):
        fx, fz = x * freq, z * freq
        ix, iz = np.floor(fx).astype(np.int64), np.floor(fz).astype(np.int64)
        tx, tz = fx - ix, fz - iz
        sx, sz = tx * tx * (3.0 - 2.0 * tx), tz * tz * (3.0 - 2.0 * tz)
        ixo, izo = ix + 4096, iz + 4096   # keep lattice coordinates non-negative
        s = seed + 1013 * o
        v00, v10 = _hash01(ixo, izo, s), _hash01(ixo + 1, izo, s)
        v01, v11 = _hash01(ixo, izo + 1, s), _hash01(ixo + 1, izo + 1, s)
        total += amp * ((v00 * (1 - sx) + v10 * sx) * (1 - sz) + (v01 * (1 - sx) + v11 * sx) * sz)
        norm += amp
        amp *= 0.5
        freq *= 2.0
    return total / norm


def heightfield_triangles(n, height=0.15, seed=SEED):
    """n x n cells over [-1,1]^2 in xz, two triangles per cell, row-major:
    triangle index = 2 * (row * n + col) + {0, 1}. Returns float32 [2*n*n, 3, 3]."""
    g = np.linspace(-1.0, 1.0, n + 1)
    x, z = np.meshgrid(g, g)                      # [row (z), col (x)]
    y = height * value_noise(x, z, 4, seed)
    p = np.stack([x, y, z], axis=-1).astype(np.float32)
    v00, v01 = p[:-1, :-1], p[:-1, 1:]
    v10, v11 = p[1:, :-1], p[1:, 1:]
    tris = np.empty((n, n, 2, 3, 3), np.float32)
    tris[:, :, 0, 0], tris[:, :, 0, 1], tris[:, :, 0, 2] = v00, v10, v01
    tris[:, :, 1, 0], tris[:, :, 1, 1], tris[:, :, 1, 2] = v01, v10, v11
    return tris.reshape(2 * n * n, 3, 3)


def _pcg32_floats(count, seed):
    """Deterministic uniform floats in [0,1): counter-based (splitmix64) so it vectorises."""
    z = (np.arange(count, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def heightfield_scene(n, spec=0.0, glass_lattice=0, small_spheres=0, seed=SEED):
    """Configurations of the throughput runs:
      n=708                                   -> 1 002 528 triangles, opaque, 4 point lights
      n=2829, spec=0.3, glass_lattice=8        -> 16 006 482 reflective triangles + 64 glass spheres
      n=5600, spec=0.3, small_spheres=1280000  -> 62 720 000 triangles + 1 280 000 spheres
    Camera (0,1.2,2.4) looking at the origin, up (0,1,0), vertical fov 0.785398."""
    f = _pkg.FlatScene()
    f.tri_pos = heightfield_triangles(n, seed=seed)
    nt = f.tri_pos.shape[0]
    ground = f.add_material((0.7, 0.7, 0.7), (0.2, 0.2, 0.2), (spec, spec, spec), 0.2, 0.0)
    f.tri_mat = np.full((nt, 3), MATERIAL_NONE, np.uint32)
    f.tri_mat[:, 0] = ground
    f.tri_flags = np.zeros((nt,), np.uint8)
    spheres, smat = [], []
    if glass_lattice:
        glass = f.add_material((0.1, 0.1, 0.1), (0.1, 0.1, 0.1), (0.357, 0.357, 0.357), 0.4, 0.714)
        k = glass_lattice
        for r in range(k):
            for c in range(k):
                spheres.append((-0.875 + 1.75 * c / (k - 1), 0.4, -0.875 + 1.75 * r / (k - 1), 0.05))
                smat.append(glass)
    if small_spheres:
        matte = f.add_material((0.8, 0.3, 0.2), (0.2, 0.2, 0.2), (0, 0, 0), 0.2, 0.0)
        u = _pcg32_floats(4 * small_spheres, seed).reshape(small_spheres, 4)
        sx, sz = u[:, 0] * 2 - 1, u[:, 1] * 2 - 1
        sy = 0.15 * value_noise(sx, sz, 4, seed) + 0.01 + 0.2 * u[:, 2]
        sr = 5e-4 * (0.5 + u[:, 3])
        block = np.stack([sx, sy, sz, sr], axis=1).astype(np.float32)
        f.sph_center_radius = block
        f.sph_mat = np.full((small_spheres,), matte, np.uint32)
    if spheres:
        lattice = np.asarray(spheres, np.float32)
        f.sph_center_radius = np.concatenate([lattice, f.sph_center_radius]) if small_spheres else lattice
        f.sph_mat = np.concatenate([np.asarray(smat, np.uint32), f.sph_mat]) if small_spheres else np.asarray(smat, np.uint32)
    for lx in (-1.5, 1.5):
        for lz in (-1.5, 1.5):
            f.add_light(LIGHT_POINT, (lx, 2.0, lz), (1, 1, 1))
    pos = np.array([0.0, 1.2, 2.4])
    view = -pos / np.linalg.norm(pos)
    f.set_camera(pos, view, (0.0, 1.0, 0.0), 0.785398)
    return f


def camera_path(flat, first, count, step=0.004):
    """`count` cameras of a fly-around of the height field, frames first .. first + count - 1 of the path: frame k looks
    at the origin from the scene camera's position turned by k * step radians about the y axis, bobbing by a few percent
    of its height (period 483 frames, so that no two frames of an orbit coincide). Frame 0 is the scene's own camera.
    All arithmetic in float64, rounded once to the f32 fields of rtx_camera: every rank computes the same cameras."""
    p0 = np.array([flat.camera.pos[k] for k in range(3)], np.float64)
    cams = []
    for k in range(first, first + count):
        a = k * step
        ca, sa = math.cos(a), math.sin(a)
        pos = np.array([ca * p0[0] + sa * p0[2], p0[1] * (1.0 + 0.04 * math.sin(2.0 * math.pi * k / 483.0)), -sa * p0[0] + ca * p0[2]])
        view = -pos / np.linalg.norm(pos)
        c = _pkg.Camera()
        for i in range(3):
            c.pos[i] = float(np.float32(pos[i]))
            c.view_dir[i] = float(np.float32(view[i]))
            c.ortho_up[i] = flat.camera.ortho_up[i]
        c.focal_dist = flat.camera.focal_dist
        c.vertical_fov = flat.camera.vertical_fov
        if k == 0:      # exactly the scene's camera
            for i in range(3):
                c.pos[i] = flat.camera.pos[i]
                c.view_dir[i] = flat.camera.view_dir[i]
        cams.append(c)
    return cams


MATERIAL_NONE = _pkg.MATERIAL_NONE
LIGHT_POINT = _pkg.LIGHT_POINT


def bytes_per_ray(n_prims):
    """Algorithmic HBM bytes of one closest-hit query: ray in (32) + both child boxes
    on a root-to-leaf path (2 * ceil(log2 N) * 32) + one primitive (36) + hit out (16)."""
    return 32 + 2 * int(math.ceil(math.log2(max(n_prims, 2)))) * 32 + 36 + 16
