"""ctypes binding of liboracle.so -- the CPU oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py. The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class RtxMaterial(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("ambient", C.c_float * 3), ("specular", C.c_float * 3),
                ("emissive", C.c_float * 3), ("shininess", C.c_float), ("transparency", C.c_float)]


class RtxLight(C.Structure):
    _fields_ = [("type", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("color", C.c_float * 3)]


class RtxCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("view_dir", C.c_float * 3), ("ortho_up", C.c_float * 3),
                ("focal_dist", C.c_float), ("vertical_fov", C.c_float)]


class RtxSceneDesc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("tri_pos", C.c_void_p), ("tri_normals", C.c_void_p),
                ("tri_mat", C.c_void_p), ("tri_flags", C.c_void_p), ("tri_prim_index", C.c_void_p),
                ("n_spheres", C.c_uint32), ("sph_center_radius", C.c_void_p), ("sph_mat", C.c_void_p),
                ("sph_prim_index", C.c_void_p), ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_lights", C.c_uint32), ("lights", C.c_void_p), ("camera", RtxCamera)]


class OrcStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("reflect_rays", C.c_uint64),
                ("refract_rays", C.c_uint64), ("shaded_nodes", C.c_uint64), ("last_level_rays", C.c_uint64),
                ("panic_pixels", C.c_uint64), ("prim_tests", C.c_uint64), ("seconds", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("ambient", "<f4", 3), ("specular", "<f4", 3),
                           ("emissive", "<f4", 3), ("shininess", "<f4"), ("transparency", "<f4")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("a", "<f4", 3), ("b", "<f4", 3), ("color", "<f4", 3)])
assert MATERIAL_DTYPE.itemsize == C.sizeof(RtxMaterial) == 56
assert LIGHT_DTYPE.itemsize == C.sizeof(RtxLight) == 40

MATERIAL_NONE = 0xFFFFFFFF
LIGHT_POINT, LIGHT_AREA, LIGHT_DIRECTIONAL = 0, 1, 2
TRI_VERTEX_NORMAL, TRI_VERTEX_MATERIAL = 1, 2


def build(force=False):
    """Compiles liboracle.so with the committed Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rt_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_parse.restype = C.c_void_p
        L.orc_scene_parse.argtypes = [C.c_char_p]
        L.orc_scene_from_desc.restype = C.c_void_p
        L.orc_scene_from_desc.argtypes = [C.c_void_p]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 5
        L.orc_scene_flatten.argtypes = [C.c_void_p] + [C.c_void_p] * 10 + [C.POINTER(RtxCamera)]
        L.orc_intersect_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(OrcStats)]
        L.orc_kat_vec_area.restype = C.c_float
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FlatScene:
    """Flattened scene arrays (numpy), convertible to an rtx_scene_desc."""

    def __init__(self):
        self.tri_pos = np.zeros((0, 3, 3), np.float32)
        self.tri_normals = None
        self.tri_mat = np.zeros((0, 3), np.uint32)
        self.tri_flags = np.zeros((0,), np.uint8)
        self.tri_prim_index = None
        self.sph_center_radius = np.zeros((0, 4), np.float32)
        self.sph_mat = np.zeros((0,), np.uint32)
        self.sph_prim_index = None
        self.materials = np.zeros((0,), MATERIAL_DTYPE)
        self.lights = np.zeros((0,), LIGHT_DTYPE)
        self.camera = RtxCamera()

    @property
    def n_tris(self):
        return int(self.tri_pos.shape[0])

    @property
    def n_spheres(self):
        return int(self.sph_center_radius.shape[0])

    @property
    def n_prims(self):
        return self.n_tris + self.n_spheres

    def set_camera(self, pos, view_dir, ortho_up, vertical_fov, focal_dist=0.0):
        c = RtxCamera()
        for i in range(3):
            c.pos[i] = float(np.float32(pos[i]))
            c.view_dir[i] = float(np.float32(view_dir[i]))
            c.ortho_up[i] = float(np.float32(ortho_up[i]))
        c.focal_dist = float(focal_dist)
        c.vertical_fov = float(np.float32(vertical_fov))
        self.camera = c

    def desc(self):
        """Returns (RtxSceneDesc, keepalive list)."""
        def prep(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        keep = [prep(self.tri_pos, np.float32), prep(self.tri_normals, np.float32), prep(self.tri_mat, np.uint32),
                prep(self.tri_flags, np.uint8), prep(self.tri_prim_index, np.uint32),
                prep(self.sph_center_radius, np.float32), prep(self.sph_mat, np.uint32),
                prep(self.sph_prim_index, np.uint32), np.ascontiguousarray(self.materials),
                np.ascontiguousarray(self.lights)]
        d = RtxSceneDesc()
        d.n_tris = self.n_tris
        d.tri_pos, d.tri_normals, d.tri_mat, d.tri_flags, d.tri_prim_index = [_ptr(a) for a in keep[0:5]]
        d.n_spheres = self.n_spheres
        d.sph_center_radius, d.sph_mat, d.sph_prim_index = [_ptr(a) for a in keep[5:8]]
        d.n_materials = len(self.materials)
        d.materials = _ptr(keep[8])
        d.n_lights = len(self.lights)
        d.lights = _ptr(keep[9])
        d.camera = self.camera
        return d, keep


class OracleScene:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: " + lib().orc_last_error().decode())
        self._h = C.c_void_p(handle)

    @classmethod
    def parse(cls, path):
        return cls(lib().orc_scene_parse(os.fsencode(path)))

    @classmethod
    def from_flat(cls, flat):
        d, keep = flat.desc()
        h = lib().orc_scene_from_desc(C.cast(C.byref(d), C.c_void_p))
        return cls(h)

    def close(self):
        if self._h:
            lib().orc_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        v = [C.c_uint32() for _ in range(5)]
        lib().orc_scene_counts(self._h, *[C.byref(x) for x in v])
        return dict(zip(["n_prims", "n_tris", "n_spheres", "n_materials", "n_lights"], [x.value for x in v]))

    def flatten(self):
        c = self.counts()
        f = FlatScene()
        f.tri_pos = np.zeros((c["n_tris"], 3, 3), np.float32)
        f.tri_normals = np.zeros((c["n_tris"], 3, 3), np.float32)
        f.tri_mat = np.zeros((c["n_tris"], 3), np.uint32)
        f.tri_flags = np.zeros((c["n_tris"],), np.uint8)
        f.tri_prim_index = np.zeros((c["n_tris"],), np.uint32)
        f.sph_center_radius = np.zeros((c["n_spheres"], 4), np.float32)
        f.sph_mat = np.zeros((c["n_spheres"],), np.uint32)
        f.sph_prim_index = np.zeros((c["n_spheres"],), np.uint32)
        f.materials = np.zeros((c["n_materials"],), MATERIAL_DTYPE)
        f.lights = np.zeros((c["n_lights"],), LIGHT_DTYPE)
        cam = RtxCamera()
        lib().orc_scene_flatten(self._h, _ptr(f.tri_pos), _ptr(f.tri_normals), _ptr(f.tri_mat), _ptr(f.tri_flags),
                                _ptr(f.tri_prim_index), _ptr(f.sph_center_radius), _ptr(f.sph_mat),
                                _ptr(f.sph_prim_index), _ptr(f.materials), _ptr(f.lights), C.byref(cam))
        f.camera = cam
        return f

    def intersect_batch(self, ori, dir, use_bvh=False):
        ori = np.ascontiguousarray(ori, np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        n = ori.shape[0]
        prim = np.empty((n,), np.int32)
        t = np.empty((n,), np.float32)
        lib().orc_intersect_batch(self._h, n, _ptr(ori), _ptr(dir), _ptr(prim), _ptr(t), int(use_bvh))
        return prim, t

    def render(self, width, height, depth=10, samples=10, stride=(1, 1), threads=1, use_bvh=False):
        rgb = np.zeros((height, width, 3), np.uint8)
        prim = np.full((height, width), -1, np.int32)
        t = np.zeros((height, width), np.float32)
        panic = np.zeros((height, width), np.uint8)
        rdir = np.zeros((height, width, 3), np.float32)
        st = OrcStats()
        lib().orc_render(self._h, width, height, depth, samples, stride[0], stride[1], threads, int(use_bvh),
                         _ptr(rgb), _ptr(prim), _ptr(t), _ptr(panic), _ptr(rdir), C.byref(st))
        return {"rgb": rgb, "prim_id": prim, "t": t, "panic": panic, "ray_dir": rdir, "stats": st.as_dict()}
