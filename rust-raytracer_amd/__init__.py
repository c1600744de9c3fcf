"""rust-raytracer_amd -- Python plumbing around librtx_hip.so (the C ABI in include/rtx.h).

The product is the HIP library; this module only loads it with ctypes, moves
numpy / torch buffers in and out, and mirrors the reference's entry points
(`RayTracer.init / set_scene / trace_rays`, src/lib.rs:55,64,257) for tests and
bench.py. There is no CPU fallback: if the library is missing, or no GPU is
visible, every call raises.

The directory name contains a hyphen, so import it with
`importlib.import_module("rust-raytracer_amd")` (see rtx_amd.py at the repo root).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB", os.path.join(_HERE, "librtx_hip.so"))   # RTX_LIB: another build of the same library
HOST_LIB_PATH = os.path.join(_HERE, "librstracer.so")
CLI_PATH = os.path.join(_HERE, "rstracer")

RTX_OK = 0
STATUS_NAMES = {0: "RTX_OK", 1: "RTX_ERR_INVALID_ARG", 2: "RTX_ERR_NO_DEVICE", 3: "RTX_ERR_HIP", 4: "RTX_ERR_NO_SCENE",
                5: "RTX_ERR_NOT_BUILT", 6: "RTX_ERR_UNSUPPORTED", 7: "RTX_ERR_OUT_OF_MEMORY"}
MAX_DEPTH = 32
MATERIAL_NONE = 0xFFFFFFFF
LIGHT_POINT, LIGHT_AREA, LIGHT_DIRECTIONAL = 0, 1, 2
TRI_VERTEX_NORMAL, TRI_VERTEX_MATERIAL = 1, 2
TILE = 8

# Every symbol include/rtx.h declares; tests check that the library exports all of them.
EXPORTED_SYMBOLS = [
    "rtx_abi_version", "rtx_source_hash", "rtx_create", "rtx_destroy", "rtx_last_error", "rtx_set_option", "rtx_scene_set", "rtx_build",
    "rtx_intersect_batch", "rtx_intersect_batch_device", "rtx_intersect_batch_exhaustive", "rtx_primary_hits",
    "rtx_render", "rtx_render_device", "rtx_render_tiles", "rtx_tiles_per_rank", "rtx_assemble_tiles", "rtx_stream",
    "rtx_debug_stamps", "rtx_render_device_async", "rtx_render_tiles_async", "rtx_render_wait", "rtx_assemble_tiles_async",
    "rtx_intersect_batch_device_noalloc", "rtx_scene_set_device", "rtx_clone",
    "rtx_comm_unique_id", "rtx_comm_init", "rtx_comm_destroy", "rtx_scene_broadcast", "rtx_gather_frame",
    "rtx_render_frames", "rtx_render_frames_async", "rtx_assemble_frames_async", "rtx_gather_frames",
]
MAX_BATCH_FRAMES = 64


def _header_defaults():
    """RTX_DEFAULT_TUNE* of include/rtx.h (the one place the defaults are written down)."""
    import re
    text = open(os.path.join(os.path.dirname(_HERE), "include", "rtx.h")).read()
    out = {}
    for name, expr in re.findall(r"#define\s+(RTX_DEFAULT_TUNE\d)\s+([^/\n]+)", text):
        out[name] = int(eval(expr, {"__builtins__": {}}))      # integer arithmetic only: "48", "(32 + (10 << 8))"
    return out


DEFAULT_TUNE = _header_defaults()


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("ambient", C.c_float * 3), ("specular", C.c_float * 3),
                ("emissive", C.c_float * 3), ("shininess", C.c_float), ("transparency", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("a", C.c_float * 3), ("b", C.c_float * 3), ("color", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("view_dir", C.c_float * 3), ("ortho_up", C.c_float * 3),
                ("focal_dist", C.c_float), ("vertical_fov", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("tri_pos", C.c_void_p), ("tri_normals", C.c_void_p),
                ("tri_mat", C.c_void_p), ("tri_flags", C.c_void_p), ("tri_prim_index", C.c_void_p),
                ("n_spheres", C.c_uint32), ("sph_center_radius", C.c_void_p), ("sph_mat", C.c_void_p),
                ("sph_prim_index", C.c_void_p), ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_lights", C.c_uint32), ("lights", C.c_void_p), ("camera", Camera)]


class BuildStats(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("max_depth", C.c_uint32),
                ("uses_bvh", C.c_uint32), ("key_bits", C.c_uint32), ("build_ms", C.c_float), ("upload_ms", C.c_float),
                ("n_always", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RenderStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("reflect_rays", C.c_uint64),
                ("refract_rays", C.c_uint64), ("skipped_rays", C.c_uint64), ("shaded_nodes", C.c_uint64),
                ("panic_pixels", C.c_uint64), ("render_ms", C.c_float), ("closest_ms", C.c_float),
                ("shadow_ms", C.c_float), ("closest_launches", C.c_uint32), ("shadow_launches", C.c_uint32),
                ("batches", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @property
    def traced_rays(self):
        """Rays actually traced on the device."""
        return self.primary_rays + self.shadow_rays + self.reflect_rays + self.refract_rays - self.skipped_rays

    @property
    def reference_rays(self):
        """Rays by the reference's definition (one per scene.intersects call)."""
        return self.primary_rays + self.shadow_rays + self.reflect_rays + self.refract_rays


MATERIAL_DTYPE = np.dtype([("diffuse", "<f4", 3), ("ambient", "<f4", 3), ("specular", "<f4", 3),
                           ("emissive", "<f4", 3), ("shininess", "<f4"), ("transparency", "<f4")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("a", "<f4", 3), ("b", "<f4", 3), ("color", "<f4", 3)])
assert MATERIAL_DTYPE.itemsize == C.sizeof(Material) == 56
assert LIGHT_DTYPE.itemsize == C.sizeof(Light) == 40


def build_native(force=False, jobs=8):
    """Compiles librtx_hip.so (hipcc, gfx950) and the host library + CLI (g++) in-tree."""
    if force:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "clean"])
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-s", "clean"])
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j%d" % jobs])
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "host"), "-s"])
    return LIB_PATH


_lib = None
_host = None


def lib():
    """Loads librtx_hip.so. Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("librtx_hip.so is missing (%s): run __graft_entry__.build() / make -C rust-raytracer_amd/csrc"
                               % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp, u32, u64, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64
        L.rtx_abi_version.restype = C.c_int
        L.rtx_source_hash.restype = C.c_char_p
        L.rtx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rtx_destroy.argtypes = [vp]
        L.rtx_destroy.restype = None
        L.rtx_last_error.argtypes = [vp]
        L.rtx_last_error.restype = C.c_char_p
        L.rtx_set_option.argtypes = [vp, C.c_char_p, i64]
        L.rtx_scene_set.argtypes = [vp, vp]
        L.rtx_scene_set_device.argtypes = [vp, vp]
        L.rtx_clone.argtypes = [vp, C.POINTER(vp)]
        L.rtx_build.argtypes = [vp, C.POINTER(BuildStats)]
        L.rtx_intersect_batch.argtypes = [vp, u64, vp, vp, vp, vp]
        L.rtx_intersect_batch_device.argtypes = [vp, u64, vp, vp, vp, vp]
        L.rtx_intersect_batch_exhaustive.argtypes = [vp, u64, vp, vp, vp, vp]
        L.rtx_intersect_batch_device_noalloc.argtypes = [vp, u64, vp, vp, vp, vp]
        L.rtx_primary_hits.argtypes = [vp, u32, u32, vp, vp, vp]
        L.rtx_render.argtypes = [vp, u32, u32, u32, u32, vp, C.POINTER(RenderStats)]
        L.rtx_render_device.argtypes = [vp, u32, u32, u32, u32, vp, C.POINTER(RenderStats)]
        L.rtx_render_tiles.argtypes = [vp, u32, u32, u32, u32, u32, u32, vp, C.POINTER(RenderStats)]
        L.rtx_tiles_per_rank.argtypes = [u32, u32, u32]
        L.rtx_tiles_per_rank.restype = u64
        L.rtx_assemble_tiles.argtypes = [vp, u32, u32, u32, vp, vp]
        L.rtx_assemble_tiles_async.argtypes = [vp, u32, u32, u32, vp, vp]
        L.rtx_stream.argtypes = [vp]
        L.rtx_stream.restype = vp
        L.rtx_render_device_async.argtypes = [vp, u32, u32, u32, u32, vp]
        L.rtx_render_tiles_async.argtypes = [vp, u32, u32, u32, u32, u32, u32, vp]
        L.rtx_render_wait.argtypes = [vp, C.POINTER(RenderStats)]
        L.rtx_render_frames.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32, vp, C.c_int, C.POINTER(RenderStats)]
        L.rtx_render_frames_async.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u32, vp, C.c_int]
        L.rtx_assemble_frames_async.argtypes = [vp, u32, u32, u32, u32, vp, vp]
        L.rtx_comm_unique_id.argtypes = [vp]
        L.rtx_comm_init.argtypes = [vp, u32, u32, vp]
        L.rtx_comm_destroy.argtypes = [vp]
        L.rtx_scene_broadcast.argtypes = [vp, vp, u32]
        L.rtx_gather_frame.argtypes = [vp, u32, u32, vp, vp, u32]
        L.rtx_gather_frames.argtypes = [vp, u32, u32, u32, vp, vp, u32]
        L.rtx_debug_stamps.argtypes = [vp, C.c_int, vp, u64]
        L.rtx_debug_stamps.restype = u64
        _lib = L
    return _lib


def source_hash():
    """Names the sources the LOADED library was built from (rtx_source_hash)."""
    return lib().rtx_source_hash().decode()


def host_lib():
    """Loads librstracer.so (C++ mirror of the reference's parser / RayTracer)."""
    global _host
    if _host is None:
        lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError("librstracer.so is missing: run make -C rust-raytracer_amd/host")
        H = C.CDLL(HOST_LIB_PATH)
        H.rst_last_error.restype = C.c_char_p
        H.rst_scene_parse.argtypes = [C.c_char_p]
        H.rst_scene_parse.restype = C.c_void_p
        H.rst_scene_parse_objects.argtypes = [C.c_char_p]
        H.rst_scene_parse_objects.restype = C.c_void_p
        H.rst_scene_desc.argtypes = [C.c_void_p]
        H.rst_scene_desc.restype = C.POINTER(SceneDesc)
        H.rst_scene_free.argtypes = [C.c_void_p]
        H.rst_scene_free.restype = None
        H.rst_render_file.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        _host = H
    return _host


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FlatScene:
    """The arrays of an rtx_scene_desc as numpy arrays (see include/rtx.h)."""

    def __init__(self):
        self.tri_pos = np.zeros((0, 3, 3), np.float32)
        self.tri_normals = None
        self.tri_mat = np.zeros((0, 3), np.uint32)
        self.tri_flags = None
        self.tri_prim_index = None
        self.sph_center_radius = np.zeros((0, 4), np.float32)
        self.sph_mat = np.zeros((0,), np.uint32)
        self.sph_prim_index = None
        self.materials = np.zeros((0,), MATERIAL_DTYPE)
        self.lights = np.zeros((0,), LIGHT_DTYPE)
        self.camera = Camera()

    n_tris = property(lambda self: int(self.tri_pos.shape[0]))
    n_spheres = property(lambda self: int(self.sph_center_radius.shape[0]))
    n_prims = property(lambda self: self.n_tris + self.n_spheres)

    def set_camera(self, pos, view_dir, ortho_up, vertical_fov, focal_dist=0.0):
        c = Camera()
        for i in range(3):
            c.pos[i] = float(np.float32(pos[i]))
            c.view_dir[i] = float(np.float32(view_dir[i]))
            c.ortho_up[i] = float(np.float32(ortho_up[i]))
        c.focal_dist = float(focal_dist)
        c.vertical_fov = float(np.float32(vertical_fov))
        self.camera = c

    def add_material(self, diffuse, ambient=(0.2, 0.2, 0.2), specular=(0, 0, 0), shininess=0.2, transparency=0.0):
        m = np.zeros((1,), MATERIAL_DTYPE)
        m["diffuse"], m["ambient"], m["specular"] = diffuse, ambient, specular
        m["shininess"], m["transparency"] = shininess, transparency
        self.materials = np.concatenate([self.materials, m])
        return len(self.materials) - 1

    def add_light(self, kind, a, color=(1, 1, 1), b=(0, 0, 0)):
        l = np.zeros((1,), LIGHT_DTYPE)
        l["type"], l["a"], l["b"], l["color"] = kind, a, b, color
        self.lights = np.concatenate([self.lights, l])

    def desc(self):
        """Returns (SceneDesc, keepalive)."""
        def prep(a, dt):
            return None if a is None else np.ascontiguousarray(a, dtype=dt)
        keep = [prep(self.tri_pos, np.float32), prep(self.tri_normals, np.float32), prep(self.tri_mat, np.uint32),
                prep(self.tri_flags, np.uint8), prep(self.tri_prim_index, np.uint32),
                prep(self.sph_center_radius, np.float32), prep(self.sph_mat, np.uint32),
                prep(self.sph_prim_index, np.uint32), np.ascontiguousarray(self.materials),
                np.ascontiguousarray(self.lights)]
        d = SceneDesc()
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

    @classmethod
    def from_desc(cls, d):
        """Copies the arrays behind a SceneDesc (e.g. one owned by librstracer.so)."""
        def arr(ptr, shape, dt):
            if not ptr or int(np.prod(shape)) == 0:
                return np.zeros(shape, dt)
            n = int(np.prod(shape)) * np.dtype(dt).itemsize
            return np.frombuffer(C.string_at(ptr, n), dtype=dt).reshape(shape).copy()
        f = cls()
        nt, ns = d.n_tris, d.n_spheres
        f.tri_pos = arr(d.tri_pos, (nt, 3, 3), np.float32)
        f.tri_normals = arr(d.tri_normals, (nt, 3, 3), np.float32) if d.tri_normals else None
        f.tri_mat = arr(d.tri_mat, (nt, 3), np.uint32)
        f.tri_flags = arr(d.tri_flags, (nt,), np.uint8) if d.tri_flags else None
        f.tri_prim_index = arr(d.tri_prim_index, (nt,), np.uint32) if d.tri_prim_index else None
        f.sph_center_radius = arr(d.sph_center_radius, (ns, 4), np.float32)
        f.sph_mat = arr(d.sph_mat, (ns,), np.uint32)
        f.sph_prim_index = arr(d.sph_prim_index, (ns,), np.uint32) if d.sph_prim_index else None
        f.materials = arr(d.materials, (d.n_materials,), MATERIAL_DTYPE)
        f.lights = arr(d.lights, (d.n_lights,), LIGHT_DTYPE)
        cam = Camera()
        C.memmove(C.byref(cam), C.byref(d.camera), C.sizeof(Camera))
        f.camera = cam
        return f


def parse_scene(path, through_objects=False):
    """SceneParser::new(path).parse_scene() (src/scene/parser/mod.rs:20,314) -> FlatScene, via librstracer.so:
    one pass from the file to the arrays (SceneParser::parse_flat), or, with `through_objects`, through the
    mirror of the reference's Scene / Poly / Vertex objects and flatten()."""
    H = host_lib()
    h = (H.rst_scene_parse_objects if through_objects else H.rst_scene_parse)(os.fsencode(path))
    if not h:
        raise RuntimeError("parse error: " + H.rst_last_error().decode())
    try:
        return FlatScene.from_desc(H.rst_scene_desc(h).contents)
    finally:
        H.rst_scene_free(h)


class Context:
    """One rtx_context: a scene resident on one GPU."""

    def __init__(self, device=0, _handle=None):
        self._L = lib()
        h = C.c_void_p()
        if _handle is not None:
            h = _handle
        else:
            rc = self._L.rtx_create(int(device), C.byref(h))
            if rc != RTX_OK:
                raise RtxError(rc, self._L.rtx_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self.build_stats = None

    def clone(self):
        """A context on the same GPU sharing this one's scene and tree (rtx_clone): own stream and workspace."""
        h = C.c_void_p()
        self._check(self._L.rtx_clone(self._h, C.byref(h)))
        c = Context(self.device, _handle=h)
        c.build_stats = dict(self.build_stats) if self.build_stats else None
        return c

    def close(self):
        if getattr(self, "_h", None):
            self._L.rtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != RTX_OK:
            raise RtxError(rc, self._L.rtx_last_error(self._h).decode())

    def set_option(self, key, value):
        self._check(self._L.rtx_set_option(self._h, key.encode(), int(value)))

    def set_scene(self, flat):
        d, keep = flat.desc()
        self._check(self._L.rtx_scene_set(self._h, C.cast(C.byref(d), C.c_void_p)))

    def set_scene_device(self, meta, tensors):
        """Scene whose triangle / sphere arrays are device tensors (rtx_scene_set_device), e.g. the receive buffers of
        a broadcast. `meta`: dict with n_tris, n_spheres, materials and lights (numpy, host) and camera; `tensors` maps
        array names (tri_pos, tri_normals, tri_mat, tri_flags, tri_prim_index, sph_center_radius, sph_mat,
        sph_prim_index) to objects with data_ptr(), or None."""
        mats = np.ascontiguousarray(meta["materials"])
        lights = np.ascontiguousarray(meta["lights"])
        d = SceneDesc()
        d.n_tris, d.n_spheres = int(meta["n_tris"]), int(meta["n_spheres"])
        for name in ("tri_pos", "tri_normals", "tri_mat", "tri_flags", "tri_prim_index", "sph_center_radius", "sph_mat", "sph_prim_index"):
            t = tensors.get(name)
            setattr(d, name, None if t is None else C.c_void_p(t.data_ptr()))
        d.n_materials, d.materials = len(mats), _ptr(mats)
        d.n_lights, d.lights = len(lights), _ptr(lights)
        d.camera = meta["camera"]
        self._check(self._L.rtx_scene_set_device(self._h, C.cast(C.byref(d), C.c_void_p)))

    # -- exchange steps of a tile-sharded frame over RCCL (one process per GPU) --
    def comm_init(self, rank, world, comm_id):
        """Collective: every rank passes the 128 bytes that comm_unique_id() returned on one of them."""
        buf = C.create_string_buffer(bytes(comm_id), 128)
        self._check(self._L.rtx_comm_init(self._h, int(rank), int(world), C.cast(buf, C.c_void_p)))

    def comm_destroy(self):
        self._check(self._L.rtx_comm_destroy(self._h))

    def scene_broadcast(self, flat, root=0):
        """Collective: `flat` on the root rank, None elsewhere; afterwards every rank holds the scene (then build())."""
        if flat is None:
            self._check(self._L.rtx_scene_broadcast(self._h, None, int(root)))
        else:
            d, keep = flat.desc()
            self._check(self._L.rtx_scene_broadcast(self._h, C.cast(C.byref(d), C.c_void_p), int(root)))

    def gather_frame(self, width, height, d_rgb_tiles, d_rgb, root=0):
        """Collective: this rank's slab (device pointer) to the root, which assembles the image into d_rgb (device pointer)."""
        self._check(self._L.rtx_gather_frame(self._h, width, height, d_rgb_tiles, d_rgb, int(root)))

    def gather_frames(self, n_frames, width, height, d_rgb_tiles, d_rgb, root=0):
        """Collective: this rank's slabs of a batch to the root, which assembles the batch's images into d_rgb."""
        self._check(self._L.rtx_gather_frames(self._h, n_frames, width, height, d_rgb_tiles, d_rgb, int(root)))

    def build(self):
        st = BuildStats()
        self._check(self._L.rtx_build(self._h, C.byref(st)))
        self.build_stats = st.as_dict()
        return self.build_stats

    def stream(self):
        return self._L.rtx_stream(self._h)

    def debug_stamps(self, kind):
        """Per-wavefront records (start, end on the 100 MHz clock, units, placement) of the last
        closest-hit (kind 0) or shadow (kind 1) launch; needs set_option("stamps", 1) before build()."""
        n = int(self._L.rtx_debug_stamps(self._h, kind, None, 0))
        out = np.zeros((n, 4), np.uint64)
        if n:
            self._L.rtx_debug_stamps(self._h, kind, _ptr(out), n)
        return out

    def intersect_batch(self, ori, dir, exhaustive=False):
        ori = np.ascontiguousarray(ori, np.float32).reshape(-1, 3)
        dir = np.ascontiguousarray(dir, np.float32).reshape(-1, 3)
        n = ori.shape[0]
        prim = np.empty((n,), np.int32)
        t = np.empty((n,), np.float32)
        fn = self._L.rtx_intersect_batch_exhaustive if exhaustive else self._L.rtx_intersect_batch
        self._check(fn(self._h, n, _ptr(ori), _ptr(dir), _ptr(prim), _ptr(t)))
        return prim, t

    def intersect_batch_device(self, d_ori, d_dir, d_prim, d_t, n):
        """Device pointers (ints), e.g. torch tensors' data_ptr()."""
        self._check(self._L.rtx_intersect_batch_device(self._h, int(n), d_ori, d_dir, d_prim, d_t))

    def intersect_batch_device_noalloc(self, d_ori, d_dir, d_prim, d_t, n):
        """Enqueue-only variant that allocates nothing (see rtx_intersect_batch_device_noalloc): the caller
        synchronises the context's stream (stream()) or the device before reading the outputs."""
        self._check(self._L.rtx_intersect_batch_device_noalloc(self._h, int(n), d_ori, d_dir, d_prim, d_t))

    def primary_hits(self, width, height):
        prim = np.empty((height, width), np.int32)
        t = np.empty((height, width), np.float32)
        rdir = np.empty((height, width, 3), np.float32)
        self._check(self._L.rtx_primary_hits(self._h, width, height, _ptr(prim), _ptr(t), _ptr(rdir)))
        return prim, t, rdir

    def render(self, width, height, depth=10, samples=10):
        rgb = np.zeros((height, width, 3), np.uint8)
        st = RenderStats()
        self._check(self._L.rtx_render(self._h, width, height, depth, samples, _ptr(rgb), C.byref(st)))
        return rgb, st

    def render_device(self, width, height, depth, samples, d_rgb):
        """d_rgb: device pointer (int) to height * width * 3 bytes, row-major."""
        st = RenderStats()
        self._check(self._L.rtx_render_device(self._h, width, height, depth, samples, d_rgb, C.byref(st)))
        return st

    def render_device_async(self, width, height, depth, samples, d_rgb):
        """Enqueues a frame (see rtx_render_device_async); collect it with render_wait()."""
        self._check(self._L.rtx_render_device_async(self._h, width, height, depth, samples, d_rgb))

    def render_tiles_async(self, width, height, depth, samples, rank, world, d_rgb_tiles):
        self._check(self._L.rtx_render_tiles_async(self._h, width, height, depth, samples, rank, world, d_rgb_tiles))

    def render_wait(self):
        """Collects the oldest frame in flight and returns its RenderStats."""
        st = RenderStats()
        self._check(self._L.rtx_render_wait(self._h, C.byref(st)))
        return st

    def render_tiles(self, width, height, depth, samples, rank, world, d_rgb_tiles):
        """d_rgb_tiles: device pointer (int) to tiles_per_rank(width, height, world) * 192 bytes."""
        st = RenderStats()
        self._check(self._L.rtx_render_tiles(self._h, width, height, depth, samples, rank, world, d_rgb_tiles, C.byref(st)))
        return st

    def assemble_tiles(self, width, height, world, d_slabs, d_rgb):
        self._check(self._L.rtx_assemble_tiles(self._h, width, height, world, d_slabs, d_rgb))

    def assemble_tiles_async(self, width, height, world, d_slabs, d_rgb):
        """Enqueues the assembly on the context's stream (see stream()) and returns."""
        self._check(self._L.rtx_assemble_tiles_async(self._h, width, height, world, d_slabs, d_rgb))

    @staticmethod
    def _camera_array(n_frames, cameras):
        """cameras: None (the scene's camera for every frame) or a sequence of n_frames Camera structs."""
        if cameras is None:
            return None, None
        if len(cameras) != n_frames:
            raise ValueError("one camera per frame")
        arr = (Camera * n_frames)(*cameras)
        return arr, C.cast(arr, C.c_void_p)

    def render_frames(self, n_frames, cameras, width, height, depth, samples, rank, world, d_out, row_major):
        """A batch of frames in one set of launches (rtx_render_frames); d_out: device pointer to the frames' images
        (row_major, world 1) or to the frames' tile slabs."""
        keep, ptr = self._camera_array(n_frames, cameras)
        st = RenderStats()
        self._check(self._L.rtx_render_frames(self._h, n_frames, ptr, width, height, depth, samples, rank, world, d_out,
                                              1 if row_major else 0, C.byref(st)))
        return st

    def render_frames_async(self, n_frames, cameras, width, height, depth, samples, rank, world, d_out, row_major):
        """Enqueues a batch of frames; collect it with render_wait()."""
        keep, ptr = self._camera_array(n_frames, cameras)
        self._check(self._L.rtx_render_frames_async(self._h, n_frames, ptr, width, height, depth, samples, rank, world, d_out,
                                                    1 if row_major else 0))

    def assemble_frames_async(self, n_frames, width, height, world, d_slabs, d_rgb):
        self._check(self._L.rtx_assemble_frames_async(self._h, n_frames, width, height, world, d_slabs, d_rgb))


def comm_unique_id():
    """128 bytes naming a new communicator (rtx_comm_unique_id); call on one rank and hand them to the others."""
    buf = C.create_string_buffer(128)
    rc = lib().rtx_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != RTX_OK:
        raise RtxError(rc, lib().rtx_last_error(None).decode())
    return buf.raw


def tiles_per_rank(width, height, world):
    return int(lib().rtx_tiles_per_rank(width, height, world))


class RayTracer:
    """Mirror of the reference's RayTracer (src/lib.rs:24-278) on top of a Context."""

    def __init__(self):
        self.width = self.height = 0
        self.depth = 0
        self.num_samples = 1
        self.scene = None
        self.last_stats = None

    @classmethod
    def init(cls, width, height, depth, num_samples):   # src/lib.rs:55-62
        rt = cls()
        rt.width, rt.height, rt.depth, rt.num_samples = width, height, depth, num_samples
        return rt

    def set_scene(self, flat, device=0):                 # src/lib.rs:64-67
        ctx = Context(device)
        ctx.set_scene(flat)
        ctx.build()
        self.scene = ctx

    def trace_rays(self):                                # src/lib.rs:257-277
        if self.scene is None:
            raise RuntimeError("RayTracer has not been assigned any Scene")   # the reference panics with this text
        rgb, st = self.scene.render(self.width, self.height, self.depth, self.num_samples)
        self.last_stats = st
        return rgb
