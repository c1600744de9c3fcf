"""One process per GPU: scene replication and framebuffer assembly over
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The frame is sharded by 8x8-pixel tile, tile i -> rank i % world (interleaved, so
sky and geometry regions are spread over all ranks). The only exchange steps are
a one-time broadcast of the flattened scene from rank 0 and, per frame, an
all-gather of equal-sized tile slabs; rendering itself needs no collective."""
import importlib

import numpy as np

_pkg = importlib.import_module(__name__.rsplit(".", 1)[0])
TILE = _pkg.TILE

_ARRAYS = [("tri_pos", np.float32), ("tri_normals", np.float32), ("tri_mat", np.uint32), ("tri_flags", np.uint8),
           ("tri_prim_index", np.uint32), ("sph_center_radius", np.float32), ("sph_mat", np.uint32),
           ("sph_prim_index", np.uint32)]


def tiles_per_rank(width, height, world):
    """Same value as rtx_tiles_per_rank(): the largest per-rank tile count."""
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return (tiles + world - 1) // world


def pack_slab(image, rank, world):
    """numpy model of what rtx_render_tiles writes: the rank's tiles, in increasing
    tile order, 8*8*3 bytes each, edge tiles zero-padded."""
    h, w, _ = image.shape
    tx, ty = (w + TILE - 1) // TILE, (h + TILE - 1) // TILE
    padded = np.zeros((ty * TILE, tx * TILE, 3), np.uint8)
    padded[:h, :w] = image
    tiles = padded.reshape(ty, TILE, tx, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(ty * tx, TILE * TILE * 3)
    mine = tiles[rank::world]
    slab = np.zeros((tiles_per_rank(w, h, world), TILE * TILE * 3), np.uint8)
    slab[:mine.shape[0]] = mine
    return slab


def assemble(slabs, width, height, world):
    """numpy model of rtx_assemble_tiles: [world][tiles_per_rank][192] -> [h][w][3]."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    slabs = np.asarray(slabs, np.uint8).reshape(world, tiles_per_rank(width, height, world), TILE * TILE * 3)
    tiles = np.zeros((ty * tx, TILE * TILE * 3), np.uint8)
    for r in range(world):
        n = len(range(r, ty * tx, world))
        tiles[r::world] = slabs[r, :n]
    img = tiles.reshape(ty, tx, TILE, TILE, 3).transpose(0, 2, 1, 3, 4).reshape(ty * TILE, tx * TILE, 3)
    return img[:height, :width].copy()


def broadcast_scene(flat, src=0, device="cpu", timing=None):
    """Replicates a FlatScene from rank `src` to every rank. `flat` may be None on
    the other ranks. Arrays travel as byte tensors on `device`. `timing` (a dict) receives
    "collective_ms": the time inside the broadcast collectives alone (device-synchronised)."""
    import time
    import torch
    import torch.distributed as dist
    spent = [0.0]
    on_gpu = str(device).startswith("cuda")
    rank = dist.get_rank()
    header = [None]
    if rank == src:
        meta = {}
        for name, dt in _ARRAYS:
            a = getattr(flat, name)
            meta[name] = None if a is None else tuple(np.asarray(a).shape)
        meta["n_materials"] = len(flat.materials)
        meta["n_lights"] = len(flat.lights)
        meta["camera"] = bytes(flat.camera)
        header = [meta]
    dist.broadcast_object_list(header, src=src)
    meta = header[0]
    out = flat if rank == src else _pkg.FlatScene()

    def bcast(arr, nbytes):
        """Returns the array's bytes as a uint8 numpy array (a view of one host buffer, no further copies)."""
        if nbytes == 0:
            return np.zeros((0,), np.uint8)
        if rank == src:
            t = torch.from_numpy(arr.reshape(-1).view(np.uint8)).to(device)     # one host->device copy, no staging copies
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if on_gpu:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(t, src=src)
        if on_gpu:
            torch.cuda.synchronize()
        spent[0] += time.perf_counter() - t0
        return arr.reshape(-1).view(np.uint8) if rank == src else t.cpu().numpy()

    for name, dt in _ARRAYS:
        shape = meta[name]
        if shape is None:
            if rank != src:
                setattr(out, name, None)
            continue
        nbytes = int(np.prod(shape)) * np.dtype(dt).itemsize
        src_arr = np.ascontiguousarray(getattr(flat, name), dt) if rank == src else None
        raw = bcast(src_arr, nbytes)
        if rank != src:
            setattr(out, name, raw.view(dt).reshape(shape))
    raw = bcast(np.ascontiguousarray(flat.materials) if rank == src else None, meta["n_materials"] * _pkg.MATERIAL_DTYPE.itemsize)
    if rank != src:
        out.materials = raw.view(_pkg.MATERIAL_DTYPE)
    raw = bcast(np.ascontiguousarray(flat.lights) if rank == src else None, meta["n_lights"] * _pkg.LIGHT_DTYPE.itemsize)
    if rank != src:
        out.lights = raw.view(_pkg.LIGHT_DTYPE)
        out.camera = _pkg.Camera.from_buffer_copy(meta["camera"])
    if timing is not None:
        timing["collective_ms"] = spent[0] * 1e3
    return out


def broadcast_scene_device(flat, src, device, timing=None):
    """Replicates a scene from rank `src` into DEVICE memory of every rank: the triangle and sphere arrays are
    broadcast as device tensors and stay there (hand them to Context.set_scene_device), so a replicated scene crosses
    PCIe once, on the source rank only. Returns (meta, tensors): meta = counts, materials, lights (host numpy) and
    camera; tensors = {array name: uint8 device tensor or None}. `timing["collective_ms"]`: time inside the broadcasts."""
    import time
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    header = [None]
    if rank == src:
        meta = {"shapes": {}}
        for name, dt in _ARRAYS:
            a = getattr(flat, name)
            meta["shapes"][name] = None if a is None else tuple(np.asarray(a).shape)
        meta["n_tris"], meta["n_spheres"] = flat.n_tris, flat.n_spheres
        meta["materials"] = np.ascontiguousarray(flat.materials).tobytes()
        meta["lights"] = np.ascontiguousarray(flat.lights).tobytes()
        meta["camera"] = bytes(flat.camera)
        header = [meta]
    dist.broadcast_object_list(header, src=src)
    meta = header[0]
    spent = 0.0
    tensors = {}
    for name, dt in _ARRAYS:
        shape = meta["shapes"][name]
        nbytes = 0 if shape is None else int(np.prod(shape)) * np.dtype(dt).itemsize
        if nbytes == 0:
            tensors[name] = None
            continue
        if rank == src:
            host = np.ascontiguousarray(getattr(flat, name), dt).reshape(-1).view(np.uint8)
            t = torch.from_numpy(host).to(device)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dist.broadcast(t, src=src)
        torch.cuda.synchronize()
        spent += time.perf_counter() - t0
        tensors[name] = t
    out = {"n_tris": meta["n_tris"], "n_spheres": meta["n_spheres"],
           "materials": np.frombuffer(meta["materials"], dtype=_pkg.MATERIAL_DTYPE),
           "lights": np.frombuffer(meta["lights"], dtype=_pkg.LIGHT_DTYPE),
           "camera": _pkg.Camera.from_buffer_copy(meta["camera"])}
    if timing is not None:
        timing["collective_ms"] = spent * 1e3
    return out, tensors
