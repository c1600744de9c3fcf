#!/usr/bin/env python3
"""Fuzzing of the entry points around one frame: the same dressed random scene (tools/fuzz_render.py) rendered
  - through the host-buffer call, the device call (twice: the second frame runs on the first one's level hints), frames in
    flight, tile slabs of 2-5 ranks assembled on the device, batches of frames with and without cameras of their own, a scene
    handed over as device arrays and a clone of its context,
  - under random options: small ray batches (several passes per frame), a small node pool (frames rendered again in
    halves), level hints, recorded launches, kernel timing, cost order and the work-sharing threshold on and off,
must give the bytes of the plain synchronous frame of a fresh context every time.
usage: python tools/fuzz_api.py [--scenes N] [--seconds S] [--first-seed K] [--family 0|1]"""
import argparse, copy, ctypes, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
spec = importlib.util.spec_from_file_location("fuzz_render", os.path.join(ROOT, "tools", "fuzz_render.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)
rtx = gen.rtx
import torch


def cam_copy(c, shift):
    k = rtx.Camera()
    for i in range(3):
        k.pos[i] = c.pos[i] + shift[i]
        k.view_dir[i] = c.view_dir[i]
        k.ortho_up[i] = c.ortho_up[i]
    k.focal_dist = c.focal_dist
    k.vertical_fov = c.vertical_fov
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=50)
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--family", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=1, help="run every scene this many times (intermittent failures)")
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    bad = done = 0
    dev = torch.device("cuda:0")
    for seed in [x for x in range(args.first_seed, args.first_seed + args.scenes) for _ in range(args.repeat)]:
        if time.time() > t_end:
            break
        f, kind, depth, samples = gen.make(seed, args.family)
        if f is None or f.n_prims > 20000:
            continue
        rng = np.random.default_rng(seed + 7919)
        w, h = int(rng.integers(1, 120)), int(rng.integers(1, 90))
        with rtx.Context(0) as ctx:
            ctx.set_scene(f)
            ctx.build()
            ref, st_ref = ctx.render(w, h, depth, samples)
        other = cam_copy(f.camera, rng.normal(0, 0.05, 3) * np.linalg.norm([f.camera.pos[i] for i in range(3)]))
        f2 = copy.copy(f)
        with rtx.Context(0) as ctx:
            f2.camera = other
            ctx.set_scene(f2)
            ctx.build()
            ref_other, _ = ctx.render(w, h, depth, samples)
        opts = {"batch_pixels": int(rng.choice([1 << 24, 64 * int(rng.integers(1, 64)), 4096])),
                "pool_factor": int(rng.choice([4, 4, 2, 1])), "level_hints": int(rng.integers(0, 2)), "graphs": int(rng.integers(0, 2)),
                "kernel_timing": int(rng.integers(0, 2)), "cost_order": int(rng.integers(0, 2)), "tune0": int(rng.choice([48, 0, 64]))}
        fails = []
        try:
            with rtx.Context(0) as ctx:
                for k, v in opts.items():
                    ctx.set_option(k, v)
                ctx.set_scene(f)
                ctx.build()
                a, st = ctx.render(w, h, depth, samples)
                if not np.array_equal(a, ref) or st.reference_rays != st_ref.reference_rays:
                    fails.append("host call")
                # torch fills its buffers on ITS stream; the library renders on its own: every fill is waited for before the library sees the buffer
                bufs = [torch.full((h, w, 3), 7, dtype=torch.uint8, device=dev) for _ in range(3)]
                torch.cuda.synchronize()
                for i in range(2):
                    ctx.render_device(w, h, depth, samples, bufs[i].data_ptr())
                    if not np.array_equal(bufs[i].cpu().numpy(), ref):
                        fails.append("device call %d" % i)
                for b in bufs:
                    b.fill_(9)
                torch.cuda.synchronize()
                for i in range(3):      # at most two wait in the ring: the third call collects the first itself or is refused
                    try:
                        ctx.render_device_async(w, h, depth, samples, bufs[i].data_ptr())
                    except rtx.RtxError:
                        ctx.render_wait()
                        ctx.render_device_async(w, h, depth, samples, bufs[i].data_ptr())
                while True:
                    try:
                        ctx.render_wait()
                    except rtx.RtxError:
                        break
                torch.cuda.synchronize()
                for i in range(3):
                    if not np.array_equal(bufs[i].cpu().numpy(), ref):
                        fails.append("frame in flight %d" % i)
                world = int(rng.integers(2, 6))
                tpr = rtx.tiles_per_rank(w, h, world)
                slabs = torch.zeros((world, tpr * 192), dtype=torch.uint8, device=dev)
                out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize()
                for r in range(world):
                    if rng.random() < 0.5:
                        ctx.render_tiles(w, h, depth, samples, r, world, slabs[r].data_ptr())
                    else:
                        ctx.render_tiles_async(w, h, depth, samples, r, world, slabs[r].data_ptr())
                        ctx.render_wait()
                ctx.assemble_tiles(w, h, world, slabs.data_ptr(), out.data_ptr())
                torch.cuda.synchronize()
                if not np.array_equal(out.cpu().numpy(), ref):
                    dd = (out.cpu().numpy() != ref).any(axis=2)
                    ys, xs = np.nonzero(dd)
                    tiles = sorted({int((y // 8) * ((w + 7) // 8) + x // 8) for x, y in zip(xs, ys)})
                    fails.append("tiles of %d ranks: %d pixels in tiles %s (ranks %s)" % (world, int(dd.sum()), tiles[:12], sorted({t % world for t in tiles})))
                n_fr = int(rng.integers(2, 4))
                if n_fr * ((w + 7) // 8) * ((h + 7) // 8) * 64 <= opts["batch_pixels"]:      # a frame's rays come in whole 8x8 tiles
                    frames = torch.zeros((n_fr, h, w, 3), dtype=torch.uint8, device=dev)
                    torch.cuda.synchronize()
                    ctx.render_frames(n_fr, None, w, h, depth, samples, 0, 1, frames.data_ptr(), True)
                    torch.cuda.synchronize()
                    if not all(np.array_equal(frames[i].cpu().numpy(), ref) for i in range(n_fr)):
                        fails.append("batch of %d copies" % n_fr)
                    cams = [cam_copy(f.camera, (0, 0, 0)), other] + [cam_copy(f.camera, (0, 0, 0))] * (n_fr - 2)
                    frames.zero_()
                    torch.cuda.synchronize()
                    ctx.render_frames_async(n_fr, cams, w, h, depth, samples, 0, 1, frames.data_ptr(), True)
                    ctx.render_wait()
                    torch.cuda.synchronize()
                    got = frames.cpu().numpy()
                    if not (np.array_equal(got[0], ref) and np.array_equal(got[1], ref_other) and np.array_equal(got[-1], ref if n_fr > 2 else ref_other)):
                        fails.append("batch with cameras")
            # the same scene handed over as device arrays (rtx_scene_set_device), and a clone of that context
            names = {"tri_pos": np.float32, "tri_normals": np.float32, "tri_mat": np.uint32, "tri_flags": np.uint8, "tri_prim_index": np.uint32,
                     "sph_center_radius": np.float32, "sph_mat": np.uint32, "sph_prim_index": np.uint32}
            tensors = {}
            for name, dt in names.items():
                arr = getattr(f, name)
                tensors[name] = None if arr is None or np.asarray(arr).size == 0 else \
                    torch.from_numpy(np.ascontiguousarray(arr, dt).reshape(-1).view(np.uint8).copy()).to(dev)
            torch.cuda.synchronize()
            meta = {"n_tris": f.n_tris, "n_spheres": f.n_spheres, "materials": f.materials, "lights": f.lights, "camera": f.camera}
            with rtx.Context(0) as ctx:
                for k, v in opts.items():
                    ctx.set_option(k, v)
                ctx.set_scene_device(meta, tensors)
                ctx.build()
                a, st = ctx.render(w, h, depth, samples)
                if not np.array_equal(a, ref) or st.reference_rays != st_ref.reference_rays:
                    fails.append("device-resident scene")
                with ctx.clone() as twin:
                    b, st2 = twin.render(w, h, depth, samples)
                    if not np.array_equal(b, ref) or st2.reference_rays != st_ref.reference_rays:
                        fails.append("clone")
        except rtx.RtxError as e:
            # a node pool too small for the ray tree of ONE tile is the documented limit of the pool_factor option
            if not (e.status == 7 and (opts["pool_factor"] < 4 or opts["batch_pixels"] < 4096) and "even for one tile" in str(e)):
                fails.append("status %d: %s" % (e.status, str(e)[:160]))
        done += 1
        print("seed", seed, "kind", kind, "prims", f.n_prims, "%dx%d" % (w, h), "depth", depth, "samples", samples, opts, "FAILED: " + "; ".join(fails) if fails else "ok", flush=True)
        bad += bool(fails)
    print("scenes:", done, "with failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
