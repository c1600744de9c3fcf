#!/usr/bin/env python3
"""Frame-by-frame rate of the bench scene (one 1080p frame per set of launches, one context, two frames in flight) with and
without the events that time the traversal launches (the "kernel_timing" option; without them the frame is replayed as a graph).
The interpreter's cyclic collector is switched off: its first full collection (35-40 ms with torch imported, after about 900
frames' worth of allocations) otherwise lands in one of the runs, whatever its mode (tools/stall_probe.py).
usage: python tools/single_frame_probe.py [--frames 300] [--modes 11,01,00] [--width 1920 --height 1080]"""
import argparse, gc, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=300)
ap.add_argument("--modes", default="11,01,00,11,01", help="kernel_timing and graphs, two digits per run")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--lib", default="", help="another build of the library (file name in rust-raytracer_amd/)")
ap.add_argument("--opt", action="append", default=[], help="name=value, set before the runs (e.g. tune1=8224)")
a = ap.parse_args()
if a.lib:
    os.environ["RTX_LIB"] = os.path.join(ROOT, "rust-raytracer_amd", a.lib)
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
flat = synth.heightfield_scene(708)
w, h, depth = a.width, a.height, 10
path = synth.camera_path(flat, 0, a.frames + 12)
bufs = [torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
torch.cuda.synchronize()
gc.collect()
gc.disable()
with rtx.Context(0) as ctx:
    ctx.set_scene(flat)
    ctx.build()
    for kv in a.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    def run(k0, n):
        rays = 0
        for k in range(n):
            if k >= 2:
                rays += ctx.render_wait().traced_rays
            ctx.render_frames_async(1, [path[k0 + k]], w, h, depth, 10, 0, 1, bufs[k % 3].data_ptr(), True)
        for _ in range(min(n, 2)):
            rays += ctx.render_wait().traced_rays
        torch.cuda.synchronize()
        return rays
    for timing, graphs in [(int(m[0]), int(m[1])) for m in a.modes.split(",")]:
        ctx.set_option("kernel_timing", timing)
        ctx.set_option("graphs", graphs)
        run(0, 12)
        t0 = time.perf_counter()
        rays = run(12, a.frames)
        dt = time.perf_counter() - t0
        print("kernel_timing", timing, "graphs", graphs, "ms/frame %.4f" % (dt / a.frames * 1e3), "Mrays/s %.0f" % (rays / dt / 1e6), flush=True)
