#!/usr/bin/env python3
"""Renders N frames (plain launches, two in flight) and lists the intervals between consecutive rtx_render_wait returns that
exceed ten times the median, with the frame index at which they occur, for several frame sizes. With the interpreter's cyclic
collector on (--gc 1, Python's default) each process shows ONE interval of 35-40 ms around frame 850-1050 whatever the frame
size: the first full collection, due after ~70 000 container allocations; with it off (--gc 0) none. bench.py freezes and
disables the collector before its warm-up for that reason. usage: python tools/stall_probe.py [--frames 3000] [--gc 0|1]"""
import argparse, gc, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=3000)
ap.add_argument("--sizes", default="64x64,480x270,1920x1080")
ap.add_argument("--timing", type=int, default=0)
ap.add_argument("--gc", type=int, default=1)
a = ap.parse_args()
if not a.gc:
    gc.collect()
    gc.disable()
flat = synth.heightfield_scene(708)
path = synth.camera_path(flat, 0, 400)
for size in a.sizes.split(","):
    w, h = (int(x) for x in size.split("x"))
    bufs = [torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    with rtx.Context(0) as ctx:
        ctx.set_scene(flat)
        ctx.build()
        ctx.set_option("kernel_timing", a.timing)
        ctx.set_option("graphs", 0)
        stamps = []
        t0 = time.perf_counter()
        for k in range(a.frames):
            if k >= 2:
                ctx.render_wait()
                stamps.append(time.perf_counter())
            ctx.render_frames_async(1, [path[k % 400]], w, h, 10, 10, 0, 1, bufs[k % 3].data_ptr(), True)
        ctx.render_wait(); ctx.render_wait()
        torch.cuda.synchronize()
        d = np.diff(np.array(stamps))
        med = float(np.median(d))
        big = np.flatnonzero(d > 10 * med)
        print(size, "frames", a.frames, "median interval %.4f ms" % (med * 1e3), "total %.1f ms" % ((time.perf_counter() - t0) * 1e3),
              "stalls:", [(int(i) + 3, "%.1f ms" % (d[i] * 1e3), "at %.0f ms" % ((stamps[i] - t0) * 1e3)) for i in big[:12]], flush=True)
