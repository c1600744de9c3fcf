#!/usr/bin/env python3
"""Closest-hit launch time of the bench scene with the scene in view and with a camera that looks away (a frame of
nothing but sky: uniformly cheap work units), one frame and a batch of eight.  usage: python tools/sky_probe.py [lib]"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
if len(sys.argv) > 1:
    os.environ["RTX_LIB"] = os.path.join(R, "rust-raytracer_amd", sys.argv[1])
import torch
rtx = importlib.import_module("rust-raytracer_amd"); synth = importlib.import_module("rust-raytracer_amd.synth")
w, h = 1920, 1080
for away in (False, True):
    flat = synth.heightfield_scene(708)
    if away:
        flat.set_camera((0.0, 1.2, 2.4), (0.0, 0.6, 0.8), (0.0, 1.0, 0.0), 0.785398)
    for frames in (1, 8):
        with rtx.Context(0) as ctx:
            ctx.set_scene(flat); ctx.build()
            img = torch.zeros(frames * h * w * 3, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
            ts, ss = [], []
            for _ in range(10):
                st = ctx.render_frames(frames, None, w, h, 10, 10, 0, 1, img.data_ptr(), True) if frames > 1 else ctx.render_device(w, h, 10, 10, img.data_ptr())
                ts.append(st.closest_ms); ss.append(st.shadow_ms)
            print("looking away" if away else "scene in view", "frames", frames, "closest_ms median %.4f" % np.median(ts[3:]), "shadow_ms %.4f" % np.median(ss[3:]), flush=True)
