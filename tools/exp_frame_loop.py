import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
rtx = importlib.import_module("rust-raytracer_amd")
synth = importlib.import_module("rust-raytracer_amd.synth")
flat = synth.heightfield_scene(708)
ctx = rtx.Context(0); ctx.set_scene(flat); ctx.build()
w, h = 1920, 1080
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bufs = [torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0") for _ in range(nb)]
torch.cuda.synchronize()
mode = sys.argv[2] if len(sys.argv) > 2 else "a"
c = 0.0; s = 0.0; n = 0
t0 = time.perf_counter()
for k in range(2400):
    if mode == "a":
        if k >= 2:
            st = ctx.render_wait(); c += st.closest_ms; s += st.shadow_ms; n += 1
        ctx.render_device_async(w, h, 10, 10, bufs[k % nb].data_ptr())
    else:   # bench order: enqueue, then collect the previous one
        ctx.render_device_async(w, h, 10, 10, bufs[k % nb].data_ptr())
        if k >= 1:
            st = ctx.render_wait(); c += st.closest_ms; s += st.shadow_ms; n += 1
    if k % 400 == 399:
        print("frames to %d: closest %.4f shadow %.4f  wall %.4f ms/frame" % (k + 1, c / n, s / n, (time.perf_counter() - t0) / 400 * 1e3), flush=True)
        c = s = 0.0; n = 0; t0 = time.perf_counter()
