#!/usr/bin/env python3
"""bench.py -- throughput of the frame hot path on N MI355X GPUs of one node.

Workload (BASELINE.json configs[2], the largest single-GPU configuration): synthetic
1 002 528-triangle tessellated height field + 4 point lights at 1920x1080, depth 10
(the reference's default; the scene has no reflective or refractive material, so one
level of rays is traced: primary + 4 shadow rays per hit). One step = --frames-per-step
frames (default 100 for this workload, so that the timed region of a 20-step run is about a
second, not ten milliseconds) of a camera path around the scene -- every frame has its own
camera (--cameras fixed: the scene's camera for every frame) -- rendered eight per GPU and set of
launches (--frames-per-batch), scene and BVH resident in HBM; the same path rendered one frame
per set of launches is reported beside it (`value_single_frame`). Batches are pipelined (the next
one is enqueued while the last one is collected) and all of them are complete before the clock stops. N > 1: the frame is sharded by interleaved 8x8 tiles,
the scene is broadcast once from rank 0 over RCCL and the tile slabs of a batch's frames are exchanged
so that every frame is assembled whole on one rank (an all-to-all: rank d gets the slabs of its run of the
batch's frames; --exchange all_gather delivers every frame to every rank). A step is --frames-per-step x N frames, so
the work per GPU and step does not depend on N (weak scaling: `value` = rays of all ranks / time).
--scaling strong: the step is the same --frames-per-step frames whatever N is, one frame per set of launches, each
frame cut into N shares (BASELINE configs[3] as written: one 4K depth-5 frame tile-sharded over 2/4/8 GPUs).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (fresh child
processes, one GPU each, started before this process touches a GPU); under torchrun the ranks
already exist and WORLD_SIZE must equal N.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import glob
import hashlib
import importlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (spec)
SIMDS = 256 * 4         # MI355X: 256 CUs x 4 SIMDs


def source_hash():
    """Hash of the library's sources: counters collected on another build must not be quoted for this one."""
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + [os.path.join(csrc, "Makefile")]):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def visible_gpus():
    """GPUs this process may use, counted WITHOUT loading the HIP runtime (the ranks are started from this process,
    which must never have touched a GPU): the kernel driver's topology lists every node, GPUs are the nodes with SIMDs;
    a *_VISIBLE_DEVICES list narrows that. None when the topology cannot be read (the ranks then find out themselves)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except Exception:
            return None
    return n


def launch_ranks(n, argv, backend):
    """Parent of a `--gpus N` run that was not started by torchrun: starts N fresh ranks (this process has
    not touched a GPU and never will), passes rank 0's JSON line through and fails if any rank fails."""
    have = visible_gpus()
    if backend != "gloo" and have is not None and have < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --backend gloo to rehearse ranks sharing a GPU)" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            r = p.poll()
            if r is None:
                continue
            procs.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
                deadline = time.time() + 30.0      # a failed rank leaves the others waiting in a collective
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    raise SystemExit(rc)


def host_cores():
    """Cores of the box and cores this process may run on (a GPU box gives one GPU's share of its CPUs)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = None
    return {"online": os.cpu_count(), "usable_by_this_process": usable}


def cpu_baseline(flat, width, height, depth, samples, stride, threads=1):
    """Times the CPU oracle (a C++ restatement of the reference renderer, which is single-threaded:
    threads=1 is the reference's configuration) on a stratified pixel subset of the same frame."""
    from oracle import oracle_ffi
    oracle_ffi.build()
    scene = oracle_ffi.OracleScene.from_flat(flat)
    t0 = time.perf_counter()
    out = scene.render(width, height, depth, samples, stride=(stride, stride), threads=threads)
    wall = time.perf_counter() - t0
    st = out["stats"]
    secs = st["seconds"] if threads == 1 else wall
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflect_rays"] + st["refract_rays"]
    npix = len(range(0, width, stride)) * len(range(0, height, stride))
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "host_cores": host_cores(),
            "sample": "%d pixels (every %dth in x and y) of the %dx%d frame, %d rays, %.1f s on %d thread(s); exhaustive loop over all primitives"
                      % (npix, stride, width, height, rays, secs, threads)}


def cpu_baseline_bvh(flat, width, height, depth, samples, stride):
    """The same oracle through its restatement of the reference's own `-b` path (median-split BVH,
    src/scene/bvh.rs:88-143), one thread: what a user of the reference gets on a large scene. The
    tree is built first (the reference builds it in Tree::init); frames differ slightly from the
    exhaustive ones by the reference's tie rule (bvh.rs:134), which does not matter for a rate."""
    from oracle import oracle_ffi
    oracle_ffi.build()
    scene = oracle_ffi.OracleScene.from_flat(flat)
    t0 = time.perf_counter()
    out = scene.render(width, height, depth, samples, stride=(stride, stride), threads=1, use_bvh=True)
    wall = time.perf_counter() - t0
    st = out["stats"]
    rays = st["primary_rays"] + st["shadow_rays"] + st["reflect_rays"] + st["refract_rays"]
    npix = len(range(0, width, stride)) * len(range(0, height, stride))
    return {"value": rays / st["seconds"] / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "host_cores": host_cores(),
            "tree_build_s": max(wall - st["seconds"], 0.0),
            "sample": "%d pixels (every %dth in x and y) of the %dx%d frame, %d rays, %.2f s after a %.1f s tree build; the "
                      "reference's median-split BVH" % (npix, stride, width, height, rays, st["seconds"], max(wall - st["seconds"], 0.0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames rendered per step (one step = one batch of frames of the same view); 0 = 100 for the 1M-triangle "
                         "workload, 10 for config 4, 4 for config 5")
    ap.add_argument("--grid", type=int, default=708, help="height-field cells per side (708 -> 1 002 528 triangles)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--spec", type=float, default=0.0)
    ap.add_argument("--glass", type=int, default=0)
    ap.add_argument("--cpu-stride", type=int, default=28, help="CPU baseline samples every n-th pixel in x and y (28: about 15 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the supplementary measurements outside the timed region (profiling runs)")
    ap.add_argument("--tune", type=str, default="", help="comma-separated tune0..3 overrides (experiments)")
    ap.add_argument("--backend", type=str, default="nccl",
                    help="torch.distributed backend for N > 1: nccl (= RCCL, one GPU per rank) or gloo (rehearsal: "
                         "collectives through host memory, ranks may share a GPU)")
    ap.add_argument("--collectives", type=str, default="torch", choices=("torch", "abi"),
                    help="N > 1: who runs the exchange steps. torch: torch.distributed broadcast / all-gather (RCCL underneath); "
                         "abi: the library's own entry points (rtx_comm_init, rtx_scene_broadcast, rtx_gather_frame: RCCL "
                         "broadcast and gather-to-root), as a Rust or C++ host would call them")
    ap.add_argument("--exchange", type=str, default="all_to_all", choices=("all_to_all", "all_gather"),
                    help="N > 1, torch collectives, batches of frames: all_to_all = frame f of a batch is assembled on rank "
                         "f // (frames per batch / N) -- every rank receives the slabs of ITS frames only, 1/N of what an "
                         "all-gather delivers to every rank -- all_gather = every rank receives and assembles every frame")
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5),
                    help="BASELINE configuration: 3 = 1M triangles at 1080p (default, the single-GPU bench workload), 4 = 16M triangles + "
                         "64 glass spheres at 4K depth 5, 5 = 62.72M triangles + 1.28M spheres at 8K depth 5")
    ap.add_argument("--contexts", type=int, default=0,
                    help="contexts (streams) the frames alternate between; 0 = 1 on a single GPU (per-kernel times stay those of a "
                         "kernel running alone), 3 on several GPUs (a rank's share of the frame cannot fill the GPU)")
    ap.add_argument("--no-cost-order", action="store_true", help="hand out traversal work units in index order (experiments)")
    ap.add_argument("--batch-pixels", type=int, default=0, help="pixels per wavefront batch (0 = the library's default)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="with one process: run the multi-GPU code path (process group of one rank, tile slab, all-gather, "
                         "assembly) instead of the single-GPU one; for checking that path on a one-GPU box")
    ap.add_argument("--frames-per-batch", type=int, default=0,
                    help="frames rendered as one set of launches (rtx_render_frames_async); 0 = 8 per GPU for the 1080p "
                         "configuration (so that a rank's share of a batch is as much work as eight whole frames), 2 per GPU for "
                         "config 4, 1 for config 5 and with --scaling strong")
    ap.add_argument("--lib", type=str, default="", help="another build of the library (file name in rust-raytracer_amd/; experiments)")
    ap.add_argument("--cameras", type=str, default="path", choices=("path", "fixed", "same"),
                    help="path (default): every frame has its own camera, frame k of a fly-around of the scene (synth.camera_path); "
                         "fixed: the scene's camera for every frame (the frames of a batch are then copies of one frame); "
                         "same: like fixed, but handed over as one explicit camera per frame (experiments: the cost of the camera array)")
    ap.add_argument("--scaling", type=str, default="weak", choices=("weak", "strong"),
                    help="weak (default): a step is --frames-per-step x N frames, rendered in batches, so the work per GPU does not depend on N; "
                         "strong: a step is --frames-per-step frames whatever N is, one frame per set of launches, each frame cut into N shares")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:], args.backend)      # does not return
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%s: they must agree" % (args.gpus, os.environ.get("WORLD_SIZE")))
    small_spheres = 0
    if args.config == 4:
        args.grid, args.width, args.height, args.depth, args.spec, args.glass = 2829, 3840, 2160, 5, 0.3, 8
    elif args.config == 5:
        args.grid, args.width, args.height, args.depth, args.spec, small_spheres = 5600, 7680, 4320, 5, 0.3, 1280000
    if args.lib:
        os.environ["RTX_LIB"] = os.path.join(ROOT, "rust-raytracer_amd", args.lib)
    fps = args.frames_per_step if args.frames_per_step > 0 else {3: 100, 4: 10, 5: 4}[args.config]

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # multi-process GPU work here needs dmabuf IPC (RCCL over xGMI)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    strong = args.scaling == "strong"
    if args.frames_per_step <= 0 and not strong:
        fps *= world      # a step is the same amount of work per GPU whatever N is: N x as many frames, each cut into N shares
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    multi = world > 1 or args.rehearse_dist        # the sharded code path (also with a single rank when rehearsing)
    gloo = multi and args.backend == "gloo"
    dev_index = local_rank % torch.cuda.device_count() if gloo else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if gloo else dev       # where collective buffers live
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))

    rtx = importlib.import_module("rust-raytracer_amd")
    synth = importlib.import_module("rust-raytracer_amd.synth")
    rdist = importlib.import_module("rust-raytracer_amd.dist")

    flat = synth.heightfield_scene(args.grid, spec=args.spec, glass_lattice=args.glass, small_spheres=small_spheres) if rank == 0 else None
    t_b0 = time.time()
    bcast_collective_ms = None
    scene_meta = scene_tensors = None
    abi = multi and not gloo and args.collectives == "abi"
    if abi:
        pass          # the library replicates the scene itself, below, once the context exists
    elif multi:
        timing = {}
        if gloo:      # rehearsal through host memory
            flat = rdist.broadcast_scene(flat, 0, cdev, timing=timing)
        else:         # the arrays land in HBM once and stay there (rtx_scene_set_device)
            scene_meta, scene_tensors = rdist.broadcast_scene_device(flat, 0, dev, timing=timing)
        bcast_collective_ms = timing.get("collective_ms")
    bcast_ms = (time.time() - t_b0) * 1e3

    # One context = one HIP stream with its own workspace and copy of the scene. With two, the frames
    # alternate between them, so the tail of one frame's kernels overlaps the start of the next frame's.
    # The contexts of a rank share ONE copy of the scene and ONE tree (rtx_clone).
    n_contexts = args.contexts if args.contexts > 0 else (3 if multi else 1)
    ctx = rtx.Context(dev_index)
    if args.no_cost_order:
        ctx.set_option("cost_order", 0)
    if args.batch_pixels > 0:
        ctx.set_option("batch_pixels", args.batch_pixels)
    for i, v in enumerate([x for x in args.tune.split(",") if x]):
        ctx.set_option("tune%d" % i, int(v))
    t_s0 = time.time()
    if abi:
        ids = [rtx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)          # the 128-byte id travels over the host channel we already have
        ctx.comm_init(rank, world, ids[0])
        torch.cuda.synchronize()
        t_c0 = time.perf_counter()
        ctx.scene_broadcast(flat if rank == 0 else None, 0)
        bcast_collective_ms = (time.perf_counter() - t_c0) * 1e3     # upload on the root + broadcasts + device copies
        bcast_ms = bcast_collective_ms
    elif scene_tensors is not None:
        ctx.set_scene_device(scene_meta, scene_tensors)
        scene_tensors = None      # copied device-to-device by the library
        torch.cuda.empty_cache()
    else:
        ctx.set_scene(flat)
    build = ctx.build()
    scene_setup_ms = (time.time() - t_s0) * 1e3     # scene into the context + tree build, per rank
    ctxs = [ctx] + [ctx.clone() for _ in range(n_contexts - 1)]
    w, h = args.width, args.height

    class _Cam:      # the scene's camera, known on every rank (the generator is deterministic; only rank 0 built the arrays)
        camera = (flat.camera if flat is not None else scene_meta["camera"] if scene_meta is not None
                  else synth.heightfield_scene(4).camera)
    flat_cam = _Cam
    tpr = rtx.tiles_per_rank(w, h, world)
    # Frames per batch: a rank's 1/N share of one small frame cannot fill a GPU (its launches are as long as their
    # longest ray chains however few rays they hold), so sharded runs render the shares of several frames at once.
    # Larger launches also lose less at their ends (a launch finishes with its slowest wavefronts while the rest of the
    # GPU idles), so the 1080p configuration renders 8 frames per batch and GPU (64 frames per batch, the library's limit,
    # on 8 ranks) -- measured 8.9 / 10.5 / 13.4 / 14.3 Grays/s with 1 / 2 / 4 / 8 frames per
    # batch on one GPU. The 4K configuration renders two frames per batch and GPU, the 8K configuration frame by frame.
    if args.frames_per_batch > 0:
        fpb = args.frames_per_batch
    elif strong:                # the frame is the unit: one per set of launches, cut N ways
        fpb = 1
    elif args.config == 3:
        fpb = min(8 * world, rtx.MAX_BATCH_FRAMES)
    elif args.config == 4:      # two 4K frames fill the default ray batch; their deeper levels' launches are twice as large too (+16 %)
        fpb = min(2 * world, rtx.MAX_BATCH_FRAMES)
    else:                       # one 8K frame is two ray batches by itself
        fpb = 1
    fps = (fps + fpb - 1) // fpb * fpb      # a step is a whole number of batches
    # Exchange step of a sharded batch. A frame has to be whole on ONE GPU, not on all of them: with all_to_all the
    # batch's frames are dealt out in equal runs, rank d assembles frames [d * fpr, (d + 1) * fpr) and receives only
    # those frames' slabs (N = 8: 50 MB in and out per rank and batch; an all-gather delivers 400 MB to every rank,
    # and every rank then assembles all 64 frames).
    a2a = multi and not abi and args.exchange == "all_to_all" and fpb > 1 and fpb % world == 0
    fpr = fpb // world if a2a else fpb      # frames this rank assembles per batch
    # Frames are pipelined: while the GPU renders frame k the host collects frame k - 1 (and, for
    # N > 1, its slabs travel while frame k renders and frame k - 2 is assembled). Every frame that
    # was enqueued inside the timed region is complete before the clock stops.
    in_flight = max(1, len(ctxs))     # frames enqueued beyond the one being collected
    # (sharded runs: two slabs per context, so that a context's recorded frames see the same two buffers again)
    # (and at least in_flight + 2: slab k is read by exchange k, which is only waited for when batch k + 1 is collected,
    # i.e. after render k + in_flight + 1 was enqueued -- that render must not be the one that rewrites slab k)
    ring = max(2 * len(ctxs), in_flight + 2) if multi else in_flight + 2
    images = [torch.zeros(fpb * h * w * 3, dtype=torch.uint8, device=dev) for _ in range(in_flight + 1)]
    slabs = [torch.zeros(fpb * tpr * 192, dtype=torch.uint8, device=dev) for _ in range(ring)] if multi else None
    gathered = [torch.zeros(world * fpr * tpr * 192, dtype=torch.uint8, device=dev) for _ in range(ring)] if multi else None
    gathered_host = torch.zeros(world * fpr * tpr * 192, dtype=torch.uint8).pin_memory() if gloo else None
    torch.cuda.synchronize()      # torch clears these on its own stream; the library's streams do not wait for it
    image = images[0]
    acc = {}
    gather_wait = [0.0]   # seconds the host waited for slabs that had not arrived when they were needed
    state = {"enqueued": 0, "collected": 0, "gather": None, "timed_base": 0}

    def reset_acc():
        acc.update({"closest_ms": 0.0, "shadow_ms": 0.0, "closest_launches": 0, "shadow_launches": 0, "render_ms": 0.0,
                    "closest_rays": 0, "shadow_rays": 0, "traced": 0, "reference": 0, "primary": 0, "shaded_primary": 0})
        gather_wait[0] = 0.0

    def account(st):
        if st.closest_ms > 0.0:      # a frame whose traversal launches carried timing events
            acc["closest_ms"] += st.closest_ms
            acc["shadow_ms"] += st.shadow_ms
            acc["closest_launches"] += st.closest_launches
            acc["shadow_launches"] += st.shadow_launches
            acc["closest_rays"] += st.primary_rays + st.reflect_rays + st.refract_rays - st.skipped_rays
            acc["shadow_rays"] += st.shadow_rays
        acc["render_ms"] += st.render_ms
        acc["traced"] += st.traced_rays
        if st.reflect_rays + st.refract_rays == 0:      # one level of rays: every shaded node is a primary hit
            acc["primary"] += st.primary_rays
            acc["shaded_primary"] += st.shaded_nodes
        acc["reference"] += st.reference_rays

    # GPU-side ordering between the library's streams, torch's current stream and RCCL's stream (no host waits):
    # slab k is read by all-gather k and rewritten by render k + ring; gathered k is written by all-gather k,
    # read by assembly k and rewritten by all-gather k + ring.
    lib_streams = [torch.cuda.ExternalStream(c.stream(), device=dev) for c in ctxs] if (multi and not gloo and not abi) else None
    # The assembly of a received batch (a copy kernel over the slabs of this rank's frames; with --exchange all_gather
    # over world x the rank's slabs: 0.8 GB of traffic per batch on 8 ranks) runs on a context of its own, i.e. on its
    # own stream, beside the rendering contexts' traversal kernels.
    asm_ctx = ctx.clone() if lib_streams is not None else None
    asm_stream = torch.cuda.ExternalStream(asm_ctx.stream(), device=dev) if asm_ctx is not None else None
    TIMED_EVERY = 16
    timing_state = [1] * len(ctxs)
    gather_done = [None] * ring      # recorded on torch's stream once all-gather k has been waited for there
    assemble_done = [None] * ring    # recorded on the first context's stream after assembly k
    event_pool = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(ring)] if (multi and not gloo) else None

    def finish_gather():
        if state["gather"] is None:
            return
        work, idx = state["gather"]
        state["gather"] = None
        t_w = time.perf_counter()
        work.wait()                                     # torch's current stream waits for the collective
        ev = event_pool[idx][0]
        ev.record()
        gather_done[idx] = ev
        asm_stream.wait_event(ev)                       # ... and so does the stream that assembles
        if fpb > 1:
            asm_ctx.assemble_frames_async(fpr, w, h, world, gathered[idx].data_ptr(), image.data_ptr())
        else:
            asm_ctx.assemble_tiles_async(w, h, world, gathered[idx].data_ptr(), image.data_ptr())
        ev2 = event_pool[idx][1]
        ev2.record(asm_stream)
        assemble_done[idx] = ev2
        gather_wait[0] += time.perf_counter() - t_w

    def collect():
        """Collects the oldest frame in flight; for N > 1 sends its slab on its way."""
        account(ctxs[state["collected"] % len(ctxs)].render_wait())
        idx = state["collected"] % ring
        state["collected"] += 1
        if not multi:
            return
        if abi:      # gather-to-root and assembly inside the library; rank 0 ends up with the image(s)
            ctx.gather_frames(fpb, w, h, slabs[idx].data_ptr(), image.data_ptr(), 0)
            return
        if gloo:
            if a2a:
                dist.all_to_all_single(gathered_host, slabs[idx].cpu())
            else:
                dist.all_gather_into_tensor(gathered_host, slabs[idx].cpu())
            gathered[idx].copy_(gathered_host)
            torch.cuda.current_stream().synchronize()
            ctx.assemble_frames_async(fpr, w, h, world, gathered[idx].data_ptr(), image.data_ptr())
            torch.cuda.ExternalStream(ctx.stream(), device=dev).synchronize()
            return
        finish_gather()            # the frame before it
        if assemble_done[idx] is not None:              # the assembly that last read this gather buffer
            torch.cuda.current_stream().wait_event(assemble_done[idx])
        if a2a:
            state["gather"] = (dist.all_to_all_single(gathered[idx], slabs[idx], async_op=True), idx)
        else:
            state["gather"] = (dist.all_gather_into_tensor(gathered[idx], slabs[idx], async_op=True), idx)

    # Cameras: batch k renders frames [k * fpb, (k + 1) * fpb) of a fly-around of the scene, one camera per frame
    # (every rank computes the same path); --cameras fixed renders the scene's camera every time.
    use_path = args.cameras == "path"
    path_cache = {}

    def cams_of(k):
        if args.cameras == "same":
            return path_cache.setdefault(0, synth.camera_path(flat_cam, 0, 1) * fpb)
        if not use_path:
            return None
        if k not in path_cache:
            path_cache[k] = synth.camera_path(flat_cam, k * fpb, fpb)
        return path_cache[k]

    def step():     # one batch of `fpb` frames; a bench step is `fps` frames
        k = state["enqueued"]
        state["enqueued"] += 1
        cams = cams_of(k)
        if not multi:   # single GPU: the frames are written row-major in place
            if fpb > 1 or cams is not None:
                ctxs[k % len(ctxs)].render_frames_async(fpb, cams, w, h, args.depth, 10, 0, 1, images[k % len(images)].data_ptr(), True)
            else:
                ctxs[k % len(ctxs)].render_device_async(w, h, args.depth, 10, images[k % len(images)].data_ptr())
        else:
            if lib_streams is not None and gather_done[k % ring] is not None:   # the all-gather that last read this slab
                lib_streams[k % len(ctxs)].wait_event(gather_done[k % ring])
            # A rank's share of the frame is small enough for the host's launch calls to matter: most frames are
            # replayed from a recorded graph (no kernel timing); every TIMED_EVERY-th one is launched plainly with
            # timing events on its traversal kernels, which is where the per-kernel figures come from.
            timed = 1 if ((k - state["timed_base"]) % TIMED_EVERY == 0) else 0
            c = k % len(ctxs)
            if timing_state[c] != timed:
                ctxs[c].set_option("kernel_timing", timed)
                timing_state[c] = timed
            if fpb > 1 or cams is not None:
                ctxs[c].render_frames_async(fpb, cams, w, h, args.depth, 10, rank, world, slabs[k % ring].data_ptr(), False)
            else:
                ctxs[c].render_tiles_async(w, h, args.depth, 10, rank, world, slabs[k % ring].data_ptr())
        if state["enqueued"] - state["collected"] > in_flight:
            collect()

    def sync():
        while state["collected"] < state["enqueued"]:
            collect()
        finish_gather()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if use_path:      # the path is computed before anything is timed: the loop only hands cameras over
        n_batches_total = (args.warmup + args.steps) * fps // fpb + 3 * ring + 16
        for k in range(n_batches_total):
            cams_of(k)
    reset_acc()
    # The interpreter's cyclic collector stays out of the measurements: a full collection of a process that has imported
    # torch takes 35-40 ms, comes once after about 70 000 container allocations (a thousand frames of this loop) and
    # would land in the timed region of one run in a few (tools/stall_probe.py shows it). Nothing here builds cycles.
    gc.collect()
    gc.freeze()
    gc.disable()
    if multi and not gloo:      # set-up, like the BVH build: every (context, slab) pair renders once plainly and is recorded once
        for _ in range(3 * ring):
            step()
        sync()
    for _ in range(args.warmup * fps // fpb):
        step()
    sync()
    reset_acc()
    state["timed_base"] = state["enqueued"]      # the first batch of the timed region carries timing events
    n_frames = args.steps * fps
    t0 = time.perf_counter()
    for _ in range(n_frames // fpb):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    image = images[(state["enqueued"] - 1) % len(images)] if not multi else image
    total_traced = acc["traced"]
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tr = torch.tensor([acc["traced"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        total_traced = float(tr.item())

    sharded_equal = None
    last_cams = cams_of(state["enqueued"] - 1)      # cameras of the batch whose frames `image` holds
    if multi and rank == 0:   # outside the timed region: the assembled frames against single-GPU frames with the same cameras
        whole = torch.zeros(h * w * 3, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()      # torch fills the buffer on ITS stream; the library renders on its own
        sharded_equal = True
        for f in range(fpr):          # rank 0 holds the first fpr frames of the batch (all of them without the all-to-all)
            if last_cams is None:
                ctx.render_device(w, h, args.depth, 10, whole.data_ptr())
            else:
                ctx.render_frames(1, [last_cams[f]], w, h, args.depth, 10, 0, 1, whole.data_ptr(), True)
            torch.cuda.synchronize()
            sharded_equal = sharded_equal and bool(torch.equal(whole, image[f * h * w * 3:(f + 1) * h * w * 3]))
        del whole
    frames_differ = None
    if rank == 0 and fpb > 1 and (not multi or fpr > 1):      # the frames of a batch are different frames (or, with --cameras fixed, copies)
        torch.cuda.synchronize()
        frames_differ = not bool(torch.equal(image[:h * w * 3], image[h * w * 3:2 * h * w * 3]))
        if use_path and not frames_differ:
            raise SystemExit("bench.py: frames 0 and 1 of a batch with distinct cameras came out identical")
    gather_isolated_ms = None
    if abi:                    # the exchange step alone: 10 gathers to the root, nothing else on the GPU
        torch.cuda.synchronize()
        dist.barrier()
        t_g0 = time.perf_counter()
        for _ in range(10):
            ctx.gather_frame(w, h, slabs[0].data_ptr(), image.data_ptr(), 0)
        gather_isolated_ms = (time.perf_counter() - t_g0) * 1e3 / 10
        ctx.render_tiles(w, h, args.depth, 10, rank, world, slabs[0].data_ptr())      # the image the check below looks at
        ctx.gather_frame(w, h, slabs[0].data_ptr(), image.data_ptr(), 0)
    elif multi and not gloo:   # the exchange step alone: 10 all-gathers of the slabs, nothing else on the GPU
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if a2a:
                dist.all_to_all_single(gathered[0], slabs[0])
            else:
                dist.all_gather_into_tensor(gathered[0], slabs[0])
        e1.record()
        torch.cuda.synchronize()
        gather_isolated_ms = e0.elapsed_time(e1) / 10
    if rank == 0:
        n_prims = flat.n_prims
        b_ray = synth.bytes_per_ray(n_prims)
        src_hash = rtx.source_hash()      # of the library that is loaded, not of the source tree
        # dominant kernel = the traversal kernel with the larger share of device time on this rank
        if acc["shadow_ms"] >= acc["closest_ms"] and acc["shadow_launches"]:
            kname, kkey, kms, kl, krays = "k_shadow (shadow-ray traversal: any-hit, or closest-hit where materials are transparent)", "shadow", acc["shadow_ms"], acc["shadow_launches"], acc["shadow_rays"]
        else:
            kname, kkey, kms, kl, krays = ("k_trace_queue (closest-hit traversal)", "closest", acc["closest_ms"], acc["closest_launches"],
                                           acc["closest_rays"])
        avg_ms = kms / max(kl, 1)
        rays_per_launch = krays / max(kl, 1)
        algorithmic = rays_per_launch * b_ray / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # Counters of the SAME build and workload (profiles/pmc_traffic.json is stamped with the hash of the library's
        # sources and with the workload it was collected on); anything else is not quoted.
        pmc = None
        workload_key = "config%d_grid%d_%dx%d_depth%d_spec%g_glass%d" % (args.config, args.grid, w, h, args.depth, args.spec, args.glass)
        if fpb > 1:
            workload_key += "_fpb%d" % fpb      # counters per launch belong to a launch of that many frames
        if use_path:
            workload_key += "_campath"          # ... of that many DIFFERENT frames
        default_workload = (args.config == 3 and args.grid == 708 and (w, h) == (1920, 1080) and args.depth == 10
                            and args.spec == 0.0 and args.glass == 0 and not multi)
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if not multi and os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    allpmc = json.load(f)
                entry = allpmc.get("workloads", {}).get(workload_key)
                if entry and entry.get("source_hash") == src_hash:
                    pmc = entry["kernels"][kkey]
                    pmc_cpi = allpmc.get("valu_cycles_per_wave_instruction")
            except Exception:
                pmc = None
        traffic = pmc["bytes_per_launch"] if pmc else None
        measured = (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None
        # device-to-device copy rate of this GPU (read + write bytes), outside the timed region: the practical HBM ceiling
        copy_gbs = None
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except Exception:
            copy_gbs = None
        # `achieved` is HBM traffic from the counters over the launch time, so `frac` is a real share of the HBM
        # peak; the contract's algorithmic figure (every box on a root-to-leaf path charged to HBM) is reported
        # beside it and can exceed the peak because L2 and the Infinity Cache serve most node fetches.
        roof = {"bound": "hbm", "achieved": measured, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (measured / HBM_PEAK_GBS) if measured is not None else None, "traffic": traffic, "kernel": kname,
                "avg_launch_ms": avg_ms, "rays_per_launch": rays_per_launch, "bytes_per_ray": b_ray,
                "algorithmic_GBps": algorithmic, "algorithmic_frac": algorithmic / HBM_PEAK_GBS,
                "traffic_over_algorithmic": (traffic / (rays_per_launch * b_ray)) if (traffic and rays_per_launch) else None,
                "device_copy_GBps": copy_gbs, "source_hash": src_hash, "workload_key": workload_key,
                "counters_from": ("profiles/pmc_traffic.json[%s], same source hash" % workload_key) if pmc else None}
        if pmc:
            for k in ("lane_utilisation", "valu_issue_frac", "valu_instructions_per_launch",
                      "fetch_iterations_per_launch", "l2_hit_rate", "l1_hit_rate", "l1_miss_latency_cycles", "ta_busy_frac", "wave_wait_frac"):
                if k in pmc:
                    roof[k] = pmc[k]
            roof["limiter"] = ("VALU issue and dependent-fetch latency, not HBM: HBM-side traffic is %.0f %% of the algorithmic bytes"
                               % (100.0 * traffic / max(rays_per_launch * b_ray, 1.0)))
        else:
            roof["note"] = ("no PMC counters of this build and workload are committed (source hash %s): achieved / frac / traffic are null "
                            "rather than quoted from another build; algorithmic_GBps is the contract figure (ray in + 2*ceil(log2 N) boxes "
                            "+ one primitive + hit out per ray) over the launch time" % src_hash)
        frames = max(n_frames, 1)
        closest_per_launch = acc["closest_rays"] / max(acc["closest_launches"], 1)
        shadow_per_launch = acc["shadow_rays"] / max(acc["shadow_launches"], 1)
        closest_avg_ms = acc["closest_ms"] / max(acc["closest_launches"], 1)
        shadow_avg_ms = acc["shadow_ms"] / max(acc["shadow_launches"], 1)
        out = {
            "metric": "Mrays/sec (primary+shadow+secondary) per GPU and per node; frame time",
            "value": total_traced / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            # weak: per-GPU work per step is fixed (a step is 100 x N frames, a rank renders 1/N of each);
            # strong: the step is the same frames whatever N is
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic %d-triangle tessellated height field%s + 4 point lights at %dx%d, depth %d, "
                                   "LBVH traversal, scene and BVH resident in HBM; one step = %d frames, rendered %d per set of launches; %s"
                                   % (flat.n_tris, (" + %d spheres" % flat.n_spheres) if flat.n_spheres else "", w, h, args.depth, fps, fpb,
                                      ("every frame has its own camera (frame k of a fly-around: the scene's camera turned by 0.004 k rad about "
                                       "the vertical axis)" + ("" if fpb == 1 else ", so a batch holds %d DIFFERENT frames" % fpb)) if use_path
                                      else ("the scene's camera for every frame: a batch holds %d COPIES of one frame" % fpb)),
                       "cameras": "path: %d distinct cameras per batch" % fpb if use_path else "fixed: %d copies per batch" % fpb,
                       "batch_frames_differ": frames_differ,
                       "n_prims": n_prims, "width": w, "height": h, "depth": args.depth, "frames_per_step": fps,
                       "frames_per_batch": fpb, "batch_latency_ms_approx": elapsed / frames * 1e3 * fpb * (len(ctxs) + 1),
                       "parallelism": ("tile-sharded x%d (%s, %s collectives, %s)" % (world, args.backend, args.collectives,
                                       "all-to-all: each frame assembled on one rank" if a2a else "gather to the root" if abi else "all-gather")) if multi else "single GPU",
                       "ranks": world, "batches_in_flight_per_gpu": len(ctxs) + 1},
            "ms_per_frame": elapsed / frames * 1e3,
            "rays_per_frame": total_traced / frames,
            "primary_hit_fraction": (acc["shaded_primary"] / acc["primary"]) if acc.get("primary") else None,
            "sharded_frame_equals_single_gpu_frame": sharded_equal,
            "device_frame_ms_rank0": acc["render_ms"] / frames,
            "build_ms": build["build_ms"], "upload_ms": build["upload_ms"],
            "scene_broadcast_ms": bcast_ms, "broadcast_collective_ms": bcast_collective_ms, "scene_setup_ms_rank0": scene_setup_ms,
            "gather_ms_isolated_rank0": gather_isolated_ms,
            "gather_wait_ms_per_frame_rank0": (gather_wait[0] / frames * 1e3) if multi else None,
            "bvh": {k: build[k] for k in ("n_nodes", "n_leaves", "max_depth", "key_bits", "uses_bvh")},
            "kernels": {"closest_ms": closest_avg_ms, "shadow_ms": shadow_avg_ms,
                        "closest_Mrays_s": (closest_per_launch / (closest_avg_ms * 1e-3) / 1e6) if closest_avg_ms > 0 else None,
                        "shadow_Mrays_s": (shadow_per_launch / (shadow_avg_ms * 1e-3) / 1e6) if shadow_avg_ms > 0 else None,
                        "closest_rays_per_launch": closest_per_launch, "shadow_rays_per_launch": shadow_per_launch},
            "roofline": roof,
        }
        single_gpu_extras = (not multi) and len(ctxs) == 1 and not args.no_extras
        if single_gpu_extras:
            # First-class beside `value`: the SAME camera path rendered one frame per set of launches on one context
            # (two frames in flight) -- the configuration as BASELINE words it, one frame at a time. Outside the timed
            # region of `value`, timed the same way (wall clock around enqueue ... last frame collected).
            try:
                n1 = {3: 300, 4: 20, 5: 6}[args.config]
                warm1 = {3: 12, 4: 2, 5: 1}[args.config]
                path1 = synth.camera_path(flat_cam, 0, n1 + warm1) if use_path else None
                bufs1 = [torch.zeros(h * w * 3, dtype=torch.uint8, device=dev) for _ in range(3)]
                torch.cuda.synchronize()

                def enqueue1(c, k, buf):
                    if path1 is None:
                        c.render_device_async(w, h, args.depth, 10, buf.data_ptr())
                    else:
                        c.render_frames_async(1, [path1[k]], w, h, args.depth, 10, 0, 1, buf.data_ptr(), True)

                def run1(k0, n):      # frame by frame on one context, two frames in flight
                    tot = {"closest_ms": 0.0, "shadow_ms": 0.0, "frames": 0, "traced": 0}
                    def take(st1):
                        tot["closest_ms"] += st1.closest_ms; tot["shadow_ms"] += st1.shadow_ms
                        tot["frames"] += 1; tot["traced"] += st1.traced_rays
                    for k in range(n):
                        if k >= 2:
                            take(ctx.render_wait())
                        enqueue1(ctx, k0 + k, bufs1[k % 3])
                    for _ in range(min(n, 2)):
                        take(ctx.render_wait())
                    torch.cuda.synchronize()
                    return tot
                # Twice: with the events that time the traversal launches (the library's default; where the per-kernel
                # figures come from) and without them, as plain launches (no graph replay: with an idle host thread a
                # replayed frame costs the GPU ~10 us more than its launches cost the host).
                ctx.set_option("kernel_timing", 1)
                run1(0, warm1)
                t1 = time.perf_counter()
                tot1 = run1(warm1, n1)
                dt1 = time.perf_counter() - t1
                ctx.set_option("kernel_timing", 0)
                ctx.set_option("graphs", 0)
                run1(0, warm1)
                t1 = time.perf_counter()
                totp = run1(warm1, n1)
                dtp = time.perf_counter() - t1
                ctx.set_option("graphs", 1)
                ctx.set_option("kernel_timing", 1)
                out["value_single_frame"] = totp["traced"] / dtp / 1e6
                out["single_frame"] = {"Mrays_s": totp["traced"] / dtp / 1e6, "ms_per_frame": dtp / n1 * 1e3, "frames": n1,
                                       "rays_per_frame": totp["traced"] / n1,
                                       "with_kernel_timing": {"Mrays_s": tot1["traced"] / dt1 / 1e6, "ms_per_frame": dt1 / n1 * 1e3,
                                                              "closest_ms_per_frame": tot1["closest_ms"] / n1,
                                                              "shadow_ms_per_frame": tot1["shadow_ms"] / n1},
                                       "closest_ms_per_frame": tot1["closest_ms"] / n1, "shadow_ms_per_frame": tot1["shadow_ms"] / n1,
                                       "note": "the same camera path, one frame per set of launches on one context, two frames in flight; "
                                               "wall clock, outside the timed region of `value`; options kernel_timing 0 and graphs 0 "
                                               "(`with_kernel_timing`: the same frames with the library's default options, which "
                                               "put four events around the traversal launches of every frame)"}
                # one of those frames against the batch path: frame 0 of the headline's last batch, rendered alone
                if last_cams is not None:
                    ctx.render_frames(1, [last_cams[0]], w, h, args.depth, 10, 0, 1, bufs1[0].data_ptr(), True)
                else:
                    ctx.render_device(w, h, args.depth, 10, bufs1[0].data_ptr())
                torch.cuda.synchronize()
                out["single_frame"]["frame_equals_headline_frame"] = bool(torch.equal(bufs1[0], image[:h * w * 3]))
                del bufs1
            except Exception as e:   # never at the expense of the headline line
                out["single_frame"] = {"error": str(e)}
        if default_workload and single_gpu_extras:
            # Supplementary, outside the timed region and not `value`: frames alternating between three contexts
            # (streams), where one frame's kernel tails overlap the next frame's kernels, and the headline's batches
            # alternating between two contexts.
            try:
                extra = [ctx.clone(), ctx.clone()]    # share the scene and the BVH, own stream and workspace
                for c in extra:
                    c.set_option("kernel_timing", 0)
                ctx.set_option("kernel_timing", 0)
                trio = [ctx] + extra
                bufs = [torch.zeros(h * w * 3, dtype=torch.uint8, device=dev) for _ in range(6)]
                path3 = synth.camera_path(flat_cam, 0, 312) if use_path else None
                torch.cuda.synchronize()

                def run3(k0, n):
                    rays = 0
                    for k in range(n):
                        if k >= 3:
                            rays += trio[k % 3].render_wait().traced_rays
                        if path3 is None:
                            trio[k % 3].render_device_async(w, h, args.depth, 10, bufs[k % 6].data_ptr())
                        else:
                            trio[k % 3].render_frames_async(1, [path3[k0 + k]], w, h, args.depth, 10, 0, 1, bufs[k % 6].data_ptr(), True)
                    for k in range(n, n + 3):
                        rays += trio[k % 3].render_wait().traced_rays
                    torch.cuda.synchronize()
                    return rays
                run3(0, 12)
                t3 = time.perf_counter()
                rays3 = run3(12, 300)
                dt3 = time.perf_counter() - t3
                out["supplementary"] = {"three_contexts_ms_per_frame": dt3 / 300 * 1e3, "three_contexts_Mrays_s": rays3 / dt3 / 1e6}
                if fpb > 1:
                    pair = [ctx, extra[0]]
                    bufs_b = [torch.zeros(fpb * h * w * 3, dtype=torch.uint8, device=dev) for _ in range(4)]
                    torch.cuda.synchronize()

                    def run2(k0, n):
                        rays = 0
                        for k in range(n):
                            if k >= 2:
                                rays += pair[k % 2].render_wait().traced_rays
                            pair[k % 2].render_frames_async(fpb, cams_of(k0 + k), w, h, args.depth, 10, 0, 1, bufs_b[k % 4].data_ptr(), True)
                        for k in range(n, n + 2):
                            rays += pair[k % 2].render_wait().traced_rays
                        torch.cuda.synchronize()
                        return rays
                    run2(0, 8)
                    n2 = max(300 // fpb, 8)
                    t2 = time.perf_counter()
                    rays2 = run2(8, n2)
                    dt2 = time.perf_counter() - t2
                    out["supplementary"]["batches_on_two_contexts"] = {"ms_per_frame": dt2 / (n2 * fpb) * 1e3, "Mrays_s": rays2 / dt2 / 1e6}
                    del bufs_b
                ctx.set_option("kernel_timing", 1)
                for c in extra:
                    c.close()
            except Exception as e:   # never at the expense of the headline line
                out["supplementary"] = {"error": str(e)}
        if default_workload and not args.no_extras:
            # The target the workload description names: the >= 1M-triangle scene at 3840x2160 (same scene, same
            # context, same camera path, outside the timed region; its own object, not `value`).
            try:
                w4, h4 = 3840, 2160
                bufs4 = [torch.zeros(h4 * w4 * 3, dtype=torch.uint8, device=dev) for _ in range(3)]
                path4 = synth.camera_path(flat_cam, 0, 110) if use_path else None
                torch.cuda.synchronize()
                ctx.set_option("kernel_timing", 0)
                ctx.set_option("graphs", 0)

                def run4(k0, n):      # two frames in flight, like the 1080p figure above
                    rays = 0
                    for k in range(n):
                        if k >= 2:
                            rays += ctx.render_wait().traced_rays
                        if path4 is None:
                            ctx.render_device_async(w4, h4, args.depth, 10, bufs4[k % 3].data_ptr())
                        else:
                            ctx.render_frames_async(1, [path4[k0 + k]], w4, h4, args.depth, 10, 0, 1, bufs4[k % 3].data_ptr(), True)
                    for _ in range(min(n, 2)):
                        rays += ctx.render_wait().traced_rays
                    torch.cuda.synchronize()
                    return rays
                run4(0, 5)
                t4 = time.perf_counter()
                rays4 = run4(5, 100)
                dt4 = time.perf_counter() - t4
                out["target_4k"] = {"workload": "the same %d-triangle scene at %dx%d, depth %d, 1 GPU, 100 frames of the same camera path, frame by frame, two frames in flight, options kernel_timing 0 and graphs 0"
                                                % (flat.n_tris, w4, h4, args.depth),
                                    "Mrays_s": rays4 / dt4 / 1e6, "ms_per_frame": dt4 / 100 * 1e3, "rays_per_frame": rays4 / 100}
                del bufs4
                try:     # two 4K frames per set of launches (as many rays as the headline's eight 1080p frames)
                    bufs4 = [torch.zeros(2 * h4 * w4 * 3, dtype=torch.uint8, device=dev) for _ in range(3)]
                    torch.cuda.synchronize()

                    def run4b(k0, n):
                        rays = 0
                        for k in range(n):
                            if k >= 2:
                                rays += ctx.render_wait().traced_rays
                            cams2 = [path4[(k0 + 2 * k) % 110], path4[(k0 + 2 * k + 1) % 110]] if path4 is not None else None
                            ctx.render_frames_async(2, cams2, w4, h4, args.depth, 10, 0, 1, bufs4[k % 3].data_ptr(), True)
                        for _ in range(min(n, 2)):
                            rays += ctx.render_wait().traced_rays
                        torch.cuda.synchronize()
                        return rays
                    run4b(0, 3)
                    t4 = time.perf_counter()
                    rays4 = run4b(6, 50)
                    dt4 = time.perf_counter() - t4
                    out["target_4k"]["two_frames_per_batch"] = {"Mrays_s": rays4 / dt4 / 1e6, "ms_per_frame": dt4 / 100 * 1e3}
                    del bufs4
                except Exception as e:
                    out["target_4k"]["two_frames_per_batch"] = {"error": str(e)}
            except Exception as e:
                out["target_4k"] = {"error": str(e)}
            ctx.set_option("graphs", 1)
            ctx.set_option("kernel_timing", 1)
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, w, h, args.depth, 10, args.cpu_stride)
            if default_workload and not args.no_extras:
                try:     # all host cores this process may use (not the reference's configuration, which is one thread)
                    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
                    out["cpu_baseline_all_cores"] = cpu_baseline(flat, w, h, args.depth, 10, max(args.cpu_stride // 3, 1), threads=cores)
                except Exception as e:
                    out["cpu_baseline_all_cores"] = {"error": str(e)}
                try:     # about half a minute, most of it the tree build
                    out["cpu_baseline_bvh"] = cpu_baseline_bvh(flat, w, h, args.depth, 10, 4)
                except Exception as e:
                    out["cpu_baseline_bvh"] = {"error": str(e)}
            if "target_4k" in out and "Mrays_s" in out["target_4k"] and out["cpu_baseline"]["value"] > 0:
                t4k = out["target_4k"]
                t4k["cpu_frame_s_extrapolated"] = t4k["rays_per_frame"] / (out["cpu_baseline"]["value"] * 1e6)
                t4k["cpu_note"] = ("single-thread CPU oracle rate measured on the 1080p sample above, applied to this frame's ray count "
                                   "(the exhaustive loop costs the same per ray at any resolution)")
        print(json.dumps(out), flush=True)
    if asm_ctx is not None:
        asm_ctx.close()
    for c in ctxs:
        c.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
