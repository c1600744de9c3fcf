"""rtx_intersect_batch_device_noalloc (the entry point that allocates nothing: its traversal-stack
overflow area is a static device array of the library) against rtx_intersect_batch_device, through
the C ABI: primitive ids and t bit for bit, on two calls in a row, the second with a smaller input."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _query_both(ctx, torch, ori, d):
    n = ori.shape[0]
    dev = torch.device("cuda", 0)
    d_o = torch.from_numpy(ori).to(dev).contiguous()
    d_d = torch.from_numpy(d).to(dev).contiguous()
    p_ref = torch.full((n,), -7, dtype=torch.int32, device=dev)
    t_ref = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    p_new = torch.full((n,), -9, dtype=torch.int32, device=dev)
    t_new = torch.full((n,), -9.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()          # the library's stream does not wait for torch's
    ctx.intersect_batch_device(d_o.data_ptr(), d_d.data_ptr(), p_ref.data_ptr(), t_ref.data_ptr(), n)
    ctx.intersect_batch_device_noalloc(d_o.data_ptr(), d_d.data_ptr(), p_new.data_ptr(), t_new.data_ptr(), n)
    torch.cuda.synchronize()          # the no-allocation call only enqueues
    return p_ref.cpu().numpy(), t_ref.cpu().numpy(), p_new.cpu().numpy(), t_new.cpu().numpy()


def _rays(flat, rng, n):
    """Half of the rays start above the scene and point down into it, half start on the surface."""
    half = n // 2
    ori = np.empty((n, 3), np.float32)
    d = np.empty((n, 3), np.float32)
    ori[:half] = rng.uniform(-1.2, 1.2, (half, 3)).astype(np.float32)
    ori[:half, 1] = np.float32(1.5)
    d[:half] = rng.normal(size=(half, 3)).astype(np.float32)
    d[:half, 1] = -np.abs(d[:half, 1]) - np.float32(0.2)
    tri = flat.tri_pos[rng.integers(0, flat.n_tris, n - half)]
    bary = rng.dirichlet((1, 1, 1), n - half).astype(np.float32)
    ori[half:] = (tri * bary[:, :, None]).sum(axis=1).astype(np.float32)
    ori[half:, 1] += np.float32(1e-4)
    d[half:] = rng.normal(size=(n - half, 3)).astype(np.float32)
    d[half::2, 1] = np.abs(d[half::2, 1]) * np.float32(0.05)     # grazing rays: the long traversals
    return ori, d


def test_noalloc_equals_intersect_batch_device_twice(rtx, gpu_ctx_factory):
    torch = pytest.importorskip("torch")
    synth = importlib.import_module("rust-raytracer_amd.synth")
    flat = synth.heightfield_scene(224, glass_lattice=4)          # 100 352 triangles + 16 spheres: LBVH path
    ctx = gpu_ctx_factory(flat)
    assert ctx.build_stats["uses_bvh"] == 1
    rng = np.random.default_rng(23)
    for n in (300000, 4097):                                      # the second call is smaller and reuses the static area
        ori, d = _rays(flat, rng, n)
        p_ref, t_ref, p_new, t_new = _query_both(ctx, torch, ori, d)
        assert (p_ref >= 0).mean() > 0.3
        assert np.array_equal(p_new, p_ref)
        assert np.array_equal(t_new.view(np.uint32), t_ref.view(np.uint32))


def test_noalloc_small_scene_and_limits(rtx, gpu_ctx_factory):
    torch = pytest.importorskip("torch")
    flat = rtx.parse_scene(__import__("os").path.join(__import__("conftest").SCENES_DIR, "test04.ascii"))
    ctx = gpu_ctx_factory(flat, force_mode=2)
    rng = np.random.default_rng(3)
    n = 2000
    ori = np.tile(np.array(list(flat.camera.pos), np.float32), (n, 1))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d += np.array(list(flat.camera.view_dir), np.float32) * np.float32(2.0)
    p_ref, t_ref, p_new, t_new = _query_both(ctx, torch, ori, d)
    assert np.array_equal(p_new, p_ref)
    assert np.array_equal(t_new.view(np.uint32), t_ref.view(np.uint32))
    # an empty batch is fine; more than RTX_NOALLOC_MAX_RAYS is rejected before anything is launched
    ctx.intersect_batch_device_noalloc(0, 0, 0, 0, 0)
    with pytest.raises(rtx.RtxError) as e:
        ctx.intersect_batch_device_noalloc(1, 1, 1, 1, (1 << 26) + 1)
    assert e.value.status == 6 and "RTX_NOALLOC_MAX_RAYS" in str(e.value)


def test_noalloc_calls_leave_free_device_memory_unchanged(rtx, gpu_ctx_factory):
    """What the entry point is for: a call allocates and frees nothing and does not wait for the device. Free device
    memory is the same before and after a run of calls (the first call of the process has already happened above: code
    objects and the static array are loaded), and a call returns while its kernel is still running behind a long one."""
    torch = pytest.importorskip("torch")
    synth = importlib.import_module("rust-raytracer_amd.synth")
    flat = synth.heightfield_scene(224, glass_lattice=4)
    ctx = gpu_ctx_factory(flat)
    rng = np.random.default_rng(29)
    n = 200000
    ori, d = _rays(flat, rng, n)
    dev = torch.device("cuda", 0)
    d_o, d_d = torch.from_numpy(ori).to(dev), torch.from_numpy(d).to(dev)
    p = torch.empty((n,), dtype=torch.int32, device=dev)
    t = torch.empty((n,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ctx.intersect_batch_device_noalloc(d_o.data_ptr(), d_d.data_ptr(), p.data_ptr(), t.data_ptr(), n)     # warm
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(6):
        ctx.intersect_batch_device_noalloc(d_o.data_ptr(), d_d.data_ptr(), p.data_ptr(), t.data_ptr(), n if k % 2 == 0 else n // 3)
    free_enqueued = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    assert free_enqueued == free0 and torch.cuda.mem_get_info()[0] == free0
    assert int((p >= 0).sum()) > 0
