"""The exchange steps behind the C ABI (rtx_comm_init, rtx_scene_broadcast, rtx_gather_frame) with the one rank a
one-GPU box allows: RCCL is loaded and initialised, the scene arrives through the broadcast path (device buffers ->
rtx_scene_set_device) and the frame through gather-to-root + assembly, and both equal the plain single-context path."""
import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu


def test_one_rank_broadcast_and_gather_equal_the_plain_path(rtx, gpu_ctx_factory):
    torch = pytest.importorskip("torch")
    flat = rtx.parse_scene(scene_path("test08"))
    plain = gpu_ctx_factory(flat)
    w, h = 203, 117                      # not multiples of the tile size
    rgb_ref, st_ref = plain.render(w, h, 10, 10)
    cid = rtx.comm_unique_id()
    assert len(cid) == 128
    with rtx.Context(0) as ctx:
        with pytest.raises(rtx.RtxError):                 # no communicator yet
            ctx.scene_broadcast(flat, 0)
        ctx.comm_init(0, 1, cid)
        with pytest.raises(rtx.RtxError):                 # already initialised
            ctx.comm_init(0, 1, cid)
        ctx.scene_broadcast(flat, 0)
        ctx.build()
        tpr = rtx.tiles_per_rank(w, h, 1)
        slab = torch.zeros(tpr * 192, dtype=torch.uint8, device="cuda:0")
        img = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        for _ in range(2):                                # the second frame reuses the gather buffer
            st = ctx.render_tiles(w, h, 10, 10, 0, 1, slab.data_ptr())
            ctx.gather_frame(w, h, slab.data_ptr(), img.data_ptr(), 0)
        torch.cuda.synchronize()
        assert st.traced_rays == st_ref.traced_rays
        assert np.array_equal(img.cpu().numpy().reshape(h, w, 3), rgb_ref)
        # a batch of three frames through the batch form of the gather (the scene's camera for all: three equal images)
        slabs = torch.zeros(3 * tpr * 192, dtype=torch.uint8, device="cuda:0")
        imgs = torch.zeros(3 * h * w * 3, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        ctx.render_frames(3, None, w, h, 10, 10, 0, 1, slabs.data_ptr(), False)
        ctx.gather_frames(3, w, h, slabs.data_ptr(), imgs.data_ptr(), 0)
        torch.cuda.synchronize()
        for k in range(3):
            assert np.array_equal(imgs.cpu().numpy().reshape(3, h, w, 3)[k], rgb_ref)
        ctx.comm_destroy()
        ctx.comm_destroy()                                # idempotent
