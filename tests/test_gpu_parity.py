"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
bundled scenes at the reference's default frame (100x100, depth 10, samples 10,
src/main.rs:63-66).

Bars: primary-ray directions, hit primitive ids and hit t are bit-exact; final RGB
is within 1/255 per channel on >= 99.9 % of pixels (the only non-bit-exact step is
powf in the specular term, src/lib.rs:168); ray counts by class are equal."""
import numpy as np
import pytest

from conftest import BUNDLED, scene_path

pytestmark = pytest.mark.gpu

W = H = 100
DEPTH, SAMPLES = 10, 10


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def oracle_frames(orc):
    frames = {}
    for name in BUNDLED:
        s = orc.OracleScene.parse(scene_path(name))
        frames[name] = (s, s.render(W, H, DEPTH, SAMPLES, threads=8))
    return frames


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["auto", "exhaustive", "bvh"])
@pytest.mark.parametrize("name", BUNDLED)
def test_primary_hits_bit_exact(name, mode, rtx, gpu_ctx_factory, oracle_frames):
    flat = rtx.parse_scene(scene_path(name))
    if mode == 2 and flat.n_prims <= 4:
        pytest.skip("scene too small for a BVH")
    ctx = gpu_ctx_factory(flat, mode)
    prim, t, rdir = ctx.primary_hits(W, H)
    ref = oracle_frames[name][1]
    assert np.array_equal(_bits(rdir), _bits(ref["ray_dir"])), "primary ray directions differ"
    assert np.array_equal(prim, ref["prim_id"]), "%d pixels hit a different primitive" % int((prim != ref["prim_id"]).sum())
    hit = prim >= 0
    assert np.array_equal(_bits(t)[hit], _bits(ref["t"])[hit]), "hit t differs"


@pytest.mark.parametrize("mode", [0, 2], ids=["auto", "bvh"])
@pytest.mark.parametrize("name", BUNDLED)
def test_frame_rgb_and_ray_counts(name, mode, rtx, gpu_ctx_factory, oracle_frames):
    flat = rtx.parse_scene(scene_path(name))
    if mode == 2 and flat.n_prims <= 4:
        pytest.skip("scene too small for a BVH")
    ctx = gpu_ctx_factory(flat, mode)
    rgb, st = ctx.render(W, H, DEPTH, SAMPLES)
    ref = oracle_frames[name][1]
    diff = np.abs(rgb.astype(np.int32) - ref["rgb"].astype(np.int32)).max(axis=2)
    frac = float((diff <= 1).mean())
    assert frac >= 0.999, "only %.4f of pixels within 1/255 (max diff %d)" % (frac, int(diff.max()))
    rs = ref["stats"]
    assert st.primary_rays == rs["primary_rays"]
    assert st.shadow_rays == rs["shadow_rays"]
    assert st.reflect_rays == rs["reflect_rays"]
    assert st.refract_rays == rs["refract_rays"]
    assert st.skipped_rays == rs["last_level_rays"]
    assert st.shaded_nodes == rs["shaded_nodes"]


@pytest.mark.parametrize("name", ["test03", "test05", "test07", "test09"])
def test_intersect_batch_random_rays(name, rtx, gpu_ctx_factory, orc, oracle_frames):
    """Batched IntersectableScene::intersects on rays that are not camera rays."""
    flat = rtx.parse_scene(scene_path(name))
    ctx = gpu_ctx_factory(flat, 0)
    rng = np.random.default_rng(1234)
    n = 4096
    lo = flat.tri_pos.reshape(-1, 3).min(axis=0)
    hi = flat.tri_pos.reshape(-1, 3).max(axis=0)
    ori = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    prim, t = ctx.intersect_batch(ori, d)
    prim_x, t_x = ctx.intersect_batch(ori, d, exhaustive=True)
    ref_prim, ref_t = oracle_frames[name][0].intersect_batch(ori, d)
    assert np.array_equal(prim_x, ref_prim)
    assert np.array_equal(prim, ref_prim)
    hit = ref_prim >= 0
    assert hit.any()
    assert np.array_equal(_bits(t)[hit], _bits(ref_t)[hit])
    assert np.array_equal(_bits(t_x)[hit], _bits(ref_t)[hit])
