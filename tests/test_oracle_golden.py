"""The oracle against its committed golden frames (tests/golden/frames_100x100.npz,
written by tests/golden/make_golden.py), plus analytically known frames."""
import numpy as np
import pytest

from conftest import BUNDLED, GOLDEN_DIR, scene_path

COUNT_KEYS = ["primary_rays", "shadow_rays", "reflect_rays", "refract_rays", "shaded_nodes", "last_level_rays", "panic_pixels"]


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN_DIR + "/frames_100x100.npz")


@pytest.mark.parametrize("name", BUNDLED)
def test_oracle_matches_golden(name, orc, golden):
    if name in ("test09", "test10", "test07", "test08"):
        threads = 8   # identical results; only shortens the run
    else:
        threads = 1
    r = orc.OracleScene.parse(scene_path(name)).render(100, 100, 10, 10, threads=threads)
    assert np.array_equal(r["prim_id"], golden[name + "_prim"])
    assert np.array_equal(r["t"].view(np.uint32), golden[name + "_t"].view(np.uint32))
    assert np.array_equal(r["rgb"], golden[name + "_rgb"])
    assert [r["stats"][k] for k in COUNT_KEYS] == list(golden[name + "_counts"])


def test_scenes_without_light_are_black(golden):
    # test00 / test11 have no lights and zero ambient: every pixel is black although rays hit
    for name in ("test00", "test11"):
        assert (golden[name + "_prim"] >= 0).any()
        assert not golden[name + "_rgb"].any()


def test_no_bundled_frame_panics(golden):
    for name in BUNDLED:
        assert golden[name + "_counts"][6] == 0


def test_exhaustive_and_reference_bvh_agree_without_ties(orc):
    """The restated reference BVH (bvh.rs) returns the same hits as the exhaustive loop on a
    scene without exact t ties."""
    s = orc.OracleScene.parse(scene_path("test04"))
    a = s.render(40, 40, 3, 1, use_bvh=False)
    b = s.render(40, 40, 3, 1, use_bvh=True)
    assert np.array_equal(a["prim_id"], b["prim_id"])
    assert np.array_equal(a["rgb"], b["rgb"])


def test_stride_renders_subset(orc):
    s = orc.OracleScene.parse(scene_path("test01"))
    full = s.render(64, 48, 10, 10)
    sub = s.render(64, 48, 10, 10, stride=(16, 8))
    assert np.array_equal(sub["rgb"][::8, ::16], full["rgb"][::8, ::16])
    assert sub["stats"]["primary_rays"] == 4 * 6
