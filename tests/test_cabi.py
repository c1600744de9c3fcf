"""The C-ABI library loads and exports every symbol include/rtx.h declares (no compute
calls without a GPU), and the host-side helpers behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(rtx):
    L = rtx.lib()
    names = declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(L, n), "librtx_hip.so does not export " + n
    assert sorted(rtx.EXPORTED_SYMBOLS) == names


def test_abi_version(rtx):
    assert rtx.lib().rtx_abi_version() == 2


def test_struct_sizes_match_header(rtx):
    assert C.sizeof(rtx.Material) == 56
    assert C.sizeof(rtx.Light) == 40
    assert C.sizeof(rtx.Camera) == 44
    assert C.sizeof(rtx.BuildStats) == 40
    assert C.sizeof(rtx.RenderStats) == 7 * 8 + 3 * 4 + 5 * 4


def test_tiles_per_rank(rtx):
    rd = __import__("importlib").import_module("rust-raytracer_amd.dist")
    for (w, h, world) in [(100, 100, 1), (1920, 1080, 8), (7, 9, 3), (3840, 2160, 4), (8, 8, 2)]:
        assert rtx.tiles_per_rank(w, h, world) == rd.tiles_per_rank(w, h, world)


def test_pack_and_assemble_roundtrip():
    rd = __import__("importlib").import_module("rust-raytracer_amd.dist")
    rng = np.random.default_rng(0)
    for (w, h, world) in [(100, 100, 1), (37, 21, 3), (64, 64, 8), (17, 40, 2)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        slabs = np.stack([rd.pack_slab(img, r, world) for r in range(world)])
        assert np.array_equal(rd.assemble(slabs, w, h, world), img)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_create_without_gpu_fails_cleanly(rtx):
    with pytest.raises(rtx.RtxError) as e:
        rtx.Context(0)
    assert e.value.status in (2, 3)   # RTX_ERR_NO_DEVICE / RTX_ERR_HIP, never a crash or a fallback


def test_null_arguments_are_rejected(rtx):
    L = rtx.lib()
    assert L.rtx_create(0, None) == 1
    assert L.rtx_scene_set(None, None) == 1
    assert L.rtx_build(None, None) == 1
    assert L.rtx_last_error(None) is not None
