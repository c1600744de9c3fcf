import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES_DIR = os.path.join(ROOT, "scenes")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
# The 13 deterministic bundled scenes (area.ascii samples a random number generator in the reference).
BUNDLED = ["sphere"] + ["test%02d" % i for i in range(12)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES_DIR, name + ".ascii")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi


@pytest.fixture(scope="session")
def rtx():
    """The product package (ctypes binding of librtx_hip.so)."""
    import rtx_amd
    return rtx_amd


@pytest.fixture(scope="session")
def gpu_ctx_factory(rtx):
    """Creates contexts on cuda:0 and closes them at the end of the session."""
    made = []

    def make(flat, force_mode=0):
        ctx = rtx.Context(0)
        ctx.set_option("verify_build", 1)      # every tree built by the tests has its refit re-checked box by box
        if force_mode:
            ctx.set_option("force_mode", force_mode)
        ctx.set_scene(flat)
        ctx.build()
        made.append(ctx)
        return ctx

    yield make
    for c in made:
        c.close()
