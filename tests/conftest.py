import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One rt_context on cuda:0 for the whole GPU session (fails loudly when the extension or GPU is missing)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from raytrace_amd import capi
    ctx = capi.Context(0)
    # the shared context runs the megakernel unless a test switches it; wavefront tests pick small pools
    ctx.set_option(capi.OPT_MODE, capi.MODE_MEGAKERNEL)
    ctx.set_option(capi.OPT_WF_CAPACITY_LOG2, 20)
    # one closest-hit walk per sample, as upstream executes it: the tests compare box and triangle test counts with the
    # oracle's; the shared first walk has tests of its own
    ctx.set_option(capi.OPT_SHARE_PRIMARY, 0)
    ctx.set_option(capi.OPT_SHARE_EQUAL, 0)
    yield ctx
    ctx.close()
