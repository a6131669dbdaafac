import os
import sys

import pytest

# the tests force every tile shape / kernel variant through the pfk_debug_set_* knobs, which are inert in a process that did
# not opt in (include/pfk.h)
os.environ.setdefault("PFK_DEBUG_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the upstream reference tree at /root/reference")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("`-m gpu` tests need a visible GPU; there is no CPU fallback path to fall back to")
    import ptlflow_amd

    ptlflow_amd.load_native()
    return torch.device("cuda:0")
