import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def pytest_sessionstart(session):
    """LR_TUNE="knob=value,..." runs the whole session under forced kernel variants (lr_tune_set): a candidate kernel is put
    through the parity suite before it becomes the default."""
    spec = os.environ.get("LR_TUNE")
    if spec:
        import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch ships)
        from luciddreamer_amd import _lib
        for kv in spec.split(","):
            k, v = kv.split("=")
            _lib.tune_set(k.strip(), int(v))
