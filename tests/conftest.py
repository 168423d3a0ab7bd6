import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime loads: easevoice_trainer_amd/__init__.py

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from easevoice_trainer_amd.hip import lib

    lib.lib()  # fail loudly if the extension is missing
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _bf16_build_selected():
    """every test starts with the bfloat16 build of the library selected (hip.lib.set_half): the fp16-mode tests switch to the
    IEEE-half build (libevt_hip_f16.so) and must not leak that selection into the next test"""
    import torch
    from easevoice_trainer_amd.hip import lib

    lib.set_half(torch.bfloat16)
    yield
    lib.set_half(torch.bfloat16)
