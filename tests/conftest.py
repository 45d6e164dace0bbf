import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs inside the tests: size torch's intra-op pool to the cores we really have
    import torch
    from oracle.cpu_threads import effective_cores
    torch.set_num_threads(min(effective_cores(), 32))


@pytest.fixture(scope="session")
def have_gpu():
    import torch
    return torch.cuda.is_available()
