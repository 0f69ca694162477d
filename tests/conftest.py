import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import miopen_cache  # noqa: E402

miopen_cache.use()          # compiled MIOpen kernels of the network tests, if present in the tree (before torch is imported)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle.oracle import Reference, reference_available, build_reference
    if not reference_available():
        build_reference()
    if not reference_available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return Reference()


@pytest.fixture(scope="session")
def genre():
    """the product package; fails (does not skip) if libgenre_hip.so is missing"""
    import genre_shapehd_amd
    return genre_shapehd_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
