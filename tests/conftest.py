import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pkv():
    """the product package (ctypes binding over libmi355_paged_kv.so)"""
    import _load_pkg
    return _load_pkg.load()


@pytest.fixture(scope="session")
def O():
    """the CPU checker (oracle/) — test infrastructure"""
    from oracle import cpu_oracle
    cpu_oracle.port()
    return cpu_oracle


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    return {n[:-4]: np.load(os.path.join(d, n)) for n in os.listdir(d) if n.endswith(".npz")}


def cases_of(npz):
    return sorted({k.split("__")[0] for k in npz.files if "__" in k})
