import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def built_lib():
    """libeoc_tfhe_gpu.so, (re)built in-tree when sources changed and hipcc is available."""
    import shutil
    import eoc_tfhe_amd
    from eoc_tfhe_amd import build as b
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        b.build()
    return eoc_tfhe_amd.lib()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
