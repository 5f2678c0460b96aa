import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    import exon_amd
    c = exon_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session", autouse=True)
def _test_tools():
    """tools/bin/gen_text and tools/bin/bgzip (synthetic inputs of the pipeline tests): built on demand, plain g++."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.call(["make", "-C", os.path.join(root, "tools"), "-s"])
