import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test waits for ever: when a GPU test has not come back after 25 minutes pytest-timeout (when it is
    installed) dumps the stacks of all threads and ENDS the pytest process (method="thread": the test may be
    blocked inside a C call, where a signal-based timeout would not be served) -- the session fails instead of hanging.  (The bound is generous on
    purpose: the one stall seen so far -- three times in some forty runs -- was the first `import torch` of a process
    on a fresh GPU box waiting for the image to page in, inside `importlib`'s stat calls.)"""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(1500, method="thread"))


def product():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("rust-compression_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    return product()


def sample(i):
    with open(os.path.join(GOLDEN, "sample%d.ref" % i), "rb") as f:
        return f.read()
