import os
import sys

import pytest

# torch bundles its own HIP runtime: it has to be loaded BEFORE libkjarni_ffi.so (which then binds to the same
# runtime, as in bench.py); the other way round torch finds no GPU in a process that already initialised HIP.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
