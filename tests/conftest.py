import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with `-m gpu`; when the marker filter lets
    # them through on a box without a GPU they must fail loudly, not skip.
    pass
