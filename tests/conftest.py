import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

MODELS = os.path.join(ROOT, "fixtures", "models")

# The product routes a Tree-kind problem with few rows to the lane program specialised for it at run time (capi.cpp
# tree_prefers_static: rows <= 12).  The tests written for the tree kernel keep it under test by switching that routing off for the
# session; tests/test_gpu_static.py switches it back on and checks the routed problems against the oracle AND the tree kernel.
os.environ.setdefault("IKGPU_TREE_STATIC_ROWS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_built():
    """libikgpu.so and the oracle must exist; build them when this checkout has not been built yet."""
    lib = os.path.join(ROOT, "ik_amd", "libikgpu.so")
    worker = os.path.join(ROOT, "ik_amd", "ikgpu_precompile")   # the run-time compiler's own process (rtc.cpp): without it nothing is compiled at run time
    if not os.path.exists(lib) or not os.path.exists(worker):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")])
    import oracle as O
    O.lib()
    return lib


def urdf_path(name):
    return os.path.join(MODELS, name + ".kin.urdf")
