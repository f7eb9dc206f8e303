import os
import sys

import pytest

# Heavy imports happen HERE, at collection time, outside pytest-timeout's per-test clock: on a fresh GPU box the
# first `import torch` pages ~2 GB of libraries in and has been seen to take more than five minutes -- inside a
# test that is a spurious timeout (it looked like a hang of the test that happened to import it first).
try:
    import numpy  # noqa: F401
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (builds oracle/_build/libdartray_oracle.so on first use)."""
    import oracle.binding as binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def hip():
    """The product library; on a GPU box also selects device 0."""
    from dartray_amd import _abi
    _abi.lib()
    return _abi


@pytest.fixture(scope="session")
def gpu(hip):
    hip.init(0)
    return hip


GOLDEN = os.path.join(ROOT, "tests", "golden")
