import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/): the checker, never the thing under test on the GPU side."""
    import oracle
    oracle.lib()
    oracle.set_threads(oracle.usable_cores())
    return oracle


@pytest.fixture(scope="session")
def zk():
    import zkstark_amd
    zkstark_amd.load()
    return zkstark_amd


@pytest.fixture(scope="session")
def config4_expected(orc):
    """BASELINE.json configs[3] on the CPU oracle, computed once per session (about half a minute): the Merkle root of
    the 2^26-point LDE of the canonical trace (orc.lde + orc.merkle_build) and the first values of the vector."""
    import numpy as np
    log_n = 23
    a = orc.trace_fibsq((1 << log_n) - 1)
    f = orc.lde(a, log_n, 3)
    root = bytes(orc.merkle_build(f)[0])
    head = np.array(f[:64], dtype=np.uint32)
    del f
    return {"log_n": log_n, "root": root, "head": head}
