import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU runs: the suite's tensors are tiny and the build container's 8 vCPUs are shared -- with torch's default 8 intra-op threads a
    # barrier regularly waits for a descheduled vCPU and single tests take 50-100 x their time (the suite: 1 - 15 minutes, measured);
    # two threads: 75 s.  Child processes (gloo ranks, bench self-launch) inherit the setting through the environment.
    try:
        import torch
        if not torch.cuda.is_available():
            torch.set_num_threads(2)
            os.environ.setdefault("OMP_NUM_THREADS", "2")
            os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not asked for."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
