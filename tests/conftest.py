import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# torch FIRST, before any test has loaded libblazeseq_hip.so: the library and torch share one HIP runtime in either import order
# (blazeseq_amd/_lib.py loads torch's copy of libamdhip64 ahead of itself), but twice in round 5 a lazy `import torch` inside a test,
# AFTER minutes of GPU work in the same session, did not return within the test's time limit (not reproduced in isolation: 3-12 s in
# five tries, scripts/probes/import_order_probe.py).  The whole suite always had torch imported at collection time (a module-level import
# in one test file); this makes a single test file behave the same.
try:
    import torch  # noqa: F401
except ImportError:   # (the CPU-only tests that need no torch still run)
    pass

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORPUS = os.path.join(GOLDEN, "corpus")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def corpus_dir():
    return CORPUS
