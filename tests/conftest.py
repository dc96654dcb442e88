import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# tests/test_gpu_experimental.py is the job of libfasta_hip_experimental.so (csrc/fh_experimental.h): it is collected only when
# $FASTA_HIP_LIB points the binding at that library, and left out (not skipped) of every run against the shipped one
collect_ignore = [] if "experimental" in os.path.basename(os.environ.get("FASTA_HIP_LIB", "")) else ["test_gpu_experimental.py"]
