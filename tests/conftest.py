import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The duo list of the default Lloyd filter (lloyd_duo.hip) is switched on by the engine only where it pays --
    # lists longer than one round of stage-2 blocks, i.e. millions of rows.  The suite's inputs are small: run them
    # WITH the list (=2: always), so that every parity case covers the longer path; tests/test_gpu_duo.py runs the
    # other two settings.
    os.environ.setdefault("KMCUDA_AMD_DUO", "2")


def reference_fixture():
    """The 13000x2 fixture of the reference's own tests (src/test.py:158-169, :581-592)."""
    numpy.random.seed(0)
    arr = numpy.empty((13000, 2), dtype=numpy.float32)
    arr[:2000] = numpy.random.rand(2000, 2) + [0, 0.5]
    arr[2000:4000] = numpy.random.rand(2000, 2) + [0, 1.5]
    arr[4000:6000] = numpy.random.rand(2000, 2) - [0, 0.5]
    arr[6000:8000] = numpy.random.rand(2000, 2) + [0.5, 0]
    arr[8000:10000] = numpy.random.rand(2000, 2) - [0.5, 0]
    arr[10000:] = numpy.random.rand(3000, 2) * 5 - [2, 2]
    return arr


@pytest.fixture(scope="session")
def fixture13k():
    return reference_fixture()


def overflow_fixture(n=167772160):
    """The reference's only N >> 8M case (src/test.py:307-326, test_kmeanspp_lloyd_uint32_overflow): the 13000x2
    fixture stacked to 8 features and tiled to 167 772 160 rows -- 5.4 GB, past 2^32 BYTES."""
    base = reference_fixture()
    samples = numpy.empty((n, 8), dtype=numpy.float32)
    tile = numpy.hstack((base,) * 4)
    for i in range(0, n, base.shape[0]):
        end = i + base.shape[0]
        if end < n:
            samples[i:end] = tile
        else:
            samples[i:] = tile[:n - i]
    return samples
