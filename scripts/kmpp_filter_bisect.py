#!/usr/bin/env python
"""Filtered against plain k-means++ steps (seeding.hip) on the GPU: the seeds must be the same.  Sizes and data kinds
around the reference's uint32-overflow fixture (167 772 160 x 8), where they were not (round 4)."""
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import overflow_fixture
from kmcuda_amd import kmeans_cuda


def seeds(x, k, env):
    for a, b in env.items():
        os.environ[a] = b
    try:
        c, _ = kmeans_cuda(x, k, init="kmeans++", device=1, verbosity=0, seed=3, tolerance=1.0, yinyang_t=0)
    finally:
        for a in env:
            del os.environ[a]
    return c


def compare(name, x, k=50):
    t = time.time()
    a = seeds(x, k, {"KMCUDA_AMD_KMPP_FILTER": "0"})
    b = seeds(x, k, {})
    same = (a.view(numpy.uint32) == b.view(numpy.uint32)).all(axis=1)
    print("%-40s %9d x %d: %d of %d seeds equal (first different: %s)  %.1f s" %
          (name, x.shape[0], x.shape[1], int(same.sum()), k, None if same.all() else int(numpy.argmin(same)), time.time() - t), flush=True)


cases = sys.argv[1:] or ["tiled", "uniform", "tiled256"]
rs = numpy.random.RandomState(5)
if "tiled" in cases:
    for n in (1300000, 13000000, 33000000, 67000000, 134000000, 167772160):
        compare("tiled fixture", overflow_fixture(n))
if "uniform" in cases:
    for n in (13000000, 67000000, 167772160):
        compare("uniform", rs.rand(n, 8).astype(numpy.float32))
if "tiled256" in cases:
    base = overflow_fixture(13000)
    x = numpy.tile(numpy.hstack((base,) * 8), (200, 1))     # 2.6M x 64, every row 200 times
    compare("tiled, 64 features", x)
