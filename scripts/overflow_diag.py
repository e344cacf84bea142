#!/usr/bin/env python
"""Where the 167 772 160 x 8 run (the reference's uint32-overflow case, test.py:307-326) leaves the oracle's path:
seeds (k-means++ variants), first assignment pass, strict update, second pass -- each against the oracle on the same
inputs.  scratch/overflow_seeds.npy: the oracle's seeds (seed 3), computed once."""
import os
import sys
import time

import numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from conftest import overflow_fixture
from kmcuda_amd import kmeans_cuda

x = overflow_fixture()
n = len(x)
seeds_path = os.path.join(ROOT, "scratch", "overflow_seeds.npy")
if os.path.exists(seeds_path):
    oseeds = numpy.load(seeds_path)
else:
    t = time.time()
    oseeds = oracle.init_centroids(x, 50, init="kmeans++", seed=3)
    print("oracle seeds: %.1f s" % (time.time() - t), flush=True)


def seeds_of(env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        # tolerance 1: the first pass reassigns N <= 1.0 N rows: stop before any update -> centroids = seeds
        c, a = kmeans_cuda(x, 50, init="kmeans++", device=1, verbosity=0, seed=3, tolerance=1.0, yinyang_t=0)
    finally:
        for k in env:
            del os.environ[k]
    return c, a


for env in ({}, {"KMCUDA_AMD_KMPP_FILTER": "0"}, {"KMCUDA_AMD_KMPP_HOST": "1"}, {"KMCUDA_AMD_KMPP_FILTER": "0", "KMCUDA_AMD_KMPP_HOST": "1"}):
    t = time.time()
    c, a = seeds_of(env)
    same = (c.view(numpy.uint32) == oseeds.view(numpy.uint32)).all(axis=1)
    print("k-means++ %s: %d of 50 seeds equal the oracle's (first different: %s), %.1f s" %
          (env or "default", int(same.sum()), (int(numpy.argmin(same)) if not same.all() else None), time.time() - t), flush=True)

# from the oracle's seeds: pass 1, strict update, pass 2
t = time.time()
oa1, _, och1 = oracle.lloyd_assign(x, oseeds)
prev = numpy.full(n, 0xFFFFFFFF, numpy.uint32)
oc1, occ = oracle.adjust(x, prev, oa1, oseeds, numpy.zeros(50, numpy.uint32))
oa2, _, och2 = oracle.lloyd_assign(x, oc1, assignments=oa1.copy())
print("oracle from its seeds: pass 2 reassigns %d (%.1f s)" % (och2, time.time() - t), flush=True)
os.environ["KMCUDA_AMD_EXACT_UPDATE"] = "1"
c, a = kmeans_cuda(x, 50, init=oseeds, device=1, verbosity=1, seed=3, tolerance=1.0, yinyang_t=0)
print("GPU pass 1 from the oracle's seeds: %d assignments differ" % int((a != oa1).sum()), flush=True)
c, a = kmeans_cuda(x, 50, init=oseeds, device=1, verbosity=1, seed=3, tolerance=0.16, yinyang_t=0)
print("GPU strict update: centroids equal the oracle's: %s (max abs diff %.3g); pass 2: %d assignments differ" %
      (bool((c.view(numpy.uint32) == oc1.view(numpy.uint32)).all()), float(numpy.abs(c - oc1).max()), int((a != oa2).sum())), flush=True)
