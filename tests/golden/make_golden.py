#!/usr/bin/env python
"""Writes tests/golden/golden.npz: seeded inputs -> outputs of the hot path, bit for bit.

Where the vectors come from: the reference is CUDA-only and cannot run in the build container or
on the MI355X box, so these are outputs of oracle/kmcuda_oracle.c -- the CPU restatement that
tests/test_oracle_pins.py ties to the reference's own known-answer tests (src/test.py iteration
counts, scikit-learn agreement, exact k-NN equality).  Committing them pins the oracle against
silent drift (tests/test_golden_cpu.py) and gives the -m gpu suite a checker that does not need
the oracle at all (tests/test_gpu_golden.py compares the HIP path with these arrays directly).

Inputs are regenerated from numpy's legacy RandomState (bit-stable across numpy versions), only
outputs are stored.  Run from the repository root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bits(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float32).view(numpy.uint32)


# ---- the cases: name -> (input builder, parameters); shared with the tests -------------------
ASSIGN_CASES = {
    # name: (seed, n, d, k, metric, passes)
    "assign_l2_4000x64_k100": (101, 4000, 64, 100, "L2", 3),
    "assign_l2_3000x256_k257": (102, 3000, 256, 257, "L2", 2),
    "assign_l2_2500x300_k70": (103, 2500, 300, 70, "L2", 2),
    "assign_cos_3000x64_k40": (104, 3000, 64, 40, "cos", 2),
}


def assign_inputs(seed, n, d, k, metric, passes):
    rs = numpy.random.RandomState(seed)
    x = rs.rand(n, d).astype(numpy.float32)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1)[:, None].astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    cs = [c0]
    for p in range(1, passes):
        c = (cs[-1] + rs.randn(k, d).astype(numpy.float32) * numpy.float32(0.02)).astype(numpy.float32)
        if metric == "cos":
            c /= numpy.linalg.norm(c, axis=1)[:, None].astype(numpy.float32)
        cs.append(c)
    return x, cs


KMEANS_CASES = {
    # name: (data, clusters, kwargs) -- strict-parity update (KMCUDA_AMD_EXACT_UPDATE=1) on the HIP side
    "kmeans_13k_random_lloyd": ("fixture13k", 50, dict(init="random", seed=3, tolerance=0.05, yinyang_t=0)),
    "kmeans_13k_kmeanspp_lloyd": ("fixture13k", 50, dict(init="kmeans++", seed=3, tolerance=0.05, yinyang_t=0)),
    "kmeans_13k_kmeanspp_yinyang": ("fixture13k", 50, dict(init="kmeans++", seed=3, tolerance=0.01, yinyang_t=0.1)),
    "kmeans_6000x64_random_yinyang": ("rand6000x64", 60, dict(init="random", seed=7, tolerance=0.01, yinyang_t=0.1)),
    "kmeans_cos_3000x32_lloyd": ("unit3000x32", 25, dict(init="kmeans++", seed=5, tolerance=0.01, yinyang_t=0,
                                                          metric="cos")),
}


def kmeans_data(name):
    if name == "fixture13k":
        from conftest import reference_fixture
        return reference_fixture()
    if name == "rand6000x64":
        rs = numpy.random.RandomState(201)
        return numpy.concatenate([rs.randn(1000, 64) + 3 * rs.randn(1, 64) for _ in range(6)]).astype(numpy.float32)
    if name == "unit3000x32":
        rs = numpy.random.RandomState(202)
        x = rs.rand(3000, 32).astype(numpy.float32)
        return (x / numpy.linalg.norm(x, axis=1)[:, None]).astype(numpy.float32)
    raise KeyError(name)


KNN_CASES = {
    # name: (seed, n, d, clusters, k, metric)
    "knn_l2_2000x16_c20_k10": (301, 2000, 16, 20, 10, "L2"),
    "knn_l2_1500x64_c12_k5": (302, 1500, 64, 12, 5, "L2"),
}


def knn_inputs(seed, n, d, clusters, k, metric):
    rs = numpy.random.RandomState(seed)
    x = numpy.concatenate([rs.randn(n // 4, d) + 2 * rs.randn(1, d) for _ in range(4)]).astype(numpy.float32)
    x = x[rs.permutation(len(x))]
    init = x[rs.choice(len(x), clusters, replace=False)].copy()
    return x, init


def main():
    import oracle
    out = {}
    for name, case in ASSIGN_CASES.items():
        x, cs = assign_inputs(*case)
        asg = None
        for p, c in enumerate(cs):
            asg, prev, changed = oracle.lloyd_assign(x, c, assignments=asg, metric=case[4])
            out["%s/pass%d/assignments" % (name, p)] = asg.copy()
            out["%s/pass%d/changed" % (name, p)] = numpy.array([changed], numpy.uint32)
    for name, (data, clusters, kw) in KMEANS_CASES.items():
        x = kmeans_data(data)
        cen, asg, log = oracle.kmeans(x, clusters, **kw)
        out[name + "/centroid_bits"] = bits(cen)
        out[name + "/assignments"] = asg
        out[name + "/log"] = numpy.asarray(log, numpy.uint32)
    for name, case in KNN_CASES.items():
        x, init = knn_inputs(*case)
        cen, asg, _ = oracle.kmeans(x, case[3], init=init, tolerance=0.01, yinyang_t=0, metric=case[5])
        nb, _ = oracle.knn(case[4], x, cen, asg, metric=case[5])
        out[name + "/centroid_bits"] = bits(cen)
        out[name + "/assignments"] = asg
        out[name + "/neighbors"] = nb
    path = os.path.join(HERE, "golden.npz")
    numpy.savez_compressed(path, **out)
    print("wrote %s: %d arrays, %d bytes" % (path, len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
