"""The CPU oracle against the committed golden vectors (tests/golden/golden.npz, written by
tests/golden/make_golden.py): any change of the oracle's arithmetic, of its seeding streams or of
its iteration logic shows up here as a bit difference.  The same vectors check the HIP path in
tests/test_gpu_golden.py."""
import os
import sys

import numpy
import pytest

import oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden as mg  # noqa: E402

GOLDEN = numpy.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz"))


def test_every_case_has_vectors():
    names = set(k.split("/")[0] for k in GOLDEN.files)
    assert names == set(mg.ASSIGN_CASES) | set(mg.KMEANS_CASES) | set(mg.KNN_CASES)


@pytest.mark.parametrize("name", sorted(mg.ASSIGN_CASES))
def test_assign_passes(name):
    case = mg.ASSIGN_CASES[name]
    x, cs = mg.assign_inputs(*case)
    asg = None
    for p, c in enumerate(cs):
        asg, prev, changed = oracle.lloyd_assign(x, c, assignments=asg, metric=case[4])
        assert (asg == GOLDEN["%s/pass%d/assignments" % (name, p)]).all()
        assert changed == int(GOLDEN["%s/pass%d/changed" % (name, p)][0])


@pytest.mark.parametrize("name", sorted(mg.KMEANS_CASES))
def test_kmeans_runs(name):
    data, clusters, kw = mg.KMEANS_CASES[name]
    cen, asg, log = oracle.kmeans(mg.kmeans_data(data), clusters, **kw)
    assert list(log) == list(GOLDEN[name + "/log"])
    assert (asg == GOLDEN[name + "/assignments"]).all()
    assert (mg.bits(cen) == GOLDEN[name + "/centroid_bits"]).all()


@pytest.mark.parametrize("name", sorted(mg.KNN_CASES))
def test_knn(name):
    case = mg.KNN_CASES[name]
    x, init = mg.knn_inputs(*case)
    cen = GOLDEN[name + "/centroid_bits"].view(numpy.float32)
    nb, _ = oracle.knn(case[4], x, cen, GOLDEN[name + "/assignments"], metric=case[5])
    assert (nb == GOLDEN[name + "/neighbors"]).all()
