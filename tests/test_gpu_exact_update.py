"""Strict-parity mode (KMCUDA_AMD_EXACT_UPDATE=1): the centroid update restates the reference's
kmeans_adjust (kmeans.cu:366-429) operation for operation, so centroids are BIT-IDENTICAL to the
oracle's, and with them every later assignment: whole kmeans_cuda() runs -- Lloyd and Yinyang --
reproduce the oracle bit for bit (centroids, assignments, per-iteration reassignment counts)."""
import numpy
import pytest

import oracle
from test_gpu_kmeans import StdoutListener

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _bits(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float32).view(numpy.uint32)


@pytest.mark.parametrize("n,d,k,metric", [(5000, 2, 50, "L2"), (3000, 7, 33, "L2"), (6000, 256, 100, "L2"),
                                          (2000, 600, 70, "L2"), (3000, 64, 40, "cos"), (1500, 9, 20, "cos")])
def test_adjust_exact_bit_identical(n, d, k, metric):
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(n + d + k)
    x = rs.rand(n, d).astype(numpy.float32)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1)[:, None]
    m = oracle.COS if metric == "cos" else oracle.L2
    c0 = x[rs.choice(n, k, replace=False)].copy()
    eng = Engine(n, d, k, metric, device=0)
    xs = torch.from_numpy(x).to(dev)
    cen = torch.from_numpy(c0.copy()).to(dev)
    ccounts = torch.zeros(k, dtype=torch.int32, device=dev)
    ref_c, ref_cc = c0, numpy.zeros(k, numpy.uint32)
    asg = None
    for it in range(3):
        a, p, _ = oracle.lloyd_assign(x, ref_c, assignments=asg, metric=m)
        ref_c, ref_cc = oracle.adjust(x, p, a, ref_c, ref_cc, metric=m)
        pt = torch.from_numpy(p.view(numpy.int32).copy()).to(dev)
        at = torch.from_numpy(a.view(numpy.int32).copy()).to(dev)
        eng.adjust_exact(xs, pt, at, cen, ccounts)
        eng.sync()
        got = cen.cpu().numpy()
        assert (ccounts.cpu().numpy().view(numpy.uint32) == ref_cc).all()
        assert (_bits(got) == _bits(ref_c)).all(), "iteration %d" % it
        asg = a
    eng.close()


def _run(x, k, monkeypatch, **kw):
    from kmcuda_amd import kmeans_cuda
    monkeypatch.setenv("KMCUDA_AMD_EXACT_UPDATE", "1")
    out = StdoutListener()
    with out:
        cen, asg = kmeans_cuda(x, k, device=1, verbosity=1, **kw)
    reass = [int(l.split(":")[1].split()[0]) for l in out.text.split("\n") if l.startswith("iteration")]
    return cen, asg, reass


@pytest.mark.parametrize("kw", [dict(init="random", seed=3, tolerance=0.05, yinyang_t=0),
                                dict(init="kmeans++", seed=3, tolerance=0.05, yinyang_t=0),
                                dict(init="kmeans++", seed=3, tolerance=0.01, yinyang_t=0.1)])
def test_end_to_end_bit_identical_fixture(fixture13k, monkeypatch, kw):
    cen, asg, reass = _run(fixture13k, 50, monkeypatch, **kw)
    ocen, oasg, olog = oracle.kmeans(fixture13k, 50, **kw)
    assert reass == list(olog)
    assert (asg == oasg).all()
    assert (_bits(cen) == _bits(ocen)).all()


@pytest.mark.parametrize("yy", [0, 0.1])
def test_end_to_end_bit_identical_256d(monkeypatch, yy):
    rs = numpy.random.RandomState(0)
    x = rs.rand(20000, 256).astype(numpy.float32)
    kw = dict(init="random", seed=777, tolerance=0.01, yinyang_t=yy)
    cen, asg, reass = _run(x, 64, monkeypatch, **kw)
    ocen, oasg, olog = oracle.kmeans(x, 64, **kw)
    assert reass == list(olog)
    assert (asg == oasg).all()
    assert (_bits(cen) == _bits(ocen)).all()


def test_end_to_end_bit_identical_cosine_yinyang(monkeypatch):
    numpy.random.seed(0)
    arr = numpy.random.rand(1000, 256).astype(numpy.float32)
    arr /= numpy.linalg.norm(arr, axis=1)[:, numpy.newaxis]
    kw = dict(init="kmeans++", metric="cos", yinyang_t=0.1, seed=3)
    cen, asg, reass = _run(arr, 10, monkeypatch, **kw)
    ocen, oasg, olog = oracle.kmeans(arr, 10, **kw)
    assert len(reass) == len(olog) == 9
    # angular: acosf differs between libm and ocml in the last ulp, so near-ties may flip
    assert (asg != oasg).mean() < 0.01
