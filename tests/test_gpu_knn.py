"""GPU parity tests of knn_cuda() through the drop-in boundary (reference: src/knn.cu,
tests modelled on src/test.py:579-745).  Bar for fp32 L2: neighbour indices BIT-EXACT vs the CPU
oracle (same heap evolution, same order), for the filtered (matrix-core) search and for the
unfiltered exact search, plus the reference's own pins: exact equality with scikit-learn for
k=10, <= 2 mismatches for k=50, the sortedness / no-closer-outsider property at 40000x48."""
import numpy
import pytest

import oracle
from test_gpu_kmeans import StdoutListener

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def clustered13k(fixture13k):
    c, a, _ = oracle.kmeans(fixture13k, 50, seed=777)
    return fixture13k, c, a


@pytest.mark.parametrize("k,dmax", [(10, 0), (50, 2)])
def test_small_equals_sklearn_and_oracle(clustered13k, k, dmax):
    # test.py:594-615
    from sklearn.neighbors import NearestNeighbors
    from kmcuda_amd import knn_cuda
    x, c, a = clustered13k
    out = StdoutListener()
    with out:
        nb = knn_cuda(k, x, c, a, verbosity=2, device=1)
    bn = NearestNeighbors(n_neighbors=k).fit(x).kneighbors()[1]
    assert (nb != bn).sum() <= dmax
    ref, calced = oracle.knn(k, x, c, a)
    assert (nb == ref).all()
    line = [l for l in out.text.split("\n") if l.startswith("calculated")][0]
    assert abs(float(line.split()[1]) - calced / (13000.0 * 13000.0)) < 1e-6   # knn.cu:529-530


def test_exact_search_matches_filtered(clustered13k, monkeypatch):
    """Three searches, one answer: coarse f16 matrix-core filter (default: hi.hi products), the f32
    matrix-core filter, no filter."""
    from kmcuda_amd import knn_cuda
    x, c, a = clustered13k
    nb = knn_cuda(10, x, c, a, device=1)
    monkeypatch.setenv("KMCUDA_AMD_FILTER", "f32")
    nb32 = knn_cuda(10, x, c, a, device=1)
    monkeypatch.delenv("KMCUDA_AMD_FILTER")
    monkeypatch.setenv("KMCUDA_AMD_KNN_EXACT", "1")
    nbe = knn_cuda(10, x, c, a, device=1)
    assert (nb == nbe).all() and (nb32 == nbe).all()


def test_virtual_shards(clustered13k, monkeypatch):
    from kmcuda_amd import knn_cuda
    x, c, a = clustered13k
    nb = knn_cuda(10, x, c, a, device=1)
    monkeypatch.setenv("KMCUDA_AMD_VIRTUAL_SHARDS", "3")
    nb3 = knn_cuda(10, x, c, a, device=1)
    assert (nb == nb3).all()


def test_hostptr(clustered13k):
    # test.py:617-641
    from kmcuda_amd import knn_cuda
    x, c, a = clustered13k
    sp = x.__array_interface__["data"][0]
    cp = c.__array_interface__["data"][0]
    ap = a.__array_interface__["data"][0]
    nb = knn_cuda(10, (sp, -1, x.shape), (cp, len(c)), ap, verbosity=0)
    ref, _ = oracle.knn(10, x, c, a)
    assert (nb == ref).all()
    with pytest.raises(ValueError):
        knn_cuda(10, ("bullshit", -1, x.shape), (cp, len(c)), ap)
    with pytest.raises(TypeError):
        knn_cuda(10, "bullshit", (cp, len(c)), ap)
    with pytest.raises(ValueError):
        knn_cuda(10, (sp, -1, x.shape), ("bullshit", len(c)), ap)
    with pytest.raises(ValueError):
        knn_cuda(10, (sp, -1, x.shape), "bullshit", ap)
    with pytest.raises(ValueError):
        knn_cuda(10, (sp, -1, x.shape), (cp, len(c)), "bullshit")


def test_device_ptr(clustered13k):
    # test.py:701-733 with torch standing in for cuda4py
    from kmcuda_amd import knn_cuda
    from kmcuda_amd.api import _DEVICE_ALLOCS, free_device_ptr
    x, c, a = clustered13k
    dev = torch.device("cuda", 0)
    xs, cs = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
    at = torch.from_numpy(a.view(numpy.int32)).to(dev)
    ptr = knn_cuda(10, (xs.data_ptr(), 0, x.shape), (cs.data_ptr(), len(c)), at.data_ptr(), device=1)
    nb = _DEVICE_ALLOCS[ptr].cpu().numpy().view(numpy.uint32)
    ref, _ = oracle.knn(10, x, c, a)
    assert (nb == ref).all()
    assert (xs.cpu().numpy() == x).all()
    free_device_ptr(ptr)


@pytest.mark.parametrize("n,d,K,k", [(8000, 48, 160, 10), (6000, 256, 64, 10), (5000, 7, 40, 3), (3000, 100, 20, 33),
                                     (2000, 300, 16, 5),            # 256 < D <= 512: the one-operand-set f16 filter
                                     (3000, 512, 24, 10), (2500, 400, 300, 4),
                                     (1500, 640, 12, 6), (2200, 768, 20, 10), (2000, 1024, 16, 10),   # 513..1024: one block per CU
                                     (900, 1100, 8, 5)])            # beyond 1024: the exact search
@pytest.mark.parametrize("filt", ["f16", "f32"])
def test_matches_oracle_bit_exact(n, d, K, k, filt, monkeypatch):
    from kmcuda_amd import knn_cuda
    monkeypatch.setenv("KMCUDA_AMD_FILTER", filt)
    rs = numpy.random.RandomState(n + d)
    x = rs.rand(n, d).astype(numpy.float32)
    q = n // 4
    x[:q] += 1.0
    x[q:2 * q] -= 1.0
    x[2 * q:3 * q, 0] += 2.0
    x[3 * q:, 0] -= 2.0
    c, a, _ = oracle.kmeans(x, K, seed=777, yinyang_t=0, tolerance=0.02)
    nb = knn_cuda(k, x, c, a, device=1)
    ref, _ = oracle.knn(k, x, c, a)
    assert (nb == ref).all()


def test_duplicates_and_ties():
    """Duplicate rows => exactly equal distances: the neighbour ORDER then depends on the visiting
    order and heap mechanics, which must match the reference's (README.md:95-98)."""
    from kmcuda_amd import knn_cuda
    rs = numpy.random.RandomState(3)
    base = rs.rand(600, 16).astype(numpy.float32)
    x = numpy.concatenate([base, base, base[:300]]).astype(numpy.float32)
    c, a, _ = oracle.kmeans(x, 12, seed=777, yinyang_t=0, tolerance=0.02)
    ref, _ = oracle.knn(8, x, c, a)
    for k in (8, 12):
        nb = knn_cuda(k, x, c, a, device=1)
        refk = ref if k == 8 else oracle.knn(k, x, c, a)[0]
        assert (nb == refk).all()


def test_large_property():
    # test.py:653-699: 40000x48, K=800
    from kmcuda_amd import kmeans_cuda, knn_cuda
    rs = numpy.random.RandomState(0)
    samples = rs.rand(40000, 48).astype(numpy.float32)
    samples[:10000] += 1.0
    samples[10000:20000] -= 1.0
    samples[20000:30000, 0] += 2.0
    samples[30000:, 0] -= 2.0
    cen, asg = kmeans_cuda(samples, 800, seed=777, device=1)
    nb = knn_cuda(10, samples, cen, asg, device=1)
    for i in range(0, 40000, 41):
        sn = nb[i]
        d = numpy.linalg.norm(samples[i] - samples[sn], axis=1)
        assert (d[:-1] - d[1:] <= 3e-7).all()
        members = set(sn)
        for r in rs.randint(0, 40000, 100):
            if r == i or r in members:
                continue
            assert d[-1] <= numpy.linalg.norm(samples[i] - samples[r])


def test_cosine():
    # test.py:735-745 (scaled): angular k-NN agrees with the oracle up to acos plateaus
    from kmcuda_amd import knn_cuda
    rs = numpy.random.RandomState(0)
    x = rs.rand(6000, 16).astype(numpy.float32)
    x /= numpy.linalg.norm(x, axis=1)[:, None]
    c, a, _ = oracle.kmeans(x, 30, seed=777, metric="cos", yinyang_t=0, tolerance=0.02)
    nb = knn_cuda(10, x, c, a, metric="cos", device=1)
    ref, _ = oracle.knn(10, x, c, a, metric="cos")
    assert (nb != ref).mean() < 0.02
