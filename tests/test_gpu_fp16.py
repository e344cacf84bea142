"""fp16x2 boundary (reference: fp_abstraction.h:100-182, tests src/test.py:468-560, :643-651).

Semantics of this implementation: half buffers at the boundary, the fp32 reference
arithmetic applied to the half VALUES, centroids rounded to half (RN) after every update -- where
the reference accumulates in half2.  Hence two bars:
  * vs the oracle in its fp16-storage mode (same semantics): assignments identical, centroids
    identical halves (strict update) / within one half ulp (default fp64 update);
  * vs the reference's own fp16 pins: tolerance -- iteration counts within the spread fp16
    accumulation noise allows, sklearn agreement thresholds as in test.py."""
import numpy
import pytest

import oracle
from test_gpu_kmeans import StdoutListener, _validate

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_supports_fp16():
    from kmcuda_amd import supports_fp16
    assert supports_fp16


def test_fp16_random_lloyd(fixture13k):
    # test.py:470-485 (reference pin: 7 iterations, same as fp32)
    from kmcuda_amd import kmeans_cuda
    samples = fixture13k.astype(numpy.float16)
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(samples, 50, init="random", device=1, verbosity=2, seed=3,
                                             tolerance=0.05, yinyang_t=0)
    assert centroids.dtype == numpy.float16
    assert centroids.shape == (50, 2) and assignments.shape == (13000,)
    assert out.iterations() == 7
    _validate(fixture13k, centroids.astype(numpy.float32), assignments, 0.05)


def test_fp16_kmeanspp_lloyd(fixture13k):
    # test.py:487-497 (reference pin: 5)
    from kmcuda_amd import kmeans_cuda
    samples = fixture13k.astype(numpy.float16)
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(samples, 50, init="kmeans++", device=1, verbosity=2, seed=3,
                                             tolerance=0.05, yinyang_t=0)
    assert out.iterations() == 5
    _validate(fixture13k, centroids.astype(numpy.float32), assignments, 0.05)


def test_fp16_kmeanspp_validate(fixture13k):
    # test.py:511-521: same seeds as the fp32 run up to half quantisation
    from kmcuda_amd import kmeans_cuda
    c32, _ = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, seed=3, tolerance=1.0, yinyang_t=0)
    c16, _ = kmeans_cuda(fixture13k.astype(numpy.float16), 50, init="kmeans++", device=1, seed=3, tolerance=1.0,
                         yinyang_t=0)
    # half spacing at |x| < 4 is 2^-9 .. 2^-8: the same seed rows differ by at most half an ulp
    assert numpy.max(abs(c16[:4].astype(numpy.float32) - c32[:4])) < 2e-3


def test_fp16_kmeanspp_yinyang(fixture13k):
    # test.py:523-533: the reference's pin is 16 + 7 -- reproduced by the oracle's half2 restatement
    # (tests/test_oracle_pins.py::test_half2_iteration_pins).  The product's fp16 semantics are the fp32
    # arithmetic on the half values: 22 iterations, identical to the oracle's storage mode
    # (test_half2_vs_storage_semantics measures the deviation between the two semantics)
    from kmcuda_amd import kmeans_cuda
    samples = fixture13k.astype(numpy.float16)
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(samples, 50, init="kmeans++", device=1, verbosity=2, seed=3,
                                             tolerance=0.01, yinyang_t=0.1)
    assert out.iterations() == 22
    _validate(fixture13k, centroids.astype(numpy.float32), assignments, 0.0105)


def test_fp16_cosine_metric():
    # test.py:535-560
    from kmcuda_amd import kmeans_cuda
    from sklearn.metrics.pairwise import cosine_distances
    numpy.random.seed(0)
    arr = numpy.empty((10000, 2), dtype=numpy.float16)
    angs = numpy.random.rand(10000) * 2 * numpy.pi
    for i in range(10000):
        arr[i] = numpy.sin(angs[i]), numpy.cos(angs[i])
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(arr, 4, init="kmeans++", metric="cos", device=1, verbosity=2, seed=3)
    assert abs(out.iterations() - 5) <= 1
    assert len(centroids) == 4
    for c in centroids:
        assert 0.9995 < numpy.linalg.norm(c.astype(numpy.float32)) < 1.0005
    dists = numpy.round(cosine_distances(centroids.astype(numpy.float32))).astype(int)
    assert sorted(map(tuple, dists)) == sorted(map(tuple, [[0, 2, 1, 1], [2, 0, 1, 1], [1, 1, 0, 2], [1, 1, 2, 0]]))
    assert assignments.min() == 0 and assignments.max() == 3


@pytest.mark.parametrize("kw", [dict(init="random", seed=3, tolerance=0.05, yinyang_t=0),
                                dict(init="kmeans++", seed=3, tolerance=0.01, yinyang_t=0.1)])
def test_fp16_bit_identical_to_oracle_fp16_storage(fixture13k, monkeypatch, kw):
    """Strict update + fp16 storage: centroids (as halves), assignments and the per-iteration
    reassignment counts equal the oracle's fp16-storage run exactly."""
    from kmcuda_amd import kmeans_cuda
    monkeypatch.setenv("KMCUDA_AMD_EXACT_UPDATE", "1")
    samples = fixture13k.astype(numpy.float16)
    out = StdoutListener()
    with out:
        cen, asg = kmeans_cuda(samples, 50, device=1, verbosity=1, **kw)
    reass = [int(l.split(":")[1].split()[0]) for l in out.text.split("\n") if l.startswith("iteration")]
    ocen, oasg, olog = oracle.kmeans(samples, 50, **kw)
    assert reass == list(olog)
    assert (asg == oasg).all()
    assert (cen.view(numpy.uint16) == ocen.view(numpy.uint16)).all()


def test_fp16_256d_default_update_close_to_oracle():
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(0)
    x = rs.rand(20000, 256).astype(numpy.float16)
    cen, asg = kmeans_cuda(x, 64, init="random", seed=777, tolerance=0.02, yinyang_t=0, device=1)
    assert cen.dtype == numpy.float16 and cen.shape == (64, 256)
    nxt, _, _ = oracle.lloyd_assign(x.astype(numpy.float32), cen.astype(numpy.float32))
    assert (nxt != asg).mean() < 0.02


def test_fp16_import_and_device_ptrs(fixture13k):
    from kmcuda_amd import kmeans_cuda
    from kmcuda_amd.api import _DEVICE_ALLOCS, free_device_ptr
    samples = fixture13k.astype(numpy.float16)
    c0, _ = kmeans_cuda(samples, 50, init="random", device=1, seed=3, tolerance=0.25, yinyang_t=0)
    c1, a1 = kmeans_cuda(samples, 50, init=c0, device=1, seed=3, tolerance=0.05, yinyang_t=0)
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(samples).to(dev)
    cptr, aptr = kmeans_cuda((st.data_ptr(), 0, (13000, 1, True)), 50, init="random", device=1, seed=3,
                             tolerance=0.05, yinyang_t=0)
    cd = _DEVICE_ALLOCS[cptr].cpu().numpy()
    ad = _DEVICE_ALLOCS[aptr].cpu().numpy().view(numpy.uint32)
    ch, ah = kmeans_cuda(samples, 50, init="random", device=1, seed=3, tolerance=0.05, yinyang_t=0)
    assert cd.dtype == numpy.float16 and (cd.view(numpy.uint16) == ch.view(numpy.uint16)).all()
    assert (ad == ah).all()
    free_device_ptr(cptr)
    free_device_ptr(aptr)
    _validate(fixture13k, c1.astype(numpy.float32), a1, 0.05)


def test_fp16_knn(fixture13k):
    # test.py:643-651: < 500 mismatches vs sklearn on the half values; identical to the oracle's
    # fp32 search on the same (widened) values
    from sklearn.neighbors import NearestNeighbors
    from kmcuda_amd import kmeans_cuda, knn_cuda
    samples = fixture13k.astype(numpy.float16)
    cen, asg = kmeans_cuda(samples, 50, seed=777, device=1)
    nb = knn_cuda(10, samples, cen, asg, device=1)
    bn = NearestNeighbors(n_neighbors=10).fit(samples.astype(numpy.float32)).kneighbors()[1]
    assert (nb != bn).sum() < 500
    ref, _ = oracle.knn(10, samples.astype(numpy.float32), cen.astype(numpy.float32), asg)
    assert (nb == ref).all()


@pytest.mark.parametrize("n,d,k,metric", [(5000, 256, 1024, "L2"), (3000, 64, 100, "L2"), (2000, 16, 33, "L2"),
                                          (4096, 128, 257, "L2"), (3000, 256, 64, "cos")])
def test_f16_matrix_core_filter_bit_exact(n, d, k, metric):
    """The f16-MFMA Lloyd filter (rows read as halves, centred hi/lo-split operands) + the shared
    exact refine kernels: assignments bit-identical to the oracle on the widened values, and to the
    f32-MFMA filter on the same values."""
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(n + d + k)
    x16 = rs.rand(n, d).astype(numpy.float16)
    if metric == "cos":
        x32n = x16.astype(numpy.float32)
        x16 = (x32n / numpy.linalg.norm(x32n, axis=1)[:, None]).astype(numpy.float16)
    x32 = x16.astype(numpy.float32)
    c = x32[rs.choice(n, k, replace=False)].copy()
    c[3] = c[1]                       # duplicate centroid: exact tie
    m = oracle.COS if metric == "cos" else oracle.L2
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x32, c, metric=m)
    xs32, xs16, cs = torch.from_numpy(x32).to(dev), torch.from_numpy(x16).to(dev), torch.from_numpy(c).to(dev)

    def run(use_half):
        asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
        prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
        eng = Engine(n, d, k, metric, device=0)
        eng.set_half_rows(xs16 if use_half else None)
        eng.lloyd_assign(xs32, cs, asg, prev)
        counters = eng.counters()
        eng.close()
        return asg.cpu().numpy().view(numpy.uint32), counters

    got16, c16 = run(True)
    got32, c32 = run(False)
    if metric == "cos":   # (rows are centroids here: products at the clamp, tests/test_gpu_angular_clamp.py)
        from _angular import assert_only_acos_matters
        assert_only_acos_matters(x32, c, got16, ref, "half rows", max_fraction=1e-3)
        assert_only_acos_matters(x32, c, got32, ref, "fp32 rows", max_fraction=1e-3)
        return
    assert (got16 == ref).all() and (got32 == ref).all()
    assert c16[0] == ref_changed
    assert c16[1] + c16[3] < 0.2 * n     # the filter itself decides the bulk of the rows


# ---- KMCUDA_AMD_FP16_STRICT=1: the reference's half2 ARITHMETIC on the GPU (half2_strict.hip) ----
@pytest.mark.parametrize("kw,pin", [
    (dict(init="random", tolerance=0.05, yinyang_t=0), 7),           # test.py:468-485
    (dict(init="kmeans++", tolerance=0.05, yinyang_t=0), 5),         # test.py:487-497
    (dict(init="afkmc2", tolerance=0.05, yinyang_t=0), 4),           # test.py:499-509
    (dict(init="kmeans++", tolerance=0.01, yinyang_t=0.1), 16 + 7),  # test.py:523-533
])
def test_fp16_strict_reproduces_the_reference_pins(fixture13k, monkeypatch, kw, pin):
    """The reference's fp16 known answers through the boundary in strict mode -- including the 16 + 7 the
    default fp16 semantics miss -- and, beyond the iteration count, the WHOLE run equal to the oracle's
    half2 restatement: the per-iteration reassignment counts, every assignment, every centroid half."""
    from kmcuda_amd import kmeans_cuda
    monkeypatch.setenv("KMCUDA_AMD_FP16_STRICT", "1")
    samples = fixture13k.astype(numpy.float16)
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(samples, 50, device=1, verbosity=2, seed=3, **kw)
    assert out.iterations() == pin
    assert centroids.dtype == numpy.float16
    ref_c, ref_a, ref_log = oracle.kmeans(samples, 50, seed=3, half2=True, **kw)
    got_log = [int(l.split()[2]) for l in out.text.split("\n") if l.startswith("iteration")]
    assert got_log == [int(v) for v in ref_log]
    assert (assignments == ref_a).all()
    assert numpy.array_equal(centroids.view(numpy.uint16), ref_c.view(numpy.uint16))


def test_fp16_strict_cosine(monkeypatch):
    # test.py:535-560 in strict mode; acosf is libm on the oracle and ocml here: iteration pin + properties
    from kmcuda_amd import kmeans_cuda
    from sklearn.metrics.pairwise import cosine_distances
    monkeypatch.setenv("KMCUDA_AMD_FP16_STRICT", "1")
    numpy.random.seed(0)
    arr = numpy.empty((10000, 2), dtype=numpy.float16)
    angs = numpy.random.rand(10000) * 2 * numpy.pi
    for i in range(10000):
        arr[i] = numpy.sin(angs[i]), numpy.cos(angs[i])
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(arr, 4, init="kmeans++", metric="cos", device=1, verbosity=2, seed=3)
    assert out.iterations() == 5
    for c in centroids.astype(numpy.float32):
        assert 0.9995 < numpy.linalg.norm(c) < 1.0005
    d = numpy.round(cosine_distances(centroids.astype(numpy.float32))).astype(int)
    assert sorted(map(tuple, d.tolist())) == sorted([(0, 2, 1, 1), (2, 0, 1, 1), (1, 1, 0, 2), (1, 1, 2, 0)])


def test_fp16_strict_wider_rows_equal_the_half2_oracle(monkeypatch):
    """D = 64 halves (32 half2 pairs per row: the interleaved accumulators actually interleave), L2 and
    angular, Lloyd with average distance: assignments and centroids equal to the half2 oracle."""
    from kmcuda_amd import kmeans_cuda
    monkeypatch.setenv("KMCUDA_AMD_FP16_STRICT", "1")
    rs = numpy.random.RandomState(8)
    x = (rs.rand(4000, 64) * 2 - 0.5).astype(numpy.float16)
    c, a, avg = kmeans_cuda(x, 20, init="random", seed=5, tolerance=0.01, yinyang_t=0, device=1, average_distance=True)
    rc, ra, rlog, ravg = oracle.kmeans(x, 20, init="random", seed=5, tolerance=0.01, yinyang_t=0, half2=True,
                                       average_distance=True)
    assert (a == ra).all()
    assert numpy.array_equal(c.view(numpy.uint16), rc.view(numpy.uint16))
    assert abs(avg - ravg) < 1e-6 * max(1.0, abs(ravg))


@pytest.mark.parametrize("metric", ["L2", "cos"])
def test_fp16_strict_knn_equals_the_half2_oracle(monkeypatch, metric):
    """knn_cuda under KMCUDA_AMD_FP16_STRICT=1 (knn.cu:19-243 with F = half2: radii from 16-half2 partials,
    centroid distances from 24-half2 partials, distance_tt per candidate; half distances tie constantly, so the
    lists also pin the reference's visiting order and heap): neighbour lists equal to the half2 oracle's."""
    from kmcuda_amd import kmeans_cuda, knn_cuda
    rs = numpy.random.RandomState(11)
    x = rs.rand(3000, 64).astype(numpy.float32)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    x = x.astype(numpy.float16)
    c, a = kmeans_cuda(x, 24, init="random", seed=3, tolerance=0.02, yinyang_t=0, metric=metric, device=1)
    assert c.dtype == numpy.float16
    monkeypatch.setenv("KMCUDA_AMD_FP16_STRICT", "1")
    nb = knn_cuda(10, x, c, a, metric=metric, device=1)
    ref, calced = oracle.knn(10, x, c, a, metric=metric, half2=True)
    if metric == "L2":
        assert (nb == ref).all()
    else:
        # acosf is libm on the oracle and ocml here; a half-rounded angle differs only when the two land on
        # different sides of a half rounding boundary
        assert (nb == ref).mean() > 0.999
    # and the mode matters: the default fp16 semantics (fp32 arithmetic on the half values) give other lists
    monkeypatch.delenv("KMCUDA_AMD_FP16_STRICT")
    plain = knn_cuda(10, x, c, a, metric=metric, device=1)
    assert (plain != ref).any()
