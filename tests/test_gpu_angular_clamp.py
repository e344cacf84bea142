"""The angular metric's clamp on the GPU, every filter: the reference's distance is `p >= 1 ? 0 : p <= -1 ? pi : acos(p)`
(metric_abstraction.h:171-177), so every centroid whose product with a row reaches 1 sits at distance 0 and the
ascending strict-`<` scan of kmeans_assign_lloyd (kmeans.cu:342-346) keeps the LOWEST INDEX among them -- not the largest
product; the same at -1 / pi.  tests/test_angular_tie_rule_cpu.py pins that answer on the oracle; this file holds every
device path to it through kmamd_lloyd_assign: the two-stage f16 filter (rows converted per pass / from the row cache,
fp32 rows / half rows), the f32 matrix-core filter, the LDS-streamed filter, the exact kernels.

Bar: BIT-EQUAL to the oracle on the constructed cases (the clamp involves no acos: products at or beyond 1 are distance
0 exactly).  On random unit rows the only differences allowed are rows whose two candidate distances are within 2 ulp
of each other in the oracle's own arithmetic and not both 0 (acosf is libm on the CPU, ocml on the GPU, CUDA's in the
reference: SURVEY 8c, parity-unpinned)."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

VARIANTS = ["f16", "f16-cached", "f32", "wide", "exact"]


def _dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda", 0)


def _passes(x, cs, variant, half, monkeypatch):
    """One engine, one assignment pass per centroid set of cs (the cached variant freezes its mean at the first)."""
    from kmcuda_amd.engine import Engine
    if variant == "wide":
        monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "1")   # the streamed filter at every width
    dev = _dev()
    n, d = x.shape
    k = cs[0].shape[0]
    xs = torch.from_numpy(x).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, "cos", device=0)
    eng.set_filter("f32" if variant in ("f32", "exact") else "f16")
    if half:
        eng.set_half_rows(torch.from_numpy(x.astype(numpy.float16)).to(dev))
    if variant == "f16-cached":
        eng.set_row_cache(True)
    out = []
    for c in cs:
        eng.reset_counters(0)
        eng.lloyd_assign(xs, torch.from_numpy(c).to(dev), asg, prev, exact=(variant == "exact"))
        changed = eng.counters()[0]
        out.append((asg.cpu().numpy().view(numpy.uint32).copy(), prev.cpu().numpy().view(numpy.uint32).copy(), changed))
    eng.close()
    return out


def _check(x, cs, variant, half, monkeypatch, exact_rows=None):
    """exact_rows: rows that must match bit for bit whatever acos does (None: all of them)."""
    got = _passes(x, cs, variant, half, monkeypatch)
    ref_asg = None
    for (asg, prev, changed), c in zip(got, cs):
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg, metric=oracle.COS)
        bad = numpy.nonzero(asg != ref)[0]
        if exact_rows is None:
            assert bad.size == 0, (variant, half, bad[:10], asg[bad[:10]], ref[bad[:10]])
            assert (prev == ref_prev).all() and changed == ref_changed
        else:
            assert not numpy.isin(bad, exact_rows).any(), (variant, half, bad[:10])
            for i in bad:   # an acos plateau / last-ulp matter, never a clamp tie
                dg = oracle.distance(x[i], c[asg[i]], metric=oracle.COS)
                dr = oracle.distance(x[i], c[ref[i]], metric=oracle.COS)
                assert not (dg == 0.0 and dr == 0.0), (variant, half, i, asg[i], ref[i])
                assert abs(dg - dr) <= 2 * numpy.spacing(numpy.float32(max(dg, dr))), (variant, half, i, dg, dr)
        ref_asg = ref
        # the next pass of the device starts from the oracle's assignments only if they agree; they do on exact cases
        if bad.size:
            return


def _h(a):
    """values a float16 holds, as fp32 (rows that exist in both precisions)."""
    return a.astype(numpy.float16).astype(numpy.float32)


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("variant", VARIANTS)
def test_products_at_or_beyond_one_tie_and_the_lowest_index_wins(variant, half, monkeypatch):
    """The CPU pin's rows (tests/test_angular_tie_rule_cpu.py), padded with ordinary unit rows so that the means and
    norms are those of a real pass."""
    rs = numpy.random.RandomState(5)
    d = 16
    x = rs.randn(600, d).astype(numpy.float32)
    x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    x = _h(x)
    x[:3] = 0
    x[0, 0] = 1.0009766
    x[1, 1] = 1.0
    x[2, 2] = 0.99902344
    c = _h(x[rs.choice(numpy.arange(3, 600), 20, replace=False)])
    c[:4] = 0
    c[0, 0] = 0.9995117          # product with row 0: 1.00049 -> distance 0
    c[1, 0] = 1.0009766          # product with row 0: 1.00195 -> distance 0 too: the LARGER product, the higher index
    c[2, 1] = 1.0
    c[3, 2] = 1.0
    c2 = c.copy()
    c2[[0, 1]] = c2[[1, 0]]
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    assert ref[0] == 0 and ref[1] == 2 and ref[2] == 3
    _check(x, [c, c2], variant, half, monkeypatch)


def _near_duplicate_case(seed, n, d, k, half_rows):
    """Unit rows as a float16 cast leaves them (norms off 1 by up to 1e-3), centroids = rows and near-duplicates of
    rows scaled a few half-ulps up or down, the pair members in random index order: products of 1.0003 and 1.00001 with
    two centroids are common, and the larger one is the higher index half of the time."""
    rs = numpy.random.RandomState(seed)
    base = rs.randn(k // 2, d).astype(numpy.float32)
    base /= numpy.linalg.norm(base, axis=1, keepdims=True)
    # blobs around the base directions, tight enough that many rows reach products >= 1 with their centroid
    x = base[rs.randint(0, k // 2, n)] + rs.randn(n, d).astype(numpy.float32) * (0.002 if d <= 16 else 0.001)
    x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    x = _h(x) if half_rows else x.astype(numpy.float32)
    scale = 1.0 + rs.choice([-2, -1, 0, 1, 2, 3], k // 2) * 2.0 ** -11
    twin = base * scale[:, None].astype(numpy.float32)
    c = numpy.concatenate([base, twin]).astype(numpy.float32)
    c = c[rs.permutation(k)]
    return x, (_h(c) if half_rows else c)


@pytest.mark.parametrize("d", [2, 12, 16, 32])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("variant", VARIANTS)
def test_near_duplicate_centroids_next_to_the_clamp(variant, half, d, monkeypatch):
    x, c = _near_duplicate_case(100 + d, 6000, d, 40, half_rows=True)   # half VALUES in both precisions
    # a second pass with the centroids nudged: the cached variant's mean is frozen at the first pass's
    rs = numpy.random.RandomState(d)
    c2 = _h(c * (1.0 + rs.choice([-1, 0, 1], c.shape[0])[:, None] * 2.0 ** -11)).astype(numpy.float32)
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    prods = x.astype(numpy.float64) @ c.astype(numpy.float64).T
    at_clamp = (prods >= 1.0).sum(axis=1)
    assert (at_clamp >= 2).sum() > 50            # the case really has ties at distance 0 ...
    tied = numpy.nonzero(at_clamp >= 2)[0]
    assert (ref[tied] != prods[tied].argmax(axis=1)).sum() > 10   # ... where the largest product is NOT the answer
    _check(x, [c, c2], variant, half, monkeypatch)


@pytest.mark.parametrize("variant", VARIANTS)
def test_fp32_rows_of_256_features(variant, monkeypatch):
    """Config C's width.  fp32 unit rows are off 1 by 1e-7: a product reaches 1 only against a (near-)duplicate."""
    x, c = _near_duplicate_case(7, 4000, 256, 64, half_rows=False)
    c[5] = x[17]
    c[40] = x[17] * numpy.float32(1.0 + 2.0 ** -20)
    _check(x, [c], variant, False, monkeypatch, exact_rows=numpy.array([17]))


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("variant", VARIANTS)
def test_every_product_at_or_below_minus_one(variant, half, monkeypatch):
    """All distances pi: the scan keeps the first centroid, whatever the products are."""
    rs = numpy.random.RandomState(9)
    d = 16
    e = numpy.zeros(d, numpy.float32)
    e[3] = 1.0
    x = _h(numpy.tile(-e * 1.0009766, (300, 1)) + 0.0)
    x[150:] = _h(rs.randn(150, d) / 4.0)          # ordinary rows beside them
    c = _h(numpy.stack([e * s for s in (1.0, 1.0009766, 1.0019531, 0.9995117 + 0.001)] * 3))
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    assert (ref[:150] == 0).all()
    _check(x, [c], variant, half, monkeypatch, exact_rows=numpy.arange(150))


@pytest.mark.parametrize("variant", VARIANTS)
def test_rows_that_are_not_unit_length(variant, monkeypatch):
    """The API asks for unit rows but only probes three of them (kmcuda.cc:195-220): rows of norm 2 put MANY
    centroids beyond the clamp, and the lowest index among all of them is the reference's answer."""
    rs = numpy.random.RandomState(21)
    d = 32
    x = rs.rand(2000, d).astype(numpy.float32) + 0.1
    x *= (2.0 / numpy.linalg.norm(x, axis=1, keepdims=True)).astype(numpy.float32)
    c = rs.rand(50, d).astype(numpy.float32) + 0.1
    c /= numpy.linalg.norm(c, axis=1, keepdims=True)
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    prods = x.astype(numpy.float64) @ c.astype(numpy.float64).T
    assert ((prods >= 1.0).sum(axis=1) >= 2).mean() > 0.9
    _check(x, [c], variant, False, monkeypatch)
