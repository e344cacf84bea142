"""Model checks (CPU, numpy) of the two result-neutral filters round 3 added -- the decisions restated in float32
exactly as the kernels form them, checked against float64 ground truth on random, clustered and hostile data:

* kmcuda_amd/csrc/seeding.hip, kmpp_filter_kernel: a k-means++ step drops every row whose distance to the new seed
  provably is not below its distance so far, from a BYTE copy of the centred row (per-row scale a, measured residual
  bound rn) -- no row that the exact step would update may be dropped;
* kmcuda_amd/csrc/knn.hip, knn_centroid_bounds_kernel: lb[c][q] <= d(q, c) - R[c]; a cluster with lb > kth holds no
  row within kth of the query -- so no true neighbour may sit in a cluster the test rules out.
(The GPU parity tests compare whole runs with and without the filters; these pin the inequalities themselves.)"""
import numpy
import pytest

F = numpy.float32


def _quantise(xc):
    """kmpp_cache_kernel: q = rint(x' * (127 / max|x'|)) clamped, a = max / 127, rn = (||x' - a q|| + 1e-6 ||x'||) 1.001."""
    mx = numpy.abs(xc).max(axis=1).astype(F)
    a = (mx / F(127.0)).astype(F)
    inv = numpy.where(a > 0, F(1.0) / numpy.where(a > 0, a, F(1.0)), F(0.0)).astype(F)
    q = numpy.clip(numpy.rint(xc * inv[:, None]), -127, 127).astype(F)
    r = (xc - a[:, None] * q).astype(F)
    n2 = (xc.astype(F) ** 2).sum(axis=1, dtype=F)
    rn = ((numpy.sqrt((r ** 2).sum(axis=1, dtype=F)) + F(1e-6) * numpy.sqrt(n2)) * F(1.001)).astype(F)
    return q, a, rn, n2


def _dropped_l2(x, mu, seed, T, D):
    """The L2 decision of kmpp_filter_kernel for every row: True = dropped (keeps its distance)."""
    eps = F(1.02 * (D + 12.0) * 5.9604644775390625e-8)
    u = F(5.9604645e-8)
    xc = (x - mu[None, :]).astype(F)
    q, a, rn, n2 = _quantise(xc)
    sp = (seed - mu).astype(F)
    sn2 = (sp * sp).sum(dtype=F)
    qn = F(numpy.sqrt(sn2)) * F(1.0001)
    acc = (q * sp[None, :]).sum(axis=1, dtype=F)          # (any fp32 order is inside eps xn qn)
    xn = numpy.sqrt(n2).astype(F) * F(1.0001)
    e_q = rn * qn * F(1.001) + eps * xn * qn
    E = F(4.04) * (F(3.0) * eps + F(16.0) * u) * (sn2 + n2) + F(6e-8) * F(numpy.sqrt(D)) * (qn + xn) + F(2.0) * e_q
    score = a * acc - F(0.5) * n2
    T2 = (T * T * F(1.000001)).astype(F)
    amin = F(0.5) * (sn2 - T2 - E) - F(1e-6) * (sn2 + T2)
    with numpy.errstate(invalid="ignore"):
        return score < amin


@pytest.mark.parametrize("case", ["uniform", "blobs", "tight", "offset", "scaled"])
def test_kmeanspp_byte_filter_never_drops_a_row_the_step_would_update(case):
    rs = numpy.random.RandomState(len(case))
    n, d = 6000, 64
    if case == "uniform":
        x = rs.rand(n, d)
    elif case == "blobs":
        cen = rs.rand(12, d) * 8
        x = cen[rs.randint(0, 12, n)] + rs.randn(n, d)
    elif case == "tight":      # near duplicates: margins of the order of the rounding
        base = rs.rand(40, d)
        x = base[rs.randint(0, 40, n)] + 1e-4 * rs.randn(n, d)
    elif case == "offset":     # far from the origin: the centring matters
        x = rs.rand(n, d) + 300.0
    else:                      # features of very different scale: coarse bytes for the small ones
        x = rs.rand(n, d) * numpy.exp(rs.uniform(-6, 3, d))[None, :]
    x = x.astype(F)
    mu = x[:2000].mean(axis=0, dtype=numpy.float64).astype(F)
    x64 = x.astype(numpy.float64)
    seeds = [int(rs.randint(n))]
    T64 = numpy.sqrt(((x64 - x64[seeds[0]]) ** 2).sum(axis=1))
    dropped_total = 0
    for step in range(25):
        j = int(rs.randint(n))
        dnew = numpy.sqrt(((x64 - x64[j]) ** 2).sum(axis=1))
        # the distances the reference holds are float32 roundings of (almost) the true ones: both neighbours of the
        # float64 value are tried as the threshold
        for T in (T64.astype(F), numpy.nextafter(T64.astype(F), F(0)), numpy.nextafter(T64.astype(F), F(numpy.inf))):
            drop = _dropped_l2(x, mu, x[j], T, d)
            would_update = dnew < T64 * (1 - 1e-6)      # clearly closer in exact arithmetic
            assert not (drop & would_update).any()
        dropped_total += int(drop.sum())
        T64 = numpy.minimum(T64, dnew)
    if case in ("uniform", "blobs", "offset"):
        assert dropped_total > 0.5 * 25 * n      # and it does drop most rows where there is a margin


def _centroid_bounds(x, cen, R, D):
    """knn_centroid_bounds_kernel in float32: lb[c][q]."""
    sd = min(F(0.99998), F(1.0) - (F(D) + F(16.0)) * F(6.0e-8))
    d2 = ((x[None, :, :].astype(F) - cen[:, None, :].astype(F)) ** 2).sum(axis=2, dtype=F)
    return (numpy.sqrt(d2).astype(F) * sd - R.astype(F)[:, None] * F(1.00003)) * F(0.99997)


@pytest.mark.parametrize("case", ["blobs", "uniform", "duplicates", "offcentre"])
def test_knn_centroid_bound_never_rules_out_a_cluster_that_holds_a_neighbour(case):
    rs = numpy.random.RandomState(7 + len(case))
    n, d, K, k = 3000, 48, 40, 8
    if case == "blobs":
        cen0 = rs.rand(15, d) * 6
        x = cen0[rs.randint(0, 15, n)] + rs.randn(n, d)
    elif case == "uniform":
        x = rs.rand(n, d)
    elif case == "duplicates":
        base = rs.rand(300, d)
        x = base[rs.randint(0, 300, n)]
    else:
        x = rs.rand(n, d) * 3
    x = x.astype(F)
    x64 = x.astype(numpy.float64)
    # any centroids and any assignment are legal inputs of knn_cuda: a crude clustering, and for "offcentre"
    # centroids that are NOT the members' means
    cen = x[rs.choice(n, K, replace=False)].astype(numpy.float64)
    asg = ((x64[:, None, :] - cen[None, :, :]) ** 2).sum(axis=2).argmin(axis=1)
    if case == "offcentre":
        cen = cen + rs.randn(K, d) * 0.5
    cen = cen.astype(F)
    dmember = numpy.sqrt(((x64 - cen.astype(numpy.float64)[asg]) ** 2).sum(axis=1))
    R = numpy.array([dmember[asg == c].max() if (asg == c).any() else numpy.nan for c in range(K)])
    # the reference's radii are float32 evaluations: a relative 1e-6 either way is tried
    for Rf in (R.astype(F), (R * (1 - 1e-6)).astype(F), (R * (1 + 1e-6)).astype(F)):
        lb = _centroid_bounds(x, cen, Rf, d)                        # K x n
        for s in rs.choice(n, 120, replace=False):
            dist = numpy.sqrt(((x64 - x64[s]) ** 2).sum(axis=1))
            dist[s] = numpy.inf
            nb = numpy.argsort(dist, kind="stable")[:k]
            kth = dist[nb[-1]]
            ruled_out = lb[:, s].astype(numpy.float64) > kth         # against the FINAL kth: the strictest use
            ruled_out[asg[s]] = False                               # (the own cluster is never tested)
            # every row within kth (ties included) must sit in a cluster that is not ruled out
            within = numpy.nonzero(dist <= kth)[0]
            assert not ruled_out[asg[within]].any()
