"""Model check (CPU, numpy) of the carried bounds (kmcuda_amd/csrc/lloyd_carry.hip, lloyd_coarse.hpp CARRY != 0,
centroid_prep_frozen_kernel's drift): the float32 formulas restated exactly as the kernels form them, against
float64 ground truth.

Claim: a row the skip test keeps has, in exact arithmetic, d(x, c)^2 - d(x, a)^2 > 4 E_ref for EVERY other centroid c
of the NEW centroid set, E_ref being the bound on the reference's own rounding -- so the reference's
scan (kmeans.cu:293-364), whose computed score is within 2 E_ref (in squared-distance units) of the exact one, keeps
a, strictly.  The coarse scores the bounds are read from are modelled adversarially: any value within +-e_c of the
exact score, including the extremes that make the upper bound as small and the lower bound as large as e_c allows.
(The GPU tests compare whole loops with and without the bounds; this pins the inequality.)"""
import numpy
import pytest

F = numpy.float32
U = F(5.9604645e-8)


def _bounds_from_scores(xn2, v1, v2, e_c, eps, xn, cmaxc):
    """lloyd_coarse.hpp, finish(), CARRY != 0 (a row the stage decided)."""
    e = (e_c * F(1.001)).astype(F)
    d2u = numpy.maximum(xn2 * (F(1.0) + F(2.0) * eps) - F(2.0) * (v1 - e), F(0.0)).astype(F)
    d2l = (xn2 * (F(1.0) - F(2.0) * eps) - F(2.0) * (v2 + e)).astype(F)
    geo = (F(2.4e-7) * (xn + cmaxc)).astype(F)
    ub = (numpy.sqrt(d2u).astype(F) * F(1.000001) + geo).astype(F)
    lb = numpy.where(d2l > 0, numpy.maximum(numpy.sqrt(numpy.maximum(d2l, 0)).astype(F) * F(0.999999) - geo, F(0.0)), F(0.0)).astype(F)
    return ub, lb


def _drift(cnew_c, cold_c):
    """centroid_prep_frozen_kernel: from the centred fp32 panels of the two passes."""
    d = (cnew_c - cold_c).astype(F)
    dr2 = (d * d).sum(axis=1, dtype=F)
    n2 = (cnew_c * cnew_c).sum(axis=1, dtype=F)
    o2 = (cold_c * cold_c).sum(axis=1, dtype=F)
    return (numpy.sqrt(dr2).astype(F) * F(1.0002) + F(2.4e-7) * (numpy.sqrt(n2) + numpy.sqrt(o2)).astype(F) + F(1e-37)).astype(F)


def _skip(ub, lb, drift_a, maxdrift, xn2, mu_norm, cmaxo, tie_slack=F(0.0)):
    """carry_skip_kernel."""
    un = ((ub + drift_a) * F(1.0000005)).astype(F)
    ln = ((lb - maxdrift) * F(0.9999995)).astype(F)
    xo = ((numpy.sqrt(xn2).astype(F) * F(1.0001) + mu_norm) * F(1.0001)).astype(F)
    e_ref = (U * (F(12.0) * xo * cmaxo + F(4.0) * cmaxo * cmaxo)).astype(F)
    keep = (ln > un) & (((ln - un) * (ln + un)).astype(F) > F(4.1) * e_ref + F(2.0) * tie_slack)
    return keep, e_ref


@pytest.mark.parametrize("case", ["blobs", "uniform", "offset", "tight", "tiny-drift", "big-drift"])
def test_a_spared_row_keeps_its_centroid_in_the_references_arithmetic(case):
    rs = numpy.random.RandomState(len(case) + 17)
    n, d, k = 4000, 48, 40
    if case in ("blobs", "tiny-drift", "big-drift"):
        cen = rs.rand(k // 2, d) * 6
        x = cen[rs.randint(0, k // 2, n)] + rs.randn(n, d)
    elif case == "uniform":
        x = rs.rand(n, d)
    elif case == "offset":
        x = rs.rand(n, d) * 3 + 200.0
    else:   # near-duplicate centroids: gaps of the order of the roundings
        base = rs.rand(8, d) * 4
        x = base[rs.randint(0, 8, n)] + 0.05 * rs.randn(n, d)
    x = x.astype(F)
    c_old = x[rs.choice(n, k, replace=False)].astype(F) + (F(1e-3) * rs.randn(k, d)).astype(F)
    if case == "tight":
        c_old[1] = c_old[0] + F(1e-5)
    scale = {"tiny-drift": 1e-6, "big-drift": 0.5}.get(case, 0.02)
    c_new = (c_old + (scale * rs.randn(k, d)).astype(F)).astype(F)
    mu = c_old.mean(axis=0, dtype=numpy.float64).astype(F)            # frozen mean
    eps = F(1.02 * (d + 12.0) * 5.9604644775390625e-8)
    x64, co64, cn64 = x.astype(numpy.float64), c_old.astype(numpy.float64), c_new.astype(numpy.float64)
    # --- the pass over the OLD centroids: exact scores of the centred operands, then +- e_c ---
    xc = (x - mu[None, :]).astype(F)
    cc_old = (c_old - mu[None, :]).astype(F)
    cc_new = (c_new - mu[None, :]).astype(F)
    xn2 = (xc * xc).sum(axis=1, dtype=F)
    xn = (numpy.sqrt(xn2).astype(F) * F(1.0001)).astype(F)
    cmaxc = F(numpy.sqrt((cc_old.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    bmaxc = F(0.5 * (cc_old.astype(numpy.float64) ** 2).sum(axis=1).max())
    d2_old = ((x64[:, None, :] - co64[None, :, :]) ** 2).sum(axis=2)
    xm2 = ((x64 - mu.astype(numpy.float64)) ** 2).sum(axis=1)
    s_exact = 0.5 * (xm2[:, None] - d2_old)                            # s(c) = (||x - mu||^2 - d^2) / 2
    order = numpy.argsort(-s_exact, axis=1)
    a = order[:, 0]
    s1 = s_exact[numpy.arange(n), a]
    s2 = s_exact[numpy.arange(n), order[:, 1]]
    # a coarse bound of realistic size (hi.hi products: ~2^-11 relative operand rounding) -- any e_c works as long as
    # |v - s| <= e_c, which is what the coarse stage guarantees
    e_c = (F(2.0) * eps * (xn * cmaxc + bmaxc) + F(2.0 ** -10) * xn * cmaxc).astype(F)
    mu_norm = F(numpy.sqrt((mu.astype(numpy.float64) ** 2).sum()) * 1.00001)
    cmaxo = F(numpy.sqrt((cn64 ** 2).sum(axis=1).max()) * 1.000001)
    drift = _drift(cc_new, cc_old)
    # the true drifts never exceed the modelled ones
    true_drift = numpy.sqrt(((cn64 - co64) ** 2).sum(axis=1))
    assert (true_drift <= drift.astype(numpy.float64)).all()
    maxdrift = drift.max()
    d2_new = ((x64[:, None, :] - cn64[None, :, :]) ** 2).sum(axis=2)
    kept_total = 0
    for sign1, sign2 in ((+1, -1), (-1, +1), (0, 0), (+1, +1), (-1, -1)):
        v1 = (s1 + sign1 * e_c.astype(numpy.float64) * 0.999).astype(F)
        v2 = (s2 + sign2 * e_c.astype(numpy.float64) * 0.999).astype(F)
        ub, lb = _bounds_from_scores(xn2, v1, v2, e_c, eps, xn, cmaxc)
        # the bounds hold for the OLD centroids
        d_a = numpy.sqrt(d2_old[numpy.arange(n), a])
        d_o = numpy.sqrt(numpy.partition(d2_old, 1, axis=1)[:, 1])
        assert (ub.astype(numpy.float64) >= d_a).all()
        assert (lb.astype(numpy.float64) <= d_o).all()
        keep, e_ref = _skip(ub, lb, drift[a], maxdrift, xn2, mu_norm, cmaxo)
        kept_total += int(keep.sum())
        # ground truth over the NEW centroids
        da2 = d2_new[numpy.arange(n), a]
        others = d2_new.copy()
        others[numpy.arange(n), a] = numpy.inf
        gap = others.min(axis=1) - da2
        assert (gap[keep] > 4.0 * e_ref[keep].astype(numpy.float64)).all()
    if case in ("blobs", "tiny-drift"):
        assert kept_total > n          # the test is not vacuous: most rows are spared in most variants
    if case == "big-drift":
        assert kept_total < 5 * n


@pytest.mark.parametrize("case", ["unit-blobs", "unit-uniform", "not-unit"])
def test_a_spared_row_keeps_its_centroid_under_the_angular_metric(case):
    """Angular metric: the certified SCORE gap s(a) - s(c), s(c) = x'.c' + mu.c', shrunk per pass by
    ||x'|| (drift(a) + max drift) + max_c db(c) - db(a), db the change of mu.c' (carry_skip_kernel, angular branch).  A kept row must have x.c_a - x.c_c > 4 E_ref + 2 tie for every other
    NEW centroid in exact arithmetic -- for rows and centroids of any norm."""
    rs = numpy.random.RandomState(len(case) + 5)
    n, d, k = 4000, 48, 40
    if case == "unit-blobs":
        cen = rs.randn(k, d)
        x = cen[rs.randint(0, k, n)] + 0.1 * rs.randn(n, d)
    else:
        x = rs.rand(n, d)
    x = x / numpy.linalg.norm(x, axis=1, keepdims=True)
    if case == "not-unit":
        x = x * (0.5 + rs.rand(n, 1))
    x = x.astype(F)
    c_old = x[rs.choice(n, k, replace=False)].astype(F)
    c_new = (c_old + (0.003 * rs.randn(k, d)).astype(F)).astype(F)
    c_new = (c_new / numpy.linalg.norm(c_new.astype(numpy.float64), axis=1, keepdims=True)).astype(F)
    mu = c_old.mean(axis=0, dtype=numpy.float64).astype(F)
    eps = F(1.02 * (d + 12.0) * 5.9604644775390625e-8)
    tie = F(1e-6)
    x64, co64, cn64, mu64 = (a.astype(numpy.float64) for a in (x, c_old, c_new, mu))
    xc = (x - mu[None, :]).astype(F)
    cc_old = (c_old - mu[None, :]).astype(F)
    cc_new = (c_new - mu[None, :]).astype(F)
    xn2 = (xc * xc).sum(axis=1, dtype=F)
    xn = (numpy.sqrt(xn2).astype(F) * F(1.0001)).astype(F)
    cmaxc = F(numpy.sqrt((cc_old.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    bmaxc = F(numpy.abs(cc_old.astype(numpy.float64) @ mu64).max())
    s_old = x64 @ (co64 - mu64).T                       # exact scores of the old centroids
    order = numpy.argsort(-s_old, axis=1)
    a = order[:, 0]
    s1 = s_old[numpy.arange(n), a]
    s2 = s_old[numpy.arange(n), order[:, 1]]
    e_c = (F(2.0) * eps * (xn * cmaxc + bmaxc) + F(2.0 ** -10) * xn * cmaxc).astype(F)
    mu_norm = F(numpy.sqrt((mu64 ** 2).sum()) * 1.00001)
    cmaxo = F(numpy.sqrt((cn64 ** 2).sum(axis=1).max()) * 1.000001)
    drift = _drift(cc_new, cc_old)
    maxdrift = drift.max()
    xo = ((numpy.sqrt(xn2).astype(F) * F(1.0001) + mu_norm) * F(1.0001)).astype(F)
    assert (xo.astype(numpy.float64) >= numpy.linalg.norm(x64, axis=1)).all()
    e_ref = (U * (F(12.0) * xo * cmaxo + F(4.0) * cmaxo * cmaxo)).astype(F)
    p_new = x64 @ cn64.T                                # what the reference compares (up to its rounding)
    # the second term of the score, mu.c', as the preparation kernel sums it (fp32), and its change per centroid
    b_old = (cc_old * mu[None, :]).sum(axis=1, dtype=F)
    b_new = (cc_new * mu[None, :]).sum(axis=1, dtype=F)
    db = (b_new - b_old).astype(F)
    maxdb = F(max(0.0, float(db.max()) * 1.000001))
    cmaxc_new = F(numpy.sqrt((cc_new.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    eb = F(4.0) * F(520.0) * U * mu_norm * (cmaxc_new + maxdrift)
    kept_total = 0
    for sign1, sign2 in ((+1, -1), (-1, +1), (0, 0)):
        v1 = (s1 + sign1 * e_c.astype(numpy.float64) * 0.999).astype(F)
        v2 = (s2 + sign2 * e_c.astype(numpy.float64) * 0.999).astype(F)
        e = (e_c * F(1.001)).astype(F)
        gap = (((v1 - e) - (v2 + e)) * F(0.999999)).astype(F)
        g = (gap - (xn * (drift[a] + maxdrift) + (maxdb - db[a]) + eb) * F(1.000001)).astype(F)
        keep = g > F(4.1) * e_ref + F(2.0) * tie
        kept_total += int(keep.sum())
        pa = p_new[numpy.arange(n), a]
        others = p_new.copy()
        others[numpy.arange(n), a] = -numpy.inf
        true_gap = pa - others.max(axis=1)
        assert (true_gap[keep] > 4.0 * e_ref[keep].astype(numpy.float64) + 2.0 * float(tie)).all()
    if case == "unit-blobs":
        assert kept_total > n


@pytest.mark.parametrize("case", ["shared-blobs", "shared-blobs-drift", "uniform", "three-close"])
def test_a_pair_certificate_leaves_only_the_two_contenders(case):
    """Stage 2's pair state (lloyd_refine.hpp, PAIRS) and carry_skip_kernel's pair test: fp32 contender scores within
    e_mfma of the exact ones, every other centroid's coarse score (within e_c of exact) at most `rest`; u bounds the
    distances to BOTH contenders, l3 every other centroid's.  A row whose certificate survives the drifts must have, for
    the NEW centroids and in exact arithmetic, d(x, c)^2 - d(x, p)^2 > 4 E_ref for p in {p1, p2} and every other c: the
    reference's scan ends on p1 or p2, and the pair kernel finds out which in the reference's own arithmetic."""
    rs = numpy.random.RandomState(len(case) + 3)
    n, d, k = 4000, 48, 40
    if case.startswith("shared-blobs"):
        cen = rs.rand(k // 2, d) * 8
        x = cen[rs.randint(0, k // 2, n)] + rs.randn(n, d)
        c_old = numpy.concatenate([cen + 0.4 * rs.randn(k // 2, d), cen + 0.4 * rs.randn(k // 2, d)])   # two per blob
    elif case == "three-close":
        cen = rs.rand(k // 4, d) * 8
        x = cen[rs.randint(0, k // 4, n)] + rs.randn(n, d)
        c_old = numpy.concatenate([cen + 0.3 * rs.randn(k // 4, d) for _ in range(4)])                    # four per blob
    else:
        x = rs.rand(n, d)
        c_old = x[rs.choice(n, k, replace=False)]
    x = x.astype(F)
    c_old = c_old.astype(F)
    scale = 0.2 if case == "shared-blobs-drift" else 0.01
    c_new = (c_old + (scale * rs.randn(k, d)).astype(F)).astype(F)
    mu = c_old.mean(axis=0, dtype=numpy.float64).astype(F)
    eps = F(1.02 * (d + 12.0) * 5.9604644775390625e-8)
    x64, co64, cn64, mu64 = (v.astype(numpy.float64) for v in (x, c_old, c_new, mu))
    xc = (x - mu[None, :]).astype(F)
    cc_old = (c_old - mu[None, :]).astype(F)
    cc_new = (c_new - mu[None, :]).astype(F)
    xn2 = (xc * xc).sum(axis=1, dtype=F)
    xn = (numpy.sqrt(xn2).astype(F) * F(1.0001)).astype(F)
    cmaxc = F(numpy.sqrt((cc_old.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    bmaxc = F(0.5 * (cc_old.astype(numpy.float64) ** 2).sum(axis=1).max())
    d2_old = ((x64[:, None, :] - co64[None, :, :]) ** 2).sum(axis=2)
    d2_new = ((x64[:, None, :] - cn64[None, :, :]) ** 2).sum(axis=2)
    xm2 = ((x64 - mu64) ** 2).sum(axis=1)
    s_exact = 0.5 * (xm2[:, None] - d2_old)
    order = numpy.argsort(-s_exact, axis=1)
    rows = numpy.arange(n)
    p1, p2 = order[:, 0], order[:, 1]
    s1, s2, s3 = s_exact[rows, p1], s_exact[rows, p2], s_exact[rows, order[:, 2]]
    e_mfma = (F(2.0) * eps * (xn * cmaxc + bmaxc)).astype(F)
    dcmax = F(2.0 ** -11) * cmaxc
    dxw = (F(4.8829e-4) * xn).astype(F)
    e_c = (F(2.0) * eps * (xn * cmaxc + bmaxc) + (xn * dcmax + dxw * cmaxc + dxw * dcmax) * F(1.001) +
           F(6e-8) * F(numpy.sqrt(64.0)) * (xn + cmaxc) + F(2.0e-6) * (F(1.001) * xn * cmaxc + bmaxc)).astype(F)
    geo = (F(2.4e-7) * (xn + cmaxc)).astype(F)
    mu_norm = F(numpy.sqrt((mu64 ** 2).sum()) * 1.00001)
    cmaxo = F(numpy.sqrt((cn64 ** 2).sum(axis=1).max()) * 1.000001)
    drift = _drift(cc_new, cc_old)
    maxdrift = drift.max()
    xo = ((numpy.sqrt(xn2).astype(F) * F(1.0001) + mu_norm) * F(1.0001)).astype(F)
    e_ref = (U * (F(12.0) * xo * cmaxo + F(4.0) * cmaxo * cmaxo)).astype(F)
    certified = 0
    # two contenders (the third centroid is the best of the rest), or three (v3 a contender's fp32 score); the scores at
    # the extremes of their error bands that make u small and l3 large
    for three in (False, True):
        e = (e_mfma * F(1.001)).astype(F)
        v2 = (s2 + 0.999 * e_mfma.astype(numpy.float64)).astype(F)             # high v2: small u
        if three:
            v3 = (s3 - 0.999 * e_mfma.astype(numpy.float64)).astype(F)         # low third contender
            rest = (s_exact[rows, order[:, 3]] - 0.999 * e_c.astype(numpy.float64)).astype(F)
        else:
            v3 = numpy.full(n, -numpy.inf, dtype=F)
            rest = (s3 - 0.999 * e_c.astype(numpy.float64)).astype(F)          # low coarse score of the best of the rest
        w = numpy.maximum(rest + e_c * F(1.001), v3 + e).astype(F)
        d2l = (xn2 * (F(1.0) - F(2.0) * eps) - F(2.0) * w).astype(F)
        l3 = numpy.where(d2l > 0, numpy.maximum(numpy.sqrt(numpy.maximum(d2l, 0)).astype(F) * F(0.999999) - geo, F(0.0)), F(0.0)).astype(F)
        ub = (numpy.sqrt(numpy.maximum(xn2 * (F(1.0) + F(2.0) * eps) - F(2.0) * (v2 - e), F(0.0))).astype(F) * F(1.000001) + geo).astype(F)
        # the statements hold for the OLD centroids
        d_pair = numpy.sqrt(numpy.maximum(d2_old[rows, p1], d2_old[rows, p2]))
        others = d2_old.copy()
        others[rows, p1] = numpy.inf
        others[rows, p2] = numpy.inf
        assert (ub.astype(numpy.float64) >= d_pair).all()
        assert (l3.astype(numpy.float64) <= numpy.sqrt(others.min(axis=1))).all()
        # carry_skip_kernel's pair test
        up = ((ub + numpy.maximum(drift[p1], drift[p2])) * F(1.0000005)).astype(F)
        lp = ((l3 - maxdrift) * F(0.9999995)).astype(F)
        ok = (l3 > 0) & (lp > up) & (((lp - up) * (lp + up)).astype(F) > F(4.1) * e_ref)
        certified += int(ok.sum())
        on = d2_new.copy()
        on[rows, p1] = numpy.inf
        on[rows, p2] = numpy.inf
        worst_pair = numpy.maximum(d2_new[rows, p1], d2_new[rows, p2])
        assert ((on.min(axis=1) - worst_pair)[ok] > 4.0 * e_ref[ok].astype(numpy.float64)).all()
    if case == "shared-blobs":
        assert certified > n        # not vacuous: nearly every row's third centroid is another blob's
    if case == "uniform":
        assert certified < n // 2


@pytest.mark.parametrize("case", ["unit-shared-blobs", "unit-uniform", "not-unit"])
def test_an_angular_pair_certificate_leaves_only_the_two_contenders(case):
    """The pair certificate in score space (lloyd_refine.hpp PAIRS with cy.angular, carry_skip_kernel's angular pair test):
    l3 = the gap by which both contenders' scores exceed every other centroid's, shrunk per pass by
    ||x'|| (max(drift(p1), drift(p2)) + max drift) + max_c db(c) - min(db(p1), db(p2)).  A row that keeps it has
    x.c_p - x.c_c > 4 E_ref + 2 tie for p in {p1, p2} and every other NEW centroid, in exact arithmetic."""
    rs = numpy.random.RandomState(len(case) + 11)
    n, d, k = 4000, 48, 40
    if case == "unit-shared-blobs":
        cen = rs.rand(k // 2, d) * 8
        x = cen[rs.randint(0, k // 2, n)] + rs.randn(n, d)
        c_old = numpy.concatenate([cen + 0.4 * rs.randn(k // 2, d), cen + 0.4 * rs.randn(k // 2, d)])
        c_old = c_old / numpy.linalg.norm(c_old, axis=1, keepdims=True)
    else:
        x = rs.rand(n, d)
        c_old = None
    x = x / numpy.linalg.norm(x, axis=1, keepdims=True)
    if case == "not-unit":
        x = x * (0.5 + rs.rand(n, 1))
    x = x.astype(F)
    if c_old is None:
        c_old = x[rs.choice(n, k, replace=False)]
    c_old = c_old.astype(F)
    c_new = (c_old + (0.002 * rs.randn(k, d)).astype(F)).astype(F)
    c_new = (c_new / numpy.linalg.norm(c_new.astype(numpy.float64), axis=1, keepdims=True)).astype(F)
    mu = c_old.mean(axis=0, dtype=numpy.float64).astype(F)
    eps = F(1.02 * (d + 12.0) * 5.9604644775390625e-8)
    tie = F(1e-6)
    x64, co64, cn64, mu64 = (v.astype(numpy.float64) for v in (x, c_old, c_new, mu))
    xc = (x - mu[None, :]).astype(F)
    cc_old = (c_old - mu[None, :]).astype(F)
    cc_new = (c_new - mu[None, :]).astype(F)
    xn2 = (xc * xc).sum(axis=1, dtype=F)
    xn = (numpy.sqrt(xn2).astype(F) * F(1.0001)).astype(F)
    cmaxc = F(numpy.sqrt((cc_old.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    bmaxc = F(numpy.abs(cc_old.astype(numpy.float64) @ mu64).max())
    s_old = x64 @ (co64 - mu64).T
    order = numpy.argsort(-s_old, axis=1)
    rows = numpy.arange(n)
    p1, p2 = order[:, 0], order[:, 1]
    s2, s3 = s_old[rows, p2], s_old[rows, order[:, 2]]
    e_mfma = (F(2.0) * eps * (xn * cmaxc + bmaxc)).astype(F)
    e_c = (e_mfma + F(2.0 ** -10) * xn * cmaxc).astype(F)
    mu_norm = F(numpy.sqrt((mu64 ** 2).sum()) * 1.00001)
    cmaxo = F(numpy.sqrt((cn64 ** 2).sum(axis=1).max()) * 1.000001)
    drift = _drift(cc_new, cc_old)
    maxdrift = drift.max()
    xo = ((numpy.sqrt(xn2).astype(F) * F(1.0001) + mu_norm) * F(1.0001)).astype(F)
    e_ref = (U * (F(12.0) * xo * cmaxo + F(4.0) * cmaxo * cmaxo)).astype(F)
    b_old = (cc_old * mu[None, :]).sum(axis=1, dtype=F)
    b_new = (cc_new * mu[None, :]).sum(axis=1, dtype=F)
    db = (b_new - b_old).astype(F)
    maxdb = F(max(0.0, float(db.max()) * 1.000001))
    cmaxc_new = F(numpy.sqrt((cc_new.astype(numpy.float64) ** 2).sum(axis=1).max()) * 1.000001)
    eb = F(4.0) * F(520.0) * U * mu_norm * (cmaxc_new + maxdrift)
    p_new = x64 @ cn64.T
    e = (e_mfma * F(1.001)).astype(F)
    v2 = (s2 + 0.999 * e_mfma.astype(numpy.float64)).astype(F)          # high: the largest gap the band allows
    rest = (s3 - 0.999 * e_c.astype(numpy.float64)).astype(F)           # low coarse score of the best of the rest
    w = (rest + e_c * F(1.001)).astype(F)
    pairg = (((v2 - e) - w) * F(0.999999)).astype(F)
    g2 = (pairg - (xn * (numpy.maximum(drift[p1], drift[p2]) + maxdrift) + (maxdb - numpy.minimum(db[p1], db[p2])) + eb) * F(1.000001)).astype(F)
    ok = (pairg > 0) & (g2 > F(4.1) * e_ref + F(2.0) * tie)
    others = p_new.copy()
    others[rows, p1] = -numpy.inf
    others[rows, p2] = -numpy.inf
    worst = numpy.minimum(p_new[rows, p1], p_new[rows, p2])
    assert ((worst - others.max(axis=1))[ok] > 4.0 * e_ref[ok].astype(numpy.float64) + 2.0 * float(tie)).all()
    if case == "unit-shared-blobs":
        assert int(ok.sum()) > n // 2
