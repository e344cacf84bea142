"""Model check of the hinted Yinyang local filter (kmcuda_amd/csrc/yinyang_hint.hip).

The kernels replace the reference's sequential scan of a row (kmeans.cu:584-672, restated in
oracle/kmcuda_oracle.c: kmo_yy_local_filter) by: group bounds <= S' folded up front, candidates taken
against S', the (b) test applied with the kernel's own second minimum, and three flags (F1, F2, F3) that
hand the row back to the plain kernel.  The claim (header of yinyang_hint.hip): whenever no flag is
raised, (min, second, nearest) are the reference's -- for ANY S' >= upper bound and without assuming
that the bounds are valid lower bounds.

This file states both procedures in plain Python over abstract inputs (exact distances, group bounds,
drifts -- no geometry, so bounds can be arbitrarily wrong and distances can tie) and checks the claim on
a few hundred thousand random rows, most of them adversarial: few distinct values, invalid bounds, S'
anywhere above the upper bound, candidate supersets.  float32 throughout, like the kernels."""
import numpy

F = numpy.float32
FLT_MAX = F(3.402823466e+38)
NONE = 0xFFFFFFFF


def reference_scan(d, groups, G, lbg, gdrift, cdrift, cluster, ub):
    """kmeans.cu:598-652 for one row; d[c] is what distance_t would return."""
    min_dist, second, nearest = ub, FLT_MAX, cluster
    for c in range(len(d)):
        if c == cluster:
            continue
        g = groups[c]
        if g >= G:
            continue
        lb = lbg[g]
        if lb >= ub:
            if lb < second:
                second = lb
            continue
        lb = F(lb + F(gdrift[g] - cdrift[c]))
        if second < lb:
            continue
        dist = d[c]
        if dist < min_dist:
            second = min_dist
            min_dist = dist
            nearest = c
        elif dist < second:
            second = dist
    return min_dist, second, nearest


def hinted_scan(d, groups, G, lbg, gdrift, cdrift, cluster, ub, hint, extra, rs, sweep):
    """yy_local_hint_kernel (sweep=True: the candidate threshold follows the live second minimum from flush
    to flush, queue of four) / yy_local_list_kernel (sweep=False: threshold S' throughout).  `extra`:
    probability of taking a centroid the threshold would drop (the matrix-core filter keeps a superset).
    Returns (flag, min, second, nearest); flag 0 = settled here."""
    K = len(d)
    members = [[c for c in range(K) if groups[c] == g] for g in range(G)]
    # low_bound_fold
    second = FLT_MAX
    for g in range(G):
        lb = lbg[g]
        if lb >= ub and lb <= hint and lb < second:
            if any(c != cluster for c in members[g]):
                second = lb
    min_dist, nearest = ub, cluster
    flag = 0
    queue = []
    thr = min(second, hint)

    def flush():
        nonlocal second, min_dist, nearest, flag, thr
        for c in queue:
            g = groups[c]
            lb = F(lbg[g] + F(gdrift[g] - cdrift[c]))
            if not (second < lb):
                if lb > hint and not flag:
                    flag = 3                                  # F1
                dist = d[c]
                if dist < min_dist:
                    second = min_dist
                    min_dist = dist
                    nearest = c
                elif dist < second:
                    second = dist
            elif d[c] < lb and not flag:
                flag = 2                                      # F3
        queue.clear()
        if sweep:
            thr = min(second, hint)

    for c in range(K):
        if c == cluster or groups[c] >= G:
            continue
        if lbg[groups[c]] >= ub:                              # an (a) centroid: folded above or irrelevant
            continue
        if d[c] <= thr or rs.rand() < extra:                  # NaN distance: never a candidate by threshold
            if len(queue) == 4:
                flush()
            queue.append(c)
    flush()
    if not (second <= hint) and not flag:
        flag = 4                                              # F2
    return flag, min_dist, second, nearest


def _row(rs, style):
    K = int(rs.randint(3, 14))
    G = int(rs.randint(1, 5))
    groups = rs.randint(0, G + (1 if rs.rand() < 0.2 else 0), K)      # now and then a groupless centroid (>= G)
    groups[rs.randint(K)] = rs.randint(G)                             # ... but never all of them
    if style == "discrete":          # few distinct values: ties everywhere
        vals = numpy.array([1.0, 1.5, 2.0, 2.5, 3.0, 4.0], numpy.float32)
        d = rs.choice(vals, K)
        lbg = rs.choice(numpy.concatenate([vals, vals - F(0.25), [F(0.0)]]), G).astype(numpy.float32)
        gdrift = rs.choice([0.0, 0.25, 0.5], G).astype(numpy.float32)
        cdrift = rs.choice([0.0, 0.25, 0.5], K).astype(numpy.float32)
        ub = F(rs.choice(vals))
    else:                            # continuous, bounds mostly (not always) valid
        d = (rs.rand(K) * 3 + 1).astype(numpy.float32)
        cdrift = (rs.rand(K) * 0.2 * (rs.rand(K) < 0.6)).astype(numpy.float32)
        gdrift = numpy.array([max([cdrift[c] for c in range(K) if groups[c] == g] + [0.0]) for g in range(G)], numpy.float32)
        lbg = numpy.empty(G, numpy.float32)
        for g in range(G):
            m = [d[c] for c in range(K) if groups[c] == g]
            base = min(m) if m else 5.0
            slack = rs.rand() * 0.8 if rs.rand() < 0.8 else -rs.rand() * 0.3      # negative slack: an invalid bound
            lbg[g] = F(base - slack - gdrift[g])
        ub = F(rs.rand() * 3 + 1)
    if rs.rand() < 0.05:
        d[rs.randint(K)] = numpy.nan
    cluster = int(rs.randint(K))
    while groups[cluster] >= G:      # the row's own centroid always has a group (kmeans.cu:653)
        cluster = int(rs.randint(K))
    return d, groups, G, lbg, gdrift, cdrift, cluster, ub


def _hint(rs, d, ub, style):
    r = rs.rand()
    if r < 0.25:
        return ub
    if r < 0.5:
        fin = d[numpy.isfinite(d)]
        second_best = numpy.sort(fin)[min(1, len(fin) - 1)] if len(fin) else ub
        return max(ub, F(second_best + (0.0 if style == "discrete" else 1e-3)))
    if r < 0.9:
        return max(ub, F(ub + rs.rand() * 2))
    return F(100.0)


def test_unflagged_rows_equal_the_reference_scan():
    rs = numpy.random.RandomState(12345)
    settled = flagged = 0
    for trial in range(150000):
        style = "discrete" if trial % 2 else "continuous"
        d, groups, G, lbg, gdrift, cdrift, cluster, ub = _row(rs, style)
        hint = _hint(rs, d, ub, style)
        assert hint >= ub
        ref = reference_scan(d, groups, G, lbg, gdrift, cdrift, cluster, ub)
        for sweep in (False, True):
            extra = 0.0 if rs.rand() < 0.5 else 0.3
            flag, m, s2, n = hinted_scan(d, groups, G, lbg, gdrift, cdrift, cluster, ub, hint, extra, rs, sweep)
            if flag:
                flagged += 1
                continue
            settled += 1
            assert (m, s2, n) == ref, (trial, sweep, hint, ub, list(d), list(groups), list(lbg), list(gdrift),
                                       list(cdrift), cluster, (m, s2, n), ref)
    # the generator must exercise both outcomes heavily, or the check above says little
    assert settled > 80000 and flagged > 20000, (settled, flagged)
