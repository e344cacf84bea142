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


def test_four_smallest_record_settles_the_fold_or_asks_for_the_walk():
    """low_bound_fold with the global filter's record (KMCUDA_AMD_YY_REC=1): the four smallest group bounds
    either determine the folded value or the function walks all G bounds; never a different value."""
    rs = numpy.random.RandomState(777)
    decided = walked = 0
    vals = numpy.array([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, numpy.nan], numpy.float32)
    for trial in range(60000):
        G = int(rs.randint(1, 9))
        lbg = rs.choice(vals, G) if trial % 2 else (rs.rand(G) * 3).astype(numpy.float32)
        has_member = rs.rand(G) < 0.85          # a group whose only member is the row's own centroid has none
        ub = F(rs.choice(vals[:-1])) if trial % 2 else F(rs.rand() * 3)
        hint = max(ub, F(ub + rs.rand() * 1.5)) if rs.rand() < 0.8 else ub
        want = FLT_MAX                           # the walk
        for g in range(G):
            if lbg[g] >= ub and lbg[g] <= hint and lbg[g] < want and has_member[g]:
                want = lbg[g]
        # the record, as yy_global_filter_kernel<REC> builds it (a NaN bound is never noted)
        l = [F(numpy.inf)] * 4
        gi = [NONE] * 4
        for g in range(G):
            v = lbg[g]
            c = [v < l[0], v < l[1], v < l[2], v < l[3]]
            l[3], gi[3] = (l[2], gi[2]) if c[2] else ((v, g) if c[3] else (l[3], gi[3]))
            l[2], gi[2] = (l[1], gi[1]) if c[1] else ((v, g) if c[2] else (l[2], gi[2]))
            l[1], gi[1] = (l[0], gi[0]) if c[0] else ((v, g) if c[1] else (l[1], gi[1]))
            l[0], gi[0] = (v, g) if c[0] else (l[0], gi[0])
        found, got = False, FLT_MAX
        for i in range(4):
            if not found and gi[i] < G and l[i] >= ub and has_member[gi[i]]:
                found = True
                if l[i] <= hint:
                    got = l[i]
        walk = not found and not (l[3] > hint)
        if walk:
            walked += 1
            continue
        decided += 1
        assert got == want, (trial, list(lbg), list(has_member), ub, hint, l, gi, got, want)
    assert decided > 30000 and walked > 2000, (decided, walked)


def _med3(a, b, c):
    return sorted([a, b, c])[1]


def test_candidate_list_certificate_is_complete():
    """yy_hint_list_kernel's bookkeeping: the best four packed scores per half-wave (register number in the
    low 4 mantissa bits), tile labels tracked by value equality once per tile, the certificate "both
    halves' fourth-best < amin and no two taken entries decode to the same centroid".  Claim: a certified
    list is EXACTLY the set of centroids whose packed score is >= amin -- also when scores tie, which is
    where equality-tracked labels can go wrong (and must then be caught by the duplicate test)."""
    rs = numpy.random.RandomState(99)
    certified = refused = 0
    for trial in range(6000):
        ntiles = int(rs.randint(1, 5))
        K = 32 * ntiles
        if trial % 3 == 0:
            raw = rs.choice(numpy.array([-2.0, -1.0, -0.5, 0.25, 0.5, 1.0], numpy.float32), K)   # ties galore
        else:
            raw = (rs.randn(K) * (0.2 if trial % 3 == 1 else 2.0)).astype(numpy.float32)
        raw[rs.rand(K) < 0.05] = -numpy.inf                                                     # padding / NaN centroids
        amin = F(rs.choice([-3.0, -0.75, 0.0, 0.3, 0.75, 1.5]) if trial % 2 else numpy.sort(raw)[-int(rs.randint(1, 7))])
        packed_of = numpy.empty(K, numpy.float32)
        lists = []
        for half in (0, 1):
            v = [F(-numpy.inf)] * 4
            t = [0] * 4
            for tile in range(ntiles):
                o = list(v)
                for r in range(16):
                    c = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half
                    bits = (numpy.array([raw[c]], numpy.float32).view(numpy.uint32)[0] & numpy.uint32(0xFFFFFFF0)) | numpy.uint32(r)
                    x = numpy.array([bits], numpy.uint32).view(numpy.float32)[0]
                    if numpy.isnan(x):          # -inf packs into a NaN pattern: v_med3 / v_max ignore it
                        packed_of[c] = -numpy.inf
                        continue
                    packed_of[c] = x
                    v[3] = _med3(v[2], v[3], x)
                    v[2] = _med3(v[1], v[2], x)
                    v[1] = _med3(v[0], v[1], x)
                    v[0] = max(v[0], x)
                t = [t[o.index(v[i])] if v[i] in o else tile for i in range(4)]
            lists.append((v, t, half))
        taken, ok = [], True
        for v, t, half in lists:
            ok = ok and bool(v[3] < amin)
            for i in range(4):
                if v[i] >= amin:
                    r = int(numpy.array([v[i]], numpy.float32).view(numpy.uint32)[0] & 15)
                    c = t[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * half
                    if c in taken:
                        ok = False
                    taken.append(c)
        want = set(int(c) for c in range(K) if packed_of[c] >= amin)
        if not ok:
            refused += 1
            continue
        certified += 1
        assert set(taken) == want and len(taken) == len(want), (trial, amin, sorted(taken), sorted(want))
    assert certified > 1500 and refused > 1000, (certified, refused)
