"""GPU parity tests of the Yinyang path (reference: kmeans.cu:431-672, :1028-1263): the HIP
kernels through the C ABI vs the CPU oracle.  Bar: bounds, drifts, the passed set, assignments
and the reassignment counter are BIT-EXACT (all of them are produced by the reference's exact
arithmetic); the end-to-end pins are the reference's own (test.py:228-234, :459-466)."""
import numpy
import pytest

import oracle
from test_gpu_kmeans import StdoutListener, _validate

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _t(a, dev):
    if a.dtype == numpy.uint32:
        a = a.view(numpy.int32)
    return torch.from_numpy(numpy.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("mode", ["mfma", "nohint", "exact"])
@pytest.mark.parametrize("n,d,k,G,metric", [(3000, 2, 50, 5, "L2"), (2500, 33, 64, 6, "L2"),
                                            (4000, 256, 128, 12, "L2"), (5000, 256, 1024, 102, "L2"),
                                            (3000, 64, 50, 7, "L2"), (2000, 64, 40, 4, "cos")])
def test_yinyang_steps_bit_exact(n, d, k, G, metric, mode, monkeypatch):
    """Both implementations of the Yinyang steps -- the matrix-core filtered kernels
    (yinyang_mfma.hip) and the plain exact kernels (yinyang.hip, KMCUDA_AMD_YY_EXACT=1) -- against
    the oracle: bounds after the refresh, drifts, then bounds / passed set / assignments / counters
    after one global+local filter pass."""
    from kmcuda_amd.engine import Engine
    monkeypatch.setenv("KMCUDA_AMD_YY_EXACT", "1" if mode == "exact" else "0")
    monkeypatch.setenv("KMCUDA_AMD_YY_HINT", "0" if mode == "nohint" else "1")
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(n + d)
    x = rs.rand(n, d).astype(numpy.float32)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1)[:, None]
    m = oracle.COS if metric == "cos" else oracle.L2
    c0 = x[rs.choice(n, k, replace=False)].copy()
    a1, p1, _ = oracle.lloyd_assign(x, c0, metric=m)
    c1, cc1 = oracle.adjust(x, p1, a1, c0, numpy.zeros(k, numpy.uint32), metric=m)
    a2, p2, _ = oracle.lloyd_assign(x, c1, assignments=a1, metric=m)
    groups = rs.randint(0, G, k).astype(numpy.uint32)
    groups[groups == G - 1] = 0          # leave one group EMPTY: its bound must stay FLT_MAX
    bounds = oracle.yy_init(x, c1, a2, groups, G, metric=m)
    c2, cc2 = oracle.adjust(x, p2, a2, c1, cc1, metric=m)
    drifts = oracle.yy_drifts(c1, c2, groups, G, metric=m)
    ra, rprev, rb, rpassed, rchanged = oracle.yy_filters(x, c2, groups, G, drifts, a2, bounds, metric=m)

    eng = Engine(n, d, k, metric, device=0)
    eng.yy_configure(G, groups)
    xs = _t(x, dev)
    gb = torch.empty((G + 1) * n, dtype=torch.float32, device=dev)
    asg = _t(a2, dev)
    eng.yy_init(xs, _t(c1, dev), asg, gb)
    eng.sync()
    got_b = gb.cpu().numpy().reshape(G + 1, n)
    if metric == "cos":
        # acosf: libm (oracle) vs ocml (GPU) differ in the last ulp -- the reference's CUDA acosf is
        # a third implementation; angular parity is tolerance-only
        numpy.testing.assert_allclose(got_b, bounds, rtol=0, atol=1e-6)
    else:
        assert (got_b.view(numpy.uint32) == bounds.view(numpy.uint32)).all()

    dr = torch.empty(k * d + k, dtype=torch.float32, device=dev)
    dr[:k * d] = _t(c1, dev).ravel()
    gdr = torch.empty(G, dtype=torch.float32, device=dev)
    cen2 = _t(c2, dev)
    eng.yy_drifts(cen2, dr, gdr)
    eng.sync()
    if metric == "cos":
        numpy.testing.assert_allclose(dr[k * d:].cpu().numpy(), drifts[k * d:], rtol=0, atol=1e-6)
        eng.close()
        return
    assert (dr[k * d:].cpu().numpy().view(numpy.uint32) == drifts[k * d:].view(numpy.uint32)).all()
    assert (gdr.cpu().numpy().view(numpy.uint32) == drifts[:G].view(numpy.uint32)).all()

    prev = torch.empty(n, dtype=torch.int32, device=dev)
    passed = torch.empty(n, dtype=torch.int32, device=dev)
    eng.reset_counters(-1)
    eng.yy_filters(xs, cen2, dr, gdr, asg, prev, gb, passed)
    counters = eng.counters()
    assert counters[2] == len(rpassed)
    got_passed = numpy.sort(passed.cpu().numpy().view(numpy.uint32)[:counters[2]])
    assert (got_passed == rpassed).all()
    assert (asg.cpu().numpy().view(numpy.uint32) == ra).all()
    assert counters[0] == rchanged
    assert (prev.cpu().numpy().view(numpy.uint32) == rprev).all()
    assert (gb.cpu().numpy().reshape(G + 1, n).view(numpy.uint32) == rb.view(numpy.uint32)).all()
    eng.close()


@pytest.mark.parametrize("n,d,k,sizes", [(2000, 64, 200, (150, 49, 1)), (1500, 256, 300, (1, 70, 33, 32, 31, 133)),
                                         (1000, 16, 100, (97, 0, 3))])
def test_yinyang_init_large_and_tiny_groups(n, d, k, sizes):
    """kmeans_yy_init (kmeans.cu:431-485) with groups of more than one 32-slot tile (the panel restarts them at
    every tile boundary and carries the minimum), of exactly a tile, of a single member (rows of that cluster
    have no other member: the bound stays FLT_MAX) and an empty one: bounds bit-exact against the oracle."""
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(k + d)
    x = rs.rand(n, d).astype(numpy.float32)
    cen = x[rs.choice(n, k, replace=False)].copy()
    asg, _, _ = oracle.lloyd_assign(x, cen)
    G = len(sizes)
    groups = numpy.concatenate([numpy.full(m, g, numpy.uint32) for g, m in enumerate(sizes)])
    assert len(groups) == k
    groups = groups[rs.permutation(k)]
    bounds = oracle.yy_init(x, cen, asg, groups, G)
    eng = Engine(n, d, k, "L2", device=0)
    eng.yy_configure(G, groups)
    gb = torch.empty((G + 1) * n, dtype=torch.float32, device=dev)
    eng.yy_init(_t(x, dev), _t(cen, dev), _t(asg, dev), gb)
    eng.sync()
    got = gb.cpu().numpy().reshape(G + 1, n)
    assert (got.view(numpy.uint32) == bounds.view(numpy.uint32)).all()
    eng.close()


@pytest.mark.parametrize("hint", ["f16", "f16cache", "off"])
@pytest.mark.parametrize("n,d,k,G,data", [(6000, 256, 256, 25, "uniform"), (8000, 64, 100, 10, "uniform"),
                                          (5000, 16, 64, 6, "uniform"), (6000, 256, 128, 12, "blobs"),
                                          (4000, 100, 60, 6, "blobs"), (3000, 24, 200, 20, "uniform")])
def test_yinyang_many_passes_bit_exact(n, d, k, G, data, hint, monkeypatch):
    """Six consecutive update + drift + filter passes against the oracle: bounds, assignments and
    the passed set stay BIT-EXACT from pass to pass, with the local filter's second-best estimate
    (yinyang_hint.hip: estimate kernel, hinted kernel, plain kernel for the rows it hands over) and
    without it (KMCUDA_AMD_YY_HINT=0: the f32 matrix-core kernel of yinyang_mfma.hip alone, the cross-check);
    "f16cache": the hinted kernels take their operands from the engine's row cache."""
    from kmcuda_amd.engine import Engine
    monkeypatch.setenv("KMCUDA_AMD_YY_EXACT", "0")
    monkeypatch.setenv("KMCUDA_AMD_YY_HINT", {"f16": "1", "f16cache": "1", "off": "0"}[hint])
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(n + d + k)
    if data == "uniform":
        x = rs.rand(n, d).astype(numpy.float32)
    else:
        x = numpy.concatenate([rs.randn(n // 10, d) + 3 * rs.randn(1, d) for _ in range(10)]).astype(numpy.float32)
        x = x[rs.permutation(len(x))]
    n = len(x)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    a_prev, p_prev, _ = oracle.lloyd_assign(x, c0)
    cen, cc = oracle.adjust(x, p_prev, a_prev, c0, numpy.zeros(k, numpy.uint32))
    asg, prv, _ = oracle.lloyd_assign(x, cen, assignments=a_prev)
    groups = (rs.permutation(k) % G).astype(numpy.uint32)
    bounds = oracle.yy_init(x, cen, asg, groups, G)

    eng = Engine(n, d, k, "L2", device=0)
    eng.yy_configure(G, groups)
    xs = _t(x, dev)
    if hint == "f16cache":
        # the sweeps' operands from the Lloyd coarse stage's row cache (what a kmeans_cuda() run has by the
        # time Yinyang starts): one assignment pass builds it and freezes the mean
        eng.set_row_cache(True)
        tmp_a = torch.zeros(n, dtype=torch.int32, device=dev)
        tmp_p = torch.zeros(n, dtype=torch.int32, device=dev)
        eng.lloyd_assign(xs, _t(cen, dev), tmp_a, tmp_p)
        eng.sync()
    gb = _t(bounds.ravel().copy(), dev)
    gasg = _t(asg, dev)
    gprev = torch.empty(n, dtype=torch.int32, device=dev)
    passed = torch.empty(n, dtype=torch.int32, device=dev)
    dr = torch.empty(k * d + k, dtype=torch.float32, device=dev)
    gdr = torch.empty(G, dtype=torch.float32, device=dev)
    total_passed = 0
    for it in range(6):
        new_cen, cc = oracle.adjust(x, prv, asg, cen, cc)
        drifts = oracle.yy_drifts(cen, new_cen, groups, G)
        asg, prv, bounds, rpassed, rchanged = oracle.yy_filters(x, new_cen, groups, G, drifts, asg, bounds)
        dr[:k * d] = _t(cen, dev).ravel()
        gcen = _t(new_cen, dev)
        eng.yy_drifts(gcen, dr, gdr)
        eng.reset_counters(-1)
        eng.yy_filters(xs, gcen, dr, gdr, gasg, gprev, gb, passed)
        counters = eng.counters()
        assert counters[2] == len(rpassed), "pass %d" % it
        assert counters[0] == rchanged, "pass %d" % it
        assert (gasg.cpu().numpy().view(numpy.uint32) == asg).all(), "pass %d" % it
        assert (gb.cpu().numpy().reshape(G + 1, n).view(numpy.uint32) == bounds.view(numpy.uint32)).all(), "pass %d" % it
        total_passed += len(rpassed)
        cen = new_cen
    stats = eng.yy_hint_stats()
    if hint != "off" and d >= 16:
        assert stats[0] == total_passed
        assert stats[1] == sum(stats[2:])
        print("hinted rows %d, handed over %d (no estimate %d, bound not holding %d, candidate bound %d, second minimum %d)"
              % tuple(stats))
    else:
        assert stats == [0] * 6
    eng.close()


@pytest.mark.parametrize("schedule", ["default", "reference", "switch-at-0.11", "switch-at-0.03"])
def test_kmeanspp_yinyang_15_3(fixture13k, monkeypatch, schedule):
    """test.py:228-234: 15 Lloyd + 3 Yinyang iterations.  `reference`: the reference's fixed schedule (bounds from
    11 % reassignments on).  `default`: bounds only when they can pay for their refresh (kmcuda_api.cpp:
    BoundsModel) -- on this fixture they cannot, every iteration is the reference's Lloyd pass, and the count is the
    same 18 because Yinyang is exact.  `switch-at-*`: the default loop forced to hand over to the bounds at that
    reassignment fraction (the hand-over takes effect one iteration after it is asked for)."""
    from kmcuda_amd import kmeans_cuda
    if schedule == "reference":
        monkeypatch.setenv("KMCUDA_AMD_YY", "reference")
    elif schedule.startswith("switch-at-"):
        monkeypatch.setenv("KMCUDA_AMD_YY_SWITCH", schedule[len("switch-at-"):])
    out = StdoutListener()
    with out:
        centroids, assignments = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, verbosity=2, seed=3,
                                             tolerance=0.01, yinyang_t=0.1)
    assert out.iterations() == 15 + 3
    _validate(fixture13k, centroids, assignments, 0.01)
    if schedule != "default":
        assert "refreshing Yinyang bounds" in out.text
    else:
        assert "refreshing Yinyang bounds" not in out.text


def test_schedules_agree_on_a_row_cached_shape(monkeypatch):
    """yinyang_t > 0 at a shape whose Lloyd passes run the row-cached two-stage filter (D = 256: the centroid
    update is then fused with the next pass's preparation), on the reference's schedule, the default one and with
    the bounds forced in late: Yinyang is exact, so the runs agree up to rows within rounding of a tie.  (Round 3: the fused update looked at the Lloyd phase's raised stop flag and skipped every update
    of the Yinyang phase on this path -- 8M x 256 "converged" after one bounds pass; the 13000 x 2 fixture never takes
    the fused path and did not see it.)"""
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(12)
    x = rs.rand(30000, 256).astype(numpy.float32)
    runs = {}
    for schedule in ("default", "reference", "switch-at-0.05", "reference-3-shards"):
        monkeypatch.delenv("KMCUDA_AMD_YY", raising=False)
        monkeypatch.delenv("KMCUDA_AMD_YY_SWITCH", raising=False)
        monkeypatch.delenv("KMCUDA_AMD_VIRTUAL_SHARDS", raising=False)
        if schedule.startswith("reference"):
            monkeypatch.setenv("KMCUDA_AMD_YY", "reference")
            if schedule.endswith("shards"):
                monkeypatch.setenv("KMCUDA_AMD_VIRTUAL_SHARDS", "3")   # the multi-shard host loop on one GPU
        elif schedule.startswith("switch-at-"):
            monkeypatch.setenv("KMCUDA_AMD_YY_SWITCH", schedule[len("switch-at-"):])
        out = StdoutListener()
        with out:
            c, a = kmeans_cuda(x, 96, init="random", device=1, verbosity=1, seed=5, tolerance=0.002, yinyang_t=0.1)
        lines = [ln for ln in out.text.splitlines() if ln.startswith("iteration")]
        runs[schedule] = (lines, a, c, "refreshing Yinyang bounds" in out.text)
    assert runs["reference"][3] and runs["switch-at-0.05"][3] and runs["reference-3-shards"][3] and not runs["default"][3]
    assert len(runs["default"][0]) > 12

    def counts(lines):
        return [int(ln.split()[2]) for ln in lines]

    def inertia(c, a):
        return float(((x - c[a]) ** 2).sum())
    base = counts(runs["default"][0])
    for schedule in ("reference", "switch-at-0.05", "reference-3-shards"):
        got = counts(runs[schedule][0])
        # a bounds pass decides by the reference's Yinyang arithmetic (sqrt of a Kahan sum of squared differences),
        # a Lloyd pass by -2 x.c + |c|^2: a row within rounding of a tie may go either way (measured: the first
        # difference is ONE row of 30000, at the first bounds pass), after which the two runs are two equally valid
        # Lloyd trajectories.  Up to the hand-over the lines are identical; afterwards the counts track each other
        hand_over = next(i for i, (u, v) in enumerate(zip(got, base)) if u != v) if got != base else len(base)
        assert hand_over >= 4, (schedule, got[:8], base[:8])
        assert abs(len(got) - len(base)) <= 3, schedule
        for u, v in list(zip(got, base))[:12]:
            assert abs(u - v) <= max(3, 0.02 * v), (schedule, got[:12], base[:12])
        # (unstructured data: a one-row difference at the hand-over grows to a few percent of the rows over the
        # twenty-odd iterations that follow -- measured 5.5 % -- at equal quality)
        assert (runs[schedule][1] != runs["default"][1]).mean() < 0.2, schedule
        assert abs(inertia(runs[schedule][2], runs[schedule][1]) / inertia(runs["default"][2], runs["default"][1]) - 1) < 1e-3


@pytest.mark.parametrize("speculate", ["1", "0"])
def test_lloyd_stop_rule_on_device_equals_host(fixture13k, monkeypatch, speculate):
    """The stop rule decided by the update kernel, with the next pass enqueued before the host has seen the
    count (default), against the host-side test in front of every update (KMCUDA_AMD_SPECULATE=0): same
    progress lines, same assignments, same centroids (one update behind the assignments, kmeans.cu:991-1000)."""
    from kmcuda_amd import kmeans_cuda
    monkeypatch.setenv("KMCUDA_AMD_SPECULATE", speculate)
    out = StdoutListener()
    with out:
        c, a = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, verbosity=1, seed=3, tolerance=0.01, yinyang_t=0)
    lines = [ln for ln in out.text.splitlines() if ln.startswith("iteration")]
    monkeypatch.setenv("KMCUDA_AMD_SPECULATE", "0" if speculate == "1" else "1")
    out2 = StdoutListener()
    with out2:
        c2, a2 = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, verbosity=1, seed=3, tolerance=0.01, yinyang_t=0)
    lines2 = [ln for ln in out2.text.splitlines() if ln.startswith("iteration")]
    assert lines == lines2 and len(lines) > 5
    assert (a == a2).all()
    assert numpy.array_equal(c, c2, equal_nan=True)
    # the returned centroids are the ones the returned assignments were computed FROM
    import oracle
    ref, _, _ = oracle.lloyd_assign(fixture13k, c)
    assert (ref == a).all()


def test_256_features_cosine_yinyang_9():
    # test.py:459-466
    from kmcuda_amd import kmeans_cuda
    numpy.random.seed(0)
    arr = numpy.random.rand(1000, 256).astype(numpy.float32)
    arr /= numpy.linalg.norm(arr, axis=1)[:, numpy.newaxis]
    out = StdoutListener()
    with out:
        kmeans_cuda(arr, 10, init="kmeans++", metric="cos", device=1, verbosity=3, yinyang_t=0.1, seed=3)
    assert out.iterations() == 9


def test_yinyang_equals_lloyd_outcome(fixture13k):
    from kmcuda_amd import kmeans_cuda
    c1, a1 = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, seed=3, tolerance=0.01, yinyang_t=0)
    c2, a2 = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, seed=3, tolerance=0.01, yinyang_t=0.1)
    assert (a1 != a2).mean() < 0.01
    numpy.testing.assert_allclose(c1, c2, rtol=1e-3, atol=1e-4)


def test_yinyang_virtual_shards(fixture13k, monkeypatch):
    from kmcuda_amd import kmeans_cuda
    c1, a1 = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, seed=3, tolerance=0.01, yinyang_t=0.1)
    monkeypatch.setenv("KMCUDA_AMD_VIRTUAL_SHARDS", "3")
    c3, a3 = kmeans_cuda(fixture13k, 50, init="kmeans++", device=1, seed=3, tolerance=0.01, yinyang_t=0.1)
    assert (a1 != a3).mean() < 0.002
