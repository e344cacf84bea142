"""Bounds carried from pass to pass (kmcuda_amd/csrc/lloyd_carry.hip; the Yinyang phase of kmeans_cuda()'s default
schedule; the reference's purpose: src/kmeans.cu:1028-1263).

Bar: a loop whose passes spare the rows their bounds decide is INDISTINGUISHABLE from the loop of plain passes --
assignments, previous assignments, reassignment counters and (hence) centroids bit for bit after every iteration,
on clustered data (where most rows are spared), unstructured data (where nearly none is), with NaN rows, clusters
that die, rows as halves, through both update paths (apply alone; the update fused with the next preparation) and
through kmeans_cuda(yinyang_t > 0) as a whole.  And, directly: EVERY pass of every carrying loop below is replayed by the
oracle (the reference's kmeans_assign_lloyd on the CPU) from the same centroids and previous assignments -- the bounds
are not only compared with HIP-without-bounds."""
import os

import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _blobs(n, d, k, seed, spread=6.0):
    rs = numpy.random.RandomState(seed)
    centres = rs.rand(k, d) * spread
    lab = rs.randint(0, k, n)
    x = (centres[lab] + rs.randn(n, d)).astype(numpy.float32)
    return x


def _uniform(n, d, seed):
    return numpy.random.RandomState(seed).rand(n, d).astype(numpy.float32)


def _run_pair(x, k, iters, carry_from, fused, half=False, seed=5, monkeypatch=None, list_max=None, metric="L2",
              cos_tol=2e-3):
    """Two loops over the same rows and seeds, one carrying bounds from iteration `carry_from` on; yields per
    iteration (changed_plain, changed_carry) after asserting the states equal.  Returns the carry engine's stats."""
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(seed)
    if half:
        x = x.astype(numpy.float16).astype(numpy.float32)
    init = x[rs.choice(len(x), k, replace=False)].copy()
    finite_rows = numpy.isfinite(init).all(axis=1)
    if not finite_rows.all():   # never seed from a NaN row
        good = numpy.nonzero(numpy.isfinite(x).all(axis=1))[0]
        init[~finite_rows] = x[good[:int((~finite_rows).sum())]]
    xs = torch.from_numpy(x).to(dev)
    loops = []
    for which in range(2):
        h = xs.to(torch.float16) if half else None
        b = HipBackend(xs, k, metric, device_index=0, half_rows=h)
        loop = ShardedLloyd(b, len(x))
        loop.set_centroids(torch.from_numpy(init).to(dev))
        loops.append(loop)
    plain, carry = loops
    if list_max is not None:
        os.environ["KMCUDA_AMD_CARRY_MAX"] = str(list_max)
    try:
        log = []
        _run_pair.kinds = []
        for it in range(iters):
            if it == carry_from:
                carry.b.engine.set_carry(True)
            # the state the CARRY loop's pass starts from (the oracle replays the pass from it below)
            cen_in = carry.b.centroids.cpu().numpy().copy()
            asg_in = carry.b.assignments.cpu().numpy().view(numpy.uint32).copy()
            for loop in loops:
                if fused:
                    loop.step(tolerance=0.0)      # device-side stop rule + update fused with the next preparation
                else:
                    loop.step()                    # plain apply; the preparation runs inside the next pass
            for loop in loops:
                loop.b.synchronize()
            a0, a1 = plain.b.assignments.cpu().numpy(), carry.b.assignments.cpu().numpy()
            p0, p1 = plain.b.assignments_prev.cpu().numpy(), carry.b.assignments_prev.cpu().numpy()
            assert (a0 == a1).all(), "iteration %d: %d assignments differ" % (it, int((a0 != a1).sum()))
            assert (p0 == p1).all(), "iteration %d: %d previous assignments differ" % (it, int((p0 != p1).sum()))
            c0, c1 = plain.b.centroids.cpu().numpy(), carry.b.centroids.cpu().numpy()
            assert (c0.view(numpy.uint32) == c1.view(numpy.uint32)).all(), "iteration %d: centroids differ" % it
            # The oracle in the loop: EVERY pass of the loop that carries bounds -- spared rows,
            # listed rows, pair certificates and all -- against the reference's kmeans_assign_lloyd restated on the CPU,
            # from the same centroids and previous assignments: assignments, previous assignments, reassignment count.
            # (angular: ocml's and libm's acosf differ in the last place -- tolerance only)
            ref, ref_prev, ref_changed = oracle.lloyd_assign(x, cen_in, assignments=asg_in,
                                                             metric=oracle.L2 if metric == "L2" else oracle.COS)
            a1u, p1u = a1.view(numpy.uint32), p1.view(numpy.uint32)
            if metric == "L2":
                assert (a1u == ref).all(), "iteration %d: %d assignments differ from the oracle's" % (it, int((a1u != ref).sum()))
                assert (p1u == ref_prev).all(), "iteration %d: previous assignments differ from the oracle's" % it
                assert int((a1u != p1u).sum()) == ref_changed, (it, int((a1u != p1u).sum()), ref_changed)
                assert carry.changed_last() == ref_changed, (it, carry.changed_last(), ref_changed)
            else:
                from _angular import assert_only_acos_matters
                assert_only_acos_matters(x, cen_in, a1u, ref, "iteration %d" % it, max_fraction=cos_tol)
            log.append(int((a0 != p0).sum()))
            _run_pair.kinds.append(carry.b.engine.filter_kind())
        spared, last = carry.b.engine.carry_stats()
        _run_pair.paired = carry.b.engine.carry_pair_stats()
        return log, spared, last
    finally:
        if list_max is not None:
            del os.environ["KMCUDA_AMD_CARRY_MAX"]


@pytest.mark.parametrize("fused", [False, True], ids=["apply", "apply+prepare"])
@pytest.mark.parametrize("shape", [(60000, 64, 64), (50000, 256, 200), (30000, 24, 40), (40000, 100, 33),
                                   (30000, 512, 48), (20000, 300, 24), (25000, 12, 12)],
                         ids=lambda s: "%dx%d@%d" % s)
def test_carried_passes_equal_plain_passes_on_clustered_rows(shape, fused, monkeypatch):
    n, d, k = shape
    # (257..512 features: the streamed filter, as beyond 512 -- test_carried_passes_on_the_streamed_filter; the
    #  register-resident filter's 512-wide instantiation carries bounds too: the parametrisation below runs these two
    #  shapes on it, KMCUDA_AMD_WIDE_MIN_D=513)
    if 256 < d <= 512:
        monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "513")
    # half as many blobs as centroids: most blobs are shared by two centroids, whose rows the bounds rarely decide;
    # as many blobs as centroids (the first shape): most rows are decided by their bounds from the first carried pass on
    x = _blobs(n, d, k if n == 60000 else max(8, k // 2), seed=n + d, spread=10.0 if n == 60000 else 6.0)
    # (the padded shapes -- 300 -> 512, 12 -> 16 features -- always take the LISTED pass, whatever the list's length:
    #  the gathered, zero-padding instantiations of the coarse stage are the ones to exercise there)
    padded = d in (300, 12)
    log, spared, last = _run_pair(x, k, iters=14, carry_from=3, fused=fused, list_max=1.0 if padded else None)
    assert spared > 0, (log, spared, last)
    if 256 < d <= 512:
        assert _run_pair.kinds[-1] == (1, 512), _run_pair.kinds
    if n == 60000:
        assert spared > 3 * n and last < n // 2, (log, spared, last)


@pytest.mark.parametrize("metric", ["L2", "cos"])
@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16x2"])
@pytest.mark.parametrize("fused", [False, True], ids=["apply", "apply+prepare"])
def test_pair_certificates_send_shared_blob_rows_to_the_pair_kernel(fused, half, metric, monkeypatch):
    """Two centroids per blob: a third of the rows sit too close to the border between the two for stage 1 to decide,
    pass after pass.  Stage 2 leaves them a pair certificate (the two contenders, an upper bound of both distances, a
    lower bound of every other centroid's); while it survives the drifts the pair kernel alone looks at the row.
    Same states as plain passes, iteration by iteration; KMCUDA_AMD_CARRY_PAIRS=0 is the A/B."""
    rs = numpy.random.RandomState(12)
    cen = rs.rand(48, 128) * 9.0
    x = cen[rs.randint(0, 48, 80000)] + rs.randn(80000, 128)
    if metric == "cos":   # (the same statements in score space)
        x /= numpy.linalg.norm(x, axis=1, keepdims=True)
        # (the angular certificates were off in round 5 -- test_angular_half_rows_stress_case_* has the story -- and are
        #  on again by default; stated here so that the test does not depend on the default)
        monkeypatch.setenv("KMCUDA_AMD_CARRY_PAIRS", "1")
    x = x.astype(numpy.float32)
    log, spared, last = _run_pair(x, 96, iters=14, carry_from=3, fused=fused, half=half, list_max=1.0, metric=metric)
    paired = _run_pair.paired
    assert paired > len(x) // 2, (log, spared, last, paired)
    monkeypatch.setenv("KMCUDA_AMD_CARRY_PAIRS", "0")
    log0, spared0, last0 = _run_pair(x, 96, iters=14, carry_from=3, fused=fused, half=half, list_max=1.0, metric=metric)
    assert _run_pair.paired == 0 and log0 == log


@pytest.mark.parametrize("d", [50, 30, 18])
def test_pair_certificates_with_feature_counts_the_settle_kernel_does_not_take(d):
    """D % 4 != 0: the pairs go to lloyd_pair_kernel, the scans to lloyd_exact_kernel (two streams) -- the carried
    pairs arrive there like stage 2's own."""
    rs = numpy.random.RandomState(d)
    cen = rs.rand(20, d) * 9.0
    x = (cen[rs.randint(0, 20, 50000)] + rs.randn(50000, d)).astype(numpy.float32)
    log, spared, last = _run_pair(x, 40, iters=12, carry_from=3, fused=(d != 30), list_max=1.0)
    assert _run_pair.paired > 0, (log, spared, last)


@pytest.mark.parametrize("metric", ["L2", "cos"])
def test_a_list_longer_than_the_listed_pass_s_grid_is_strided_over(metric, monkeypatch):
    """The listed pass is launched for the host's ESTIMATE of the list; the device-side list can be any length (a
    cluster that dies makes every drift bound +inf: every row is listed).  KMCUDA_AMD_CARRY_GRID=3: every block takes
    dozens of trips over the list."""
    monkeypatch.setenv("KMCUDA_AMD_CARRY_GRID", "3")
    x = _blobs(60000, 64, 64, seed=60064, spread=10.0)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    log, spared, last = _run_pair(x, 64, iters=12, carry_from=3, fused=True, metric=metric, list_max=1.0)
    assert spared > 0 and last > 3 * 256, (log, spared, last)
    y = _uniform(30000, 200, seed=4)   # (padded rows, next to nothing spared: ~117 trips per block)
    if metric == "cos":
        y /= numpy.linalg.norm(y, axis=1, keepdims=True)
    _run_pair(y, 100, iters=6, carry_from=2, fused=False, metric=metric, list_max=1.0)


@pytest.mark.parametrize("list_max", [None, 1.0, 0.0], ids=["default", "always-listed", "never-listed"])
def test_carried_passes_equal_plain_passes_on_unstructured_rows(list_max):
    # uniform rows: the bounds spare next to nothing -- the listed pass (forced: KMCUDA_AMD_CARRY_MAX=1) then covers
    # nearly every row from the rows themselves, the whole pass (=0) only counts the list
    x = _uniform(60000, 256, seed=3)
    log, spared, last = _run_pair(x, 300, iters=9, carry_from=2, fused=True, list_max=list_max)
    assert last > 0


def test_carried_passes_with_nan_rows_dead_clusters_and_halves():
    x = _blobs(40000, 64, 20, seed=77)
    x[5, 0] = numpy.nan          # kmeans.cu:312: assignment K
    x[9, 7] = numpy.nan          # a NaN elsewhere: the row is never touched ("search failed")
    x[100:110] = 1.0e4           # far rows: one cluster of their own, then another seed dies
    _run_pair(x, 48, iters=10, carry_from=2, fused=True)
    _run_pair(x, 48, iters=10, carry_from=2, fused=False)
    xh = _blobs(40000, 64, 48, seed=78, spread=10.0)   # as many blobs as centroids: most rows are spared
    log, spared, _ = _run_pair(xh, 48, iters=10, carry_from=2, fused=True, half=True)
    assert spared > 0, log


@pytest.mark.parametrize("case", [((20000, 640, 40), "L2", False, True), ((12000, 1024, 64), "L2", False, False),
                                  ((16000, 768, 48), "cos", False, True), ((16000, 576, 30), "L2", True, True),
                                  ((9000, 1536, 24), "cos", True, False), ((14000, 2048, 20), "L2", False, True),
                                  ((30000, 512, 48), "L2", False, True), ((20000, 300, 24), "L2", False, False),
                                  ((24000, 384, 32), "cos", True, True)],
                         ids=lambda c: "%dx%d@%d-%s-%s-%s" % (c[0] + (c[1], "fp16" if c[2] else "fp32", "fused" if c[3] else "apply")))
def test_carried_passes_on_the_streamed_filter(case):
    """Rows wider than 256 features (lloyd_wide.hip, round 5): the streamed filter's passes leave the same two bounds per
    row (MODE 2), carry_skip_kernel reads the rows' records from that filter's table, the listed rows are gathered by
    index (MODE 3).  Every pass against plain passes and against the oracle, as everywhere in this file."""
    (n, d, k), metric, half, fused = case
    # (half as many blobs as centroids: most blobs are shared by two centroids and the run keeps moving)
    x = _blobs(n, d, max(8, k // 2), seed=n + d, spread=6.0)
    if metric == "cos":
        x = (x / numpy.linalg.norm(x, axis=1, keepdims=True)).astype(numpy.float32)
    # (list_max = 1: every pass with a counted list is a LISTED one -- shared blobs at these widths list ~60 % of the
    #  rows, which the default policy answers with whole passes and then a pause: test_boundary_cpu.py has that logic)
    log, spared, last = _run_pair(x, k, iters=12, carry_from=3, fused=fused, half=half, metric=metric, list_max=1.0)
    assert spared > 0, (log, spared, last)
    assert all(kind == (2, (d + 63) // 64 * 64) for kind in _run_pair.kinds), _run_pair.kinds


def test_carried_passes_on_the_streamed_filter_with_nan_rows_ties_and_lists_of_every_length():
    x = _blobs(15000, 600, 25, seed=31)
    x[5, 0] = numpy.nan
    x[9, 7] = numpy.nan
    x[100:110] = 3.0e3
    x[200:260] = x[200]          # identical rows: ties
    for list_max in (None, 0.0, 1.0):
        _run_pair(x, 40, iters=10, carry_from=2, fused=True, list_max=list_max)
    xu = _uniform(12000, 704, seed=4)     # unstructured: the bounds decide next to nothing, the policy pauses them
    _run_pair(xu, 60, iters=14, carry_from=2, fused=False)


def test_kmeans_cuda_default_schedule_carries_on_1024_feature_rows(monkeypatch):
    """kmeans_cuda(yinyang_t = 0.1) on 1024-feature rows: bounds carried by the streamed filter behind the hand-over
    point; KMCUDA_AMD_CARRY=0 runs the same passes plain.  Same progress lines, same results."""
    from kmcuda_amd import kmeans_cuda
    from test_gpu_kmeans import StdoutListener
    x = _blobs(30000, 1024, 40, seed=19, spread=8.0)
    monkeypatch.setenv("KMCUDA_AMD_CARRY_MAX", "1.0")   # (listed passes whatever the list's length: see above)
    outs = []
    for carry in ("1", "0"):
        monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        out = StdoutListener()
        with out:
            cen, asg = kmeans_cuda(x, 64, init="random", seed=3, tolerance=0.0005, yinyang_t=0.1, device=1, verbosity=2)
        text = out.text
        lines = [l for l in text.split("\n") if l.startswith("iteration")]
        outs.append((cen, asg, lines, text))
    assert outs[0][2] == outs[1][2]
    assert (outs[0][1] == outs[1][1]).all()
    assert (outs[0][0].view(numpy.uint32) == outs[1][0].view(numpy.uint32)).all()
    assert "carrying per-sample distance bounds" in outs[0][3]
    assert [l for l in outs[0][3].split("\n") if l.startswith("carried bounds:")]   # (how many it spared: the loops above)


def test_angular_half_rows_stress_case_default_schedule_equals_plain_passes(monkeypatch):
    """Trial 201 of `scripts/stress_carry_api.py 60 58` (round 5): 300 000 x 16 half rows of unit length, K = 130, two
    virtual shards, tolerance 0.001 -- with the angular PAIR certificates the whole call ended on other centroids than
    with plain passes (4525 assignments; round 4's build too), and two identical carried runs differed from each other.
    Cause (DESIGN_LOG 13.13): the reference's clamp p >= 1 -> distance 0 makes such centroids tie and the lowest index
    win; the pair kernel followed it, the plain passes' filters did not, and which rows took which path depended on the
    timing of the host's list reports.  Since round 6 every filter leaves those rows to the exact kernels
    (tests/test_gpu_angular_clamp.py) and the certificates are on again: the call is bit-identical to plain passes,
    and two carried runs to each other."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
    from stress_carry import make
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(58)
    for trial in range(202):   # (the script's draws, replayed)
        n = int(rs.choice([3000, 20000, 90000, 300000]))
        d = int(rs.choice([16, 32, 64, 100, 128, 256, 300, 512]))
        k = int(rs.choice([20, 64, 130, 300]))
        k = min(k, n // 20)
        metric = str(rs.choice(["L2", "cos"]))
        half = bool(rs.rand() < 0.25)
        shards = int(rs.choice([1, 1, 2, 3]))
        tol = float(rs.choice([0.01, 0.001, 0.0001, 0.00002]))
        init = str(rs.choice(["random", "kmeans++"])) if n <= 90000 else "random"
        kind, x = make(rs, n, d, k, metric)
        sd = int(rs.randint(1, 1000))
    assert (n, d, k, metric, half, shards, tol, kind, sd) == (300000, 16, 130, "cos", True, 2, 0.001, "nan", 998)
    x = numpy.nan_to_num(x, nan=0.5)
    x /= numpy.maximum(numpy.linalg.norm(x, axis=1, keepdims=True), 1e-12)
    x = x.astype(numpy.float16)
    monkeypatch.setenv("KMCUDA_AMD_VIRTUAL_SHARDS", "2")
    res = []
    monkeypatch.setenv("KMCUDA_AMD_CARRY_PAIRS", "1")
    for carry in ("1", "0", None, "1"):
        if carry is None:
            monkeypatch.delenv("KMCUDA_AMD_CARRY")
        else:
            monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        res.append(kmeans_cuda(x, k, init="random", seed=sd, tolerance=tol, yinyang_t=0.1, metric="cos", device=1, verbosity=0))
    for cen, asg in (res[0], res[2], res[3]):
        assert (asg == res[1][1]).all()
        assert (cen.view(numpy.uint16) == res[1][0].view(numpy.uint16)).all()


def test_kmeans_cuda_default_schedule_carries_and_equals_the_plain_schedule(monkeypatch):
    """kmeans_cuda(yinyang_t = 0.1): after the reference's hand-over point the default schedule carries bounds;
    KMCUDA_AMD_CARRY=0 runs the same passes plain.  Same progress lines, same results."""
    from kmcuda_amd import kmeans_cuda
    from test_gpu_kmeans import StdoutListener
    x = _blobs(120000, 64, 60, seed=9)
    outs = []
    for carry in ("1", "0"):
        monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        out = StdoutListener()
        with out:
            cen, asg = kmeans_cuda(x, 100, init="random", seed=3, tolerance=0.0005, yinyang_t=0.1, device=1, verbosity=2)
        text = out.text
        lines = [l for l in text.split("\n") if l.startswith("iteration")]
        outs.append((cen, asg, lines, text))
    assert outs[0][2] == outs[1][2]
    assert (outs[0][1] == outs[1][1]).all()
    assert (outs[0][0].view(numpy.uint32) == outs[1][0].view(numpy.uint32)).all()
    assert "carrying per-sample distance bounds" in outs[0][3]
    spared = [l for l in outs[0][3].split("\n") if l.startswith("carried bounds:")]
    assert spared and int(spared[0].split()[2]) > len(x), spared


def test_random_init_and_silent_group_clustering_leave_rand_where_the_reference_does():
    """init="random" draws the reference's N - 1 shuffle numbers from a restatement of glibc's generator and then sets
    rand()'s state (kmcuda_api.cpp: GlibcRand); a silent yinyang_t > 0 call on the default schedule replays the group
    clustering's draws instead of clustering (replay_group_seeding_draws).  What a caller's next rand() returns must
    be what it returns after the plain sequences."""
    import ctypes
    from kmcuda_amd import kmeans_cuda
    libc = ctypes.CDLL(None)
    x = _blobs(30000, 16, 12, seed=4)
    n, k = x.shape[0], 24
    # the reference's own sequence for init = "random": srand(seed), one draw per shuffle step
    libc.srand(11)
    chosen = list(range(n))
    for i in range(1, n):
        j = libc.rand() % (i + 1)
        chosen[i], chosen[j] = chosen[j], chosen[i]
    after_shuffle = libc.rand()
    cen, _ = kmeans_cuda(x, k, init="random", seed=11, tolerance=1.0, yinyang_t=0, device=1, verbosity=0)
    assert libc.rand() == after_shuffle
    assert (cen.view(numpy.uint32) == x[chosen[:k]].view(numpy.uint32)).all()   # tolerance 1: no update, centroids = seeds
    # yinyang_t > 0: silent (draws replayed) against verbose (groups clustered for their progress lines)
    nxt = []
    for verbosity in (0, 1):
        from test_gpu_kmeans import StdoutListener
        with StdoutListener():
            kmeans_cuda(x, k, init="random", seed=11, tolerance=0.001, yinyang_t=0.25, device=1, verbosity=verbosity)
        nxt.append(libc.rand())
    assert nxt[0] == nxt[1]


@pytest.mark.parametrize("metric", ["L2", "cos"])
def test_carried_bounds_over_virtual_shards(metric, monkeypatch):
    """Three row shards on one GPU (KMCUDA_AMD_VIRTUAL_SHARDS: every shard its own engine, bounds, drifts and row
    list; the centroids replicated): the carried schedule equals the plain one there too, and both equal one shard."""
    from kmcuda_amd import kmeans_cuda
    x = _blobs(90000, 32, 50, seed=21, spread=8.0)
    if metric == "cos":
        x /= numpy.linalg.norm(x, axis=1, keepdims=True)
    res = {}
    for shards, carry in ((1, "1"), (3, "1"), (3, "0")):
        monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        if shards > 1:
            monkeypatch.setenv("KMCUDA_AMD_VIRTUAL_SHARDS", str(shards))
        else:
            monkeypatch.delenv("KMCUDA_AMD_VIRTUAL_SHARDS", raising=False)
        res[(shards, carry)] = kmeans_cuda(x, 50, init="random", seed=3, tolerance=0.0002, yinyang_t=0.1, metric=metric,
                                           device=1, verbosity=0)
    a = res[(3, "1")]
    b = res[(3, "0")]
    assert (a[1] == b[1]).all() and (a[0].view(numpy.uint32) == b[0].view(numpy.uint32)).all()
    # one shard against three: the same assignments (the deltas are fp64 sums of fp32 values: the split moves
    # the centroids by rounding at most, which this well-separated data does not notice)
    assert (res[(1, "1")][1] == a[1]).mean() > 0.9999


def test_an_exact_pass_or_a_filter_change_between_carried_passes_voids_the_bounds():
    """The bounds describe the last pass's assignments: a pass of another kind in between (the exact kernels, the f32
    filter) must not leave stale bounds behind for the next carried pass."""
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    dev = torch.device("cuda", 0)
    x = _blobs(50000, 64, 30, seed=13, spread=9.0)
    init = x[numpy.random.RandomState(2).choice(len(x), 30, replace=False)].copy()
    xs = torch.from_numpy(x).to(dev)
    loops = []
    for which in range(2):
        b = HipBackend(xs, 30, "L2", device_index=0)
        loop = ShardedLloyd(b, len(x))
        loop.set_centroids(torch.from_numpy(init).to(dev))
        loops.append(loop)
    plain, carry = loops
    carry.b.engine.set_carry(True)
    for it in range(12):
        for loop in loops:
            loop.step()
        if it in (4, 8):   # scramble the carry engine's state with a pass of another kind, then restore the assignments
            e = carry.b.engine
            keep_a, keep_p = carry.b.assignments.clone(), carry.b.assignments_prev.clone()
            if it == 4:
                e.lloyd_assign(carry.b.samples, carry.b.centroids, carry.b.assignments, carry.b.assignments_prev, exact=True)
            else:
                e.set_filter("f32")
                e.lloyd_assign(carry.b.samples, carry.b.centroids, carry.b.assignments, carry.b.assignments_prev)
                e.set_filter("f16")
            # pretend the caller moved rows in between as well: bounds that survived would now be wrong
            carry.b.assignments.copy_(keep_a)
            carry.b.assignments_prev.copy_(keep_p)
            e.reset_counters(0)
        for loop in loops:
            loop.b.synchronize()
        assert (plain.b.assignments == carry.b.assignments).all(), it
        assert (plain.b.centroids.view(torch.int32) == carry.b.centroids.view(torch.int32)).all(), it


@pytest.mark.parametrize("blobs", [1024, 600], ids=["a-blob-per-centroid", "shared-blobs"])
def test_carried_passes_at_shard_scale(blobs):
    """One rank's share of the benchmark's shape -- 1M x 256 rows, K = 1024 -- on a mixture of Gaussians (as many as
    centroids: most rows spared; fewer: blobs shared by two centroids, whose rows live on pair certificates): 16
    iterations with and without the bounds, compared on the device after every iteration -- and four of the CARRIED
    passes (the first one, two in the middle, the last) replayed whole by the oracle: all 1M assignments, previous
    assignments and the reassignment count against the reference's kmeans_assign_lloyd."""
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    n, d, k = 1000000, 256, 1024
    centres = torch.rand((blobs, d), device=dev, generator=g) * 10.0
    x = torch.randn((n, d), device=dev, generator=g) + centres[torch.randint(0, blobs, (n,), device=dev, generator=g)]
    init = x[torch.randperm(n, device=dev, generator=g)[:k]].clone()
    xh = x.cpu().numpy()
    loops = []
    for which in range(2):
        loop = ShardedLloyd(HipBackend(x, k, "L2", device_index=0), n)
        loop.set_centroids(init)
        loops.append(loop)
    plain, carry = loops
    for it in range(16):
        if it == 3:
            carry.b.engine.set_carry(True)
        judged = it in (4, 7, 11, 15)
        if judged:
            cen_in = carry.b.centroids.cpu().numpy().copy()
            asg_in = carry.b.assignments.cpu().numpy().view(numpy.uint32).copy()
        for loop in loops:
            loop.step(tolerance=0.0)
        for loop in loops:
            loop.b.synchronize()
        assert torch.equal(plain.b.assignments, carry.b.assignments), it
        assert torch.equal(plain.b.assignments_prev, carry.b.assignments_prev), it
        assert torch.equal(plain.b.centroids.view(torch.int32), carry.b.centroids.view(torch.int32)), it
        if judged:
            ref, ref_prev, ref_changed = oracle.lloyd_assign(xh, cen_in, assignments=asg_in)
            got = carry.b.assignments.cpu().numpy().view(numpy.uint32)
            assert (got == ref).all(), (it, int((got != ref).sum()))
            assert (carry.b.assignments_prev.cpu().numpy().view(numpy.uint32) == ref_prev).all(), it
            assert carry.changed_last() == ref_changed, (it, carry.changed_last(), ref_changed)
    spared, last = carry.b.engine.carry_stats()
    paired = carry.b.engine.carry_pair_stats()
    if blobs == k:
        assert spared > 4 * n and last < n // 2, (spared, last)
    else:
        assert spared > n and paired > 0, (spared, last, paired)


def test_kmeans_cuda_default_schedule_under_the_strict_update_equals_the_oracle_s_lloyd(monkeypatch):
    """End to end against the oracle: kmeans_cuda(yinyang_t = 0.1) on the DEFAULT schedule (Lloyd down to 11 %, the group
    clustering, then passes that carry bounds and pair certificates) with the reference's serial update
    (KMCUDA_AMD_EXACT_UPDATE=1; KMCUDA_AMD_YY=carry keeps the default schedule under it).  Every pass is the
    reference's Lloyd arithmetic, so the call must be the oracle's kmeans(yinyang_t = 0) bit for bit: the progress
    lines' reassignment counts, the assignments, the centroids."""
    from kmcuda_amd import kmeans_cuda
    from test_gpu_kmeans import StdoutListener
    rs = numpy.random.RandomState(41)
    cen = rs.rand(70, 256) * 4.0
    x = (cen[rs.randint(0, 70, 60000)] + rs.randn(60000, 256)).astype(numpy.float32)
    monkeypatch.setenv("KMCUDA_AMD_EXACT_UPDATE", "1")
    monkeypatch.setenv("KMCUDA_AMD_YY", "carry")
    monkeypatch.setenv("KMCUDA_AMD_CARRY", "1")   # (whatever the run's expected length: kmcuda_api.cpp, iterations_left)
    kw = dict(init="random", seed=3, tolerance=0.0002)
    out = StdoutListener()
    with out:
        got_cen, got_asg = kmeans_cuda(x, 100, yinyang_t=0.1, device=1, verbosity=2, **kw)
    assert "carrying per-sample distance bounds" in out.text
    # the call's own progress lines: up to the hand-over point (then the nested group clustering's lines follow,
    # numbered from 1 again -- the reference prints them too, kmeans.cu:1062-1100), and after "Lloyd goes on"
    before, after = out.text.split("Lloyd goes on", 1)
    reass = []
    for l in before.split("\n"):
        if l.startswith("iteration"):
            if int(l.split()[1].rstrip(":")) == 1 and reass:
                break
            reass.append(int(l.split(":")[1].split()[0]))
    reass += [int(l.split(":")[1].split()[0]) for l in after.split("\n") if l.startswith("iteration")]
    spared = [l for l in out.text.split("\n") if l.startswith("carried bounds:")]
    assert spared and int(spared[0].split()[2]) > len(x), out.text[-600:]
    ocen, oasg, olog = oracle.kmeans(x, 100, yinyang_t=0, **kw)
    assert reass == list(olog), (reass, list(olog))
    assert (got_asg == oasg).all()
    assert (got_cen.view(numpy.uint32) == ocen.view(numpy.uint32)).all()


def test_kmeans_cuda_fp16_l2_carries_and_equals_the_plain_schedule(monkeypatch):
    """The fp16x2 path (half rows at the boundary, centroids rounded to half after every update): carried bounds
    against plain passes through kmeans_cuda()."""
    from kmcuda_amd import kmeans_cuda
    x = _blobs(100000, 64, 40, seed=31, spread=9.0).astype(numpy.float16)
    res = []
    for carry in ("1", "0"):
        monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        res.append(kmeans_cuda(x, 40, init="random", seed=5, tolerance=0.0002, yinyang_t=0.1, device=1, verbosity=0))
    assert res[0][0].dtype == numpy.float16
    assert (res[0][1] == res[1][1]).all()
    assert (res[0][0].view(numpy.uint16) == res[1][0].view(numpy.uint16)).all()


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16x2"])
def test_carried_passes_equal_plain_passes_with_the_angular_metric(half):
    """Angular metric: one number per row -- the certified gap of the scores -- moved by ||x|| (drift(a) + max drift).
    Unit rows in well-separated directions (most rows spared) and unit rows of a uniform cloud (next to none)."""
    rs = numpy.random.RandomState(3)
    cen = rs.randn(40, 64)
    x = cen[rs.randint(0, 40, 60000)] + 0.15 * rs.randn(60000, 64)
    x = (x / numpy.linalg.norm(x, axis=1, keepdims=True)).astype(numpy.float32)
    log, spared, last = _run_pair(x, 40, iters=12, carry_from=2, fused=True, metric="cos", half=half)
    assert spared > 2 * len(x), (log, spared, last)
    y = rs.rand(40000, 128)
    y = (y / numpy.linalg.norm(y, axis=1, keepdims=True)).astype(numpy.float32)
    _run_pair(y, 100, iters=8, carry_from=2, fused=False, metric="cos", half=half, list_max=1.0)


def test_kmeans_cuda_angular_carries_and_equals_the_plain_schedule(monkeypatch):
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(8)
    cen = rs.randn(30, 32)
    x = cen[rs.randint(0, 30, 80000)] + 0.2 * rs.randn(80000, 32)
    x = (x / numpy.linalg.norm(x, axis=1, keepdims=True)).astype(numpy.float32)
    from test_gpu_kmeans import StdoutListener
    res = []
    for carry in ("1", "0"):
        monkeypatch.setenv("KMCUDA_AMD_CARRY", carry)
        out = StdoutListener()
        with out:
            res.append(kmeans_cuda(x, 30, init="random", seed=5, tolerance=0.0002, yinyang_t=0.1, metric="cos", device=1,
                                   verbosity=2))
        if carry == "1":   # the bounds did decide rows (a report judged late or twice once paused them for good)
            spared = [l for l in out.text.split("\n") if l.startswith("carried bounds:")]
            assert spared and int(spared[0].split()[2]) > len(x), out.text[-600:]
    assert (res[0][1] == res[1][1]).all()
    assert (res[0][0].view(numpy.uint32) == res[1][0].view(numpy.uint32)).all()


def test_a_run_about_to_stop_does_not_start_carrying(monkeypatch):
    """The first pass after the hand-over point only LEAVES bounds (and allocates them); the default schedule starts
    them once the host has seen the run go on past an iteration after the hand-over -- a run that stops within a pass
    or two never starts them (round 4: such calls lost 12-16 % to bounds they never used).  Same iteration lines and
    results as plain passes either way; a long run of the same data does carry."""
    from kmcuda_amd import kmeans_cuda
    from test_gpu_kmeans import StdoutListener
    monkeypatch.delenv("KMCUDA_AMD_CARRY", raising=False)
    x = _blobs(200000, 64, 100, seed=19, spread=10.0)
    outs = {}
    for tol in (0.03, 0.00002):
        out = StdoutListener()
        with out:
            cen, asg = kmeans_cuda(x, 100, init="random", seed=3, tolerance=tol, yinyang_t=0.1, device=1, verbosity=2)
        outs[tol] = (out.text, cen, asg)
    short, long_ = outs[0.03][0], outs[0.00002][0]
    assert "Lloyd goes on" in short, short[-800:]   # (the hand-over point was reached ...)
    after = [l for l in short.split("Lloyd goes on", 1)[1].split("\n") if l.startswith("iteration")]
    assert len(after) <= 2, after                    # (... the run stopped within two iterations of it ...)
    assert "carrying per-sample distance bounds" not in short   # (... and the bounds were never started)
    assert "carrying per-sample distance bounds" in long_
    spared = [l for l in long_.split("\n") if l.startswith("carried bounds:")]
    assert spared and int(spared[0].split()[2]) > len(x), long_[-600:]
    monkeypatch.setenv("KMCUDA_AMD_CARRY", "0")
    for tol in (0.03, 0.00002):
        cen0, asg0 = kmeans_cuda(x, 100, init="random", seed=3, tolerance=tol, yinyang_t=0.1, device=1, verbosity=0)
        assert (asg0 == outs[tol][2]).all() and (cen0.view(numpy.uint32) == outs[tol][1].view(numpy.uint32)).all()
