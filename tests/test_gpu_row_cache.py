"""Row cache of the coarse filter stage (kmamd_set_row_cache): x - mean as halves in matrix-core
operand order, built on the first pass, mean frozen afterwards.  Bar: assignments stay BIT-EXACT
against the oracle on every later pass, whatever the centroids do (the frozen mean only changes how
many rows the later stages have to settle)."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _engine(n, d, k, metric="L2"):
    from kmcuda_amd.engine import Engine
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return Engine(n, d, k, metric, device=0)


def _run_passes(x, cs, cached, rebuild_at=None):
    dev = torch.device("cuda", 0)
    n, d = x.shape
    k = cs[0].shape[0]
    xs = torch.from_numpy(x).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = _engine(n, d, k)
    if cached:
        eng.set_row_cache(True)
    out = []
    for i, c in enumerate(cs):
        if cached and rebuild_at is not None and i == rebuild_at:
            eng.set_row_cache(True)   # drops the copy; rebuilt with the mean of this pass's centroids
        eng.reset_counters(0)
        cd = torch.from_numpy(c).to(dev)
        eng.lloyd_assign(xs, cd, asg, prev)
        changed = eng.counters()[0]   # synchronises the engine's stream: only now are asg / prev final
        out.append((asg.cpu().numpy().view(numpy.uint32).copy(), prev.cpu().numpy().view(numpy.uint32).copy(), changed))
    eng.close()
    return out


@pytest.mark.parametrize("n,d,k", [(5000, 256, 1024), (3001, 64, 100), (2000, 16, 33), (4100, 100, 257), (300, 256, 64),
                                   (2600, 512, 300), (1900, 300, 70)])
def test_cached_passes_match_oracle(n, d, k, monkeypatch):
    if 256 < d <= 512:   # the register-resident filter's widest instantiation (the default there is the streamed
        monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "513")   # filter: tests/test_gpu_wide.py)
    rs = numpy.random.RandomState(n + k)
    x = rs.rand(n, d).astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    # centroids that wander: small steps, then a global shift that moves them far from the frozen mean
    cs = [c0, (c0 + rs.randn(k, d).astype(numpy.float32) * 0.02).astype(numpy.float32),
          (c0 * 0.9 + 0.05).astype(numpy.float32), (c0 + 3.0).astype(numpy.float32),
          (c0 * 0.5).astype(numpy.float32)]
    got = _run_passes(x, cs, cached=True)
    ref_asg = None
    for (asg, prev, changed), c in zip(got, cs):
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg == ref).all()
        assert (prev == ref_prev).all()
        assert changed == ref_changed
        ref_asg = ref


def test_cached_equals_uncached_and_rebuild():
    rs = numpy.random.RandomState(3)
    x = (rs.randn(6000, 256) * 2 + rs.randn(1, 256)).astype(numpy.float32)
    c0 = x[rs.choice(6000, 300, replace=False)].copy()
    cs = [c0] + [(c0 + rs.randn(300, 256).astype(numpy.float32) * s).astype(numpy.float32) for s in (0.01, 0.1, 0.5, 1.0)]
    a = _run_passes(x, cs, cached=False)
    b = _run_passes(x, cs, cached=True)
    c = _run_passes(x, cs, cached=True, rebuild_at=3)
    for (a1, p1, n1), (a2, p2, n2), (a3, p3, n3) in zip(a, b, c):
        assert (a1 == a2).all() and (p1 == p2).all() and n1 == n2
        assert (a1 == a3).all() and (p1 == p3).all() and n1 == n3


def test_cached_nan_rows_and_nonfinite_centroids():
    rs = numpy.random.RandomState(8)
    x = rs.rand(3000, 256).astype(numpy.float32)
    x[5, 0] = numpy.nan        # NaN first feature: assignment K (kmeans.cu:312)
    x[9, 17] = numpy.nan       # NaN elsewhere
    x[11] = 7.0e4              # centred value beyond the half range: never decided by the coarse stage
    c0 = x[rs.choice(numpy.arange(100, 3000), 64, replace=False)].copy()
    c1 = c0.copy()
    c1[3, 2] = numpy.nan       # NaN centroid: never chosen
    c1[7] = numpy.inf
    got = _run_passes(x, [c0, c1, c0], cached=True)
    ref_asg = None
    for (asg, prev, changed), c in zip(got, [c0, c1, c0]):
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg == ref).all() and (prev == ref_prev).all() and changed == ref_changed
        ref_asg = ref


@pytest.mark.parametrize("cached", [False, True])
def test_centroid_norms_at_the_edge_of_float_range(cached):
    """Centroids whose squared norm sits at the float range's end: the reference's serial Kahan sum_squares
    (csqr, what its distance uses) and the steady-state preparation's parallel sum of the same squares may land
    on different sides of FLT_MAX -- one says inf, the other finite.  Nothing may be DECIDED from that flag: such
    panels are far outside the half range, so every row goes to the exact kernels, which use csqr.  Passes 2 and
    3 run the one-kernel steady-state preparation when the row cache is on."""
    rs = numpy.random.RandomState(4)
    n, d, k = 3000, 256, 48
    x = rs.rand(n, d).astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()

    def edge(c, scale):
        c = c.copy()
        c[5] = (rs.rand(d).astype(numpy.float32) + 0.5) * numpy.float32(scale)   # ||c||^2 ~ FLT_MAX
        c[17] = -c[5]
        return c
    # 256 squares of ~(1.15e18)^2: sums between 0.9 and 1.1 of FLT_MAX depending on the draw and the order
    cs = [c0, edge(c0, 1.12e18), edge(c0 * 0.99, 1.16e18), edge(c0, 1.14e18)]
    got = _run_passes(x, cs, cached=cached)
    ref_asg = None
    for (asg, prev, changed), c in zip(got, cs):
        with numpy.errstate(over="ignore", invalid="ignore"):
            ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg == ref).all()
        assert (prev == ref_prev).all()
        assert changed == ref_changed
        ref_asg = ref


def test_kmeans_cuda_iterations_unchanged_by_cache(monkeypatch):
    """whole kmeans_cuda() runs: cached (default inside the call) vs KMCUDA_AMD_ROW_CACHE=0."""
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(21)
    x = rs.rand(20000, 64).astype(numpy.float32)
    monkeypatch.setenv("KMCUDA_AMD_EXACT_UPDATE", "1")
    res = []
    for veto in ("1", "0"):
        monkeypatch.setenv("KMCUDA_AMD_ROW_CACHE", veto)
        c, a = kmeans_cuda(x, 50, tolerance=0.002, init="k-means++", seed=5, yinyang_t=0, verbosity=0)
        res.append((c.copy(), a.copy()))
    assert (res[0][1] == res[1][1]).all()
    assert numpy.array_equal(res[0][0], res[1][0], equal_nan=True)


@pytest.mark.parametrize("n,d,k", [(1, 256, 1), (31, 256, 2), (33, 64, 3), (129, 512, 5), (257, 16, 2), (1000, 256, 1000),
                                   (65, 300, 65)])
@pytest.mark.parametrize("cached", [False, True])
def test_tiny_and_ragged_shapes(n, d, k, cached, monkeypatch):
    """row counts around the 32 / 64 / 128 / 256-row tiling steps, K around the 32 / 64-centroid tiles,
    K == N, a single row; with and without the row cache."""
    if 256 < d <= 512:   # (the register-resident filter: see test_cached_passes_match_oracle)
        monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "513")
    rs = numpy.random.RandomState(n * 7 + d + k)
    x = rs.rand(n, d).astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    cs = [c0, (c0 + rs.randn(k, d).astype(numpy.float32) * 0.01).astype(numpy.float32)]
    got = _run_passes(x, cs, cached=cached)
    ref_asg = None
    for (asg, prev, changed), c in zip(got, cs):
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg == ref).all() and (prev == ref_prev).all() and changed == ref_changed
        ref_asg = ref
