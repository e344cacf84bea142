"""Parity at the sizes that are benchmarked ("What's weak" 1): the oracle judges whole passes
over 1M x 256 rows against K = 1024 centroids on a state several iterations into a run -- the regime
bench.py times -- instead of the few-thousand-row cases of the other files.  The assignment pass of the
8M-row bench state itself is checked by `bench.py` (its "verify" entry: >= 1M rows of the timed state
against the oracle, outside the timed region); here the same check runs inside the test suite."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _t(a, dev):
    if a.dtype == numpy.uint32:
        a = a.view(numpy.int32)
    return torch.from_numpy(numpy.ascontiguousarray(a)).to(dev)


def _late_state(n, d, k, iters, seed=0):
    """Rows + a device Lloyd loop run `iters` iterations (row cache on, default filter): the state a bench
    shard is in when it is timed."""
    from kmcuda_amd.distributed import HipBackend, ShardedLloyd
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    for s in range(0, n, 1 << 20):
        x[s:s + (1 << 20)].uniform_(0.0, 1.0, generator=gen)
    b = HipBackend(x, k, "L2", device_index=0, row_cache=True)
    loop = ShardedLloyd(b, n)
    loop.set_centroids(x[torch.randperm(n, generator=gen, device=dev)[:k]].clone())
    for _ in range(iters):
        loop.step()
    torch.cuda.synchronize()
    return x, b, loop


def test_lloyd_pass_1m_rows_late_state_vs_oracle():
    """One assignment pass at 1M x 256 @ 1024 on the state after 7 iterations: every row's assignment,
    previous assignment and the reassignment counter against oracle.lloyd_assign; then the NEXT pass as
    well (steady-state preparation path, sync-free update in between)."""
    n, d, k = 1000000, 256, 1024
    x, b, loop = _late_state(n, d, k, 7)
    xh = x.cpu().numpy()
    for _ in range(2):
        cen = b.centroids.cpu().numpy()
        before = b.assignments.cpu().numpy().view(numpy.uint32).copy()
        b.reset_changed()
        b.assign()
        b.synchronize()
        got = b.assignments.cpu().numpy().view(numpy.uint32)
        prev = b.assignments_prev.cpu().numpy().view(numpy.uint32)
        ref, ref_prev, ref_changed = oracle.lloyd_assign(xh, cen, assignments=before)
        assert (got == ref).all()
        assert (prev == ref_prev).all()
        assert b.engine.counters()[0] == ref_changed
        b.fill_reduce_buffer(loop.buf)
        b.apply(loop.buf)
    b.engine.close()


def test_yinyang_pass_1m_rows_vs_oracle():
    """Bounds refresh + one update / drift / global + local filter pass at 1M x 256 @ 1024, G = 102
    (BASELINE config B's shape at one eighth of its rows) against the oracle: bounds, passed set,
    assignments, counters -- bit for bit."""
    from kmcuda_amd.engine import Engine
    n, d, k, G = 1000000, 256, 1024, 102
    x, b, loop = _late_state(n, d, k, 6, seed=3)
    dev = x.device
    xh = x.cpu().numpy()
    c1 = b.centroids.cpu().numpy()
    a1 = b.assignments.cpu().numpy().view(numpy.uint32).copy()
    cc1 = b.ccounts.cpu().numpy().view(numpy.uint32).copy()
    b.engine.close()
    # one more assignment with the oracle so that (prev, cur) are the oracle's own
    a2, p2, _ = oracle.lloyd_assign(xh, c1, assignments=a1)
    rs = numpy.random.RandomState(1)
    groups = (rs.permutation(k) % G).astype(numpy.uint32)
    bounds = oracle.yy_init(xh, c1, a2, groups, G)
    c2, _ = oracle.adjust(xh, p2, a2, c1, cc1)
    drifts = oracle.yy_drifts(c1, c2, groups, G)
    ra, rprev, rb, rpassed, rchanged = oracle.yy_filters(xh, c2, groups, G, drifts, a2, bounds)

    eng = Engine(n, d, k, "L2", device=0)
    eng.yy_configure(G, groups)
    gb = torch.empty((G + 1) * n, dtype=torch.float32, device=dev)
    asg = _t(a2, dev)
    eng.yy_init(x, _t(c1, dev), asg, gb)
    eng.sync()
    assert (gb.cpu().numpy().reshape(G + 1, n).view(numpy.uint32) == bounds.view(numpy.uint32)).all()
    dr = torch.empty(k * d + k, dtype=torch.float32, device=dev)
    dr[:k * d] = _t(c1, dev).ravel()
    gdr = torch.empty(G, dtype=torch.float32, device=dev)
    cen2 = _t(c2, dev)
    eng.yy_drifts(cen2, dr, gdr)
    prev = torch.empty(n, dtype=torch.int32, device=dev)
    passed = torch.empty(n, dtype=torch.int32, device=dev)
    eng.reset_counters(-1)
    eng.yy_filters(x, cen2, dr, gdr, asg, prev, gb, passed)
    counters = eng.counters()
    assert counters[2] == len(rpassed)
    assert (numpy.sort(passed.cpu().numpy().view(numpy.uint32)[:counters[2]]) == rpassed).all()
    assert (asg.cpu().numpy().view(numpy.uint32) == ra).all()
    assert counters[0] == rchanged
    assert (prev.cpu().numpy().view(numpy.uint32) == rprev).all()
    assert (gb.cpu().numpy().reshape(G + 1, n).view(numpy.uint32) == rb.view(numpy.uint32)).all()
    eng.close()


def test_knn_200k_rows_vs_oracle():
    """k-NN (k = 10) over 200 000 x 256 rows of a 1024-blob mixture with its k-means clustering, through
    knn_cuda(): neighbour lists (indices AND order) equal to the oracle's for every row, and the same
    count of evaluated distances."""
    from kmcuda_amd import kmeans_cuda, knn_cuda
    from test_gpu_kmeans import StdoutListener
    n, d, K = 200000, 256, 1024
    rs = numpy.random.RandomState(5)
    centres = (rs.rand(K, d) * 10).astype(numpy.float32)
    x = (centres[rs.randint(0, K, n)] + 0.3 * rs.randn(n, d)).astype(numpy.float32)
    cen, asg = kmeans_cuda(x, K, init="k-means++", seed=7, tolerance=0.01, yinyang_t=0, device=1)
    out = StdoutListener()
    with out:
        nb = knn_cuda(10, x, cen, asg, device=1, verbosity=1)
    ref, calced = oracle.knn(10, x, cen, asg)
    assert (nb == ref).all()
    line = [l for l in out.text.split("\n") if l.startswith("calculated")][0]
    assert abs(float(line.split()[1]) - calced / (float(n) * n)) < 1e-6
