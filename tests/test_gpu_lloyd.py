"""GPU parity tests of the Lloyd path: HIP kernels (through the C ABI) vs the CPU oracle.

Bar: assignments and the reassignment counter are BIT-EXACT.  Centroids: the reference's
kmeans_adjust shares ONE Kahan compensation term across all features (kmeans.cu:388,414-418), so
its per-feature sums carry plain-summation error ~sqrt(members)*2^-24 relative; our update
accumulates in fp64 and rounds once.  The stated tolerance is therefore the REFERENCE's own
error: rtol 2e-5 at a few hundred members per cluster."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda", 0)


def _assign(x, c, metric="L2", exact=False, asg0=None, filt="f16"):
    from kmcuda_amd.engine import Engine
    dev = _dev()
    n, d = x.shape
    k = c.shape[0]
    xs, cs = torch.from_numpy(x).to(dev), torch.from_numpy(c).to(dev)
    init = numpy.full(n, 0xFFFFFFFF, numpy.uint32) if asg0 is None else asg0.astype(numpy.uint32)
    asg = torch.from_numpy(init.view(numpy.int32).copy()).to(dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, metric, device=0)
    _assign.kind = eng.filter_kind()
    eng.set_filter(filt)
    eng.lloyd_assign(xs, cs, asg, prev, exact=exact)
    counters = eng.counters()
    eng.close()
    return asg.cpu().numpy().view(numpy.uint32), prev.cpu().numpy().view(numpy.uint32), counters


@pytest.mark.parametrize("n,d,k", [(3000, 2, 50), (1000, 7, 33), (2500, 16, 100), (2000, 64, 257),
                                   (4096, 128, 64), (5000, 256, 1024), (777, 300, 40), (3000, 384, 500),
                                   (2500, 512, 1024), (1500, 600, 64)])
@pytest.mark.parametrize("mode", ["f16", "f32", "exact"])
def test_assign_bit_exact(n, d, k, mode):
    """f16: two-stage f16 matrix-core filter (default), f32: f32 matrix-core filter, exact: the reference
    arithmetic for every pair -- all must reproduce the oracle bit for bit."""
    rs = numpy.random.RandomState(n + d + k)
    x = rs.rand(n, d).astype(numpy.float32)
    c = x[rs.choice(n, k, replace=False)].copy()
    got, prev, counters = _assign(x, c, exact=(mode == "exact"), filt="f32" if mode == "exact" else mode)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    assert (prev == ref_prev).all()
    assert counters[0] == ref_changed


@pytest.mark.parametrize("n,d,k", [(777, 300, 40), (3000, 384, 500), (2500, 512, 1024)])
def test_assign_bit_exact_register_resident_filter_at_its_widest(n, d, k, monkeypatch):
    """257..512 features take the LDS-streamed filter by default (tests/test_gpu_wide.py); the register-resident
    filter's one-operand-set instantiation for these widths stays reachable (KMCUDA_AMD_WIDE_MIN_D=513: carried bounds
    at these widths) and stays exact."""
    monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "513")
    rs = numpy.random.RandomState(n + d + k)
    x = rs.rand(n, d).astype(numpy.float32)
    c = x[rs.choice(n, k, replace=False)].copy()
    got, prev, counters = _assign(x, c)
    assert _assign.kind == (1, 512)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (got == ref).all() and (prev == ref_prev).all() and counters[0] == ref_changed


@pytest.mark.parametrize("filt", ["f16", "f32"])
def test_assign_gaussian_and_second_pass(filt):
    rs = numpy.random.RandomState(5)
    x = (rs.randn(6000, 256) * 3 + rs.randn(1, 256)).astype(numpy.float32)
    c = x[rs.choice(6000, 300, replace=False)].copy()
    got, _, _ = _assign(x, c, filt=filt)
    ref, _, _ = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    # second pass from the previous assignments with perturbed centroids: counter = #changed
    c2 = (c + rs.randn(*c.shape).astype(numpy.float32) * 0.05).astype(numpy.float32)
    got2, prev2, counters = _assign(x, c2, asg0=got, filt=filt)
    ref2, ref_prev2, ref_changed = oracle.lloyd_assign(x, c2, assignments=ref)
    assert (got2 == ref2).all() and (prev2 == ref_prev2).all()
    assert counters[0] == ref_changed


@pytest.mark.parametrize("settle", ["1", "0"])
@pytest.mark.parametrize("filt", ["f16", "f32"])
def test_assign_ties_duplicates_nans(filt, settle, monkeypatch):
    # settle: the undecided rows in one launch (lloyd_settle_kernel, default) / the pair + full-scan kernels
    monkeypatch.setenv("KMCUDA_AMD_SETTLE", settle)
    rs = numpy.random.RandomState(11)
    x = rs.rand(4000, 256).astype(numpy.float32)
    c = x[rs.choice(4000, 96, replace=False)].copy()
    c[40] = c[3]          # duplicate centroids: exact ties, the lower index must win
    c[77] = c[3]
    c[10, 5] = numpy.nan  # NaN centroid: never chosen (kmeans.cu:425-426)
    c[11, :] = numpy.inf
    x[5, 0] = numpy.nan   # "insane" sample -> assignment K (kmeans.cu:312, :349-356)
    x[6, 17] = numpy.nan  # NaN elsewhere: search fails, row left untouched
    x[7] = c[3]           # exact hit on a duplicated centroid
    x[8, 3] = numpy.inf   # inf feature: NaN / inf scores must end in the exact kernels
    x[9, :] = 1e30        # centred halves overflow to inf in the split: same
    got, prev, counters = _assign(x, c, filt=filt)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    assert (prev == ref_prev).all()
    assert counters[0] == ref_changed
    assert got[5] == 96 and got[6] == 0xFFFFFFFF and got[7] == 3
    assert not numpy.isin(got, [10, 11, 40, 77]).any()


@pytest.mark.parametrize("settle", ["1", "0"])
@pytest.mark.parametrize("n,d", [(1500, 64), (5000, 256), (700, 20), (900, 132)])
@pytest.mark.parametrize("filt", ["f16", "f32"])
def test_assign_all_rows_flagged_still_exact(filt, n, d, settle, monkeypatch):
    # every centroid duplicated: the filter can decide nothing, the exact kernels decide all (d = 20, 132: the
    # staged chunks of lloyd_settle_kernel end inside a chunk)
    monkeypatch.setenv("KMCUDA_AMD_SETTLE", settle)
    rs = numpy.random.RandomState(13)
    x = rs.rand(n, d).astype(numpy.float32)
    base = x[rs.choice(n, 20, replace=False)]
    c = numpy.concatenate([base, base]).astype(numpy.float32)
    got, _, counters = _assign(x, c, filt=filt)
    ref, _, _ = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    assert counters[1] + counters[3] == n  # nothing decided by the filter itself
    assert (got < 20).all()


def test_assign_angular():
    rs = numpy.random.RandomState(17)
    x = rs.randn(3000, 256).astype(numpy.float32)
    x /= numpy.linalg.norm(x, axis=1)[:, None]
    c = x[rs.choice(3000, 64, replace=False)].copy()
    got, _, _ = _assign(x, c, metric="cos")
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    # acosf is libm on the CPU and ocml on the GPU: a row whose two nearest centroids are within a last place of each
    # other in the oracle's own arithmetic may resolve either way -- nothing else may differ (tests/_angular.py)
    from _angular import assert_only_acos_matters
    assert_only_acos_matters(x, c, got, ref, "test_assign_angular", max_fraction=1e-3)


def test_update_matches_oracle():
    from kmcuda_amd.engine import Engine
    dev = _dev()
    rs = numpy.random.RandomState(23)
    n, d, k = 20000, 256, 100
    x = rs.rand(n, d).astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    a1, p1, _ = oracle.lloyd_assign(x, c0)
    c1, cc1 = oracle.adjust(x, p1, a1, c0, numpy.zeros(k, numpy.uint32))
    a2, p2, _ = oracle.lloyd_assign(x, c1, assignments=a1)
    c2, cc2 = oracle.adjust(x, p2, a2, c1, cc1)

    eng = Engine(n, d, k, "L2", device=0)
    xs = torch.from_numpy(x).to(dev)
    cen = torch.from_numpy(c0.copy()).to(dev)
    ccounts = torch.zeros(k, dtype=torch.int32, device=dev)
    delta = torch.empty(k * d, dtype=torch.float64, device=dev)
    dcount = torch.empty(k, dtype=torch.int32, device=dev)

    def step(prev, cur):
        pt = torch.from_numpy(prev.view(numpy.int32).copy()).to(dev)
        ct = torch.from_numpy(cur.view(numpy.int32).copy()).to(dev)
        eng.move_deltas(xs, pt, ct, delta, dcount)
        eng.apply_delta(delta, dcount, cen, ccounts)
        eng.sync()
        return cen.cpu().numpy(), ccounts.cpu().numpy().view(numpy.uint32)

    g1, gc1 = step(p1, a1)
    assert (gc1 == cc1).all()
    numpy.testing.assert_allclose(g1, c1, rtol=2e-5, atol=1e-7)
    g2, gc2 = step(p2, a2)
    assert (gc2 == cc2).all()
    numpy.testing.assert_allclose(g2, c2, rtol=2e-5, atol=1e-7)
    eng.close()


def test_transpose_roundtrip():
    from kmcuda_amd.engine import Engine
    dev = _dev()
    rs = numpy.random.RandomState(3)
    for rows, cols in [(1000, 256), (333, 77), (64, 64), (5, 1000), (1028, 132), (4, 4), (260, 8), (1000, 254)]:   # 16-byte and 4-byte kernels
        a = rs.rand(rows, cols).astype(numpy.float32)
        src = torch.from_numpy(a).to(dev)
        dst = torch.empty(cols * rows, dtype=torch.float32, device=dev)
        back = torch.empty(cols * rows, dtype=torch.float32, device=dev)
        eng = Engine(rows, cols, 2, "L2", device=0)
        eng.transpose(src, rows, cols, dst)
        eng.transpose(dst, cols, rows, back)
        eng.sync()
        assert (dst.cpu().numpy().reshape(cols, rows) == a.T).all()
        assert (back.cpu().numpy().reshape(rows, cols) == a).all()
        eng.close()


@pytest.mark.parametrize("filt", ["f16", "f32"])
def test_config_a_100k_256_1024(filt):
    """BASELINE config A shape: one assignment pass at 100000x256, K=1024, bit-exact."""
    rs = numpy.random.RandomState(0)
    x = rs.rand(100000, 256).astype(numpy.float32)
    c = x[rs.choice(100000, 1024, replace=False)].copy()
    got, _, counters = _assign(x, c, filt=filt)
    ref, _, changed = oracle.lloyd_assign(x, c)
    assert (got == ref).all()
    assert counters[0] == changed == 100000
    assert counters[1] + counters[3] < 20000  # the filter decides the bulk of the rows


def test_orders_with_torch_default_stream():
    """Engine steps and torch ops on the same buffers without any explicit synchronisation: torch's
    default stream is the legacy NULL stream, the engine's own stream must order with it (it once was a
    non-blocking stream: .cpu() read assignments the kernels had not written yet, and an NCCL
    all-reduce would have read the deltas early)."""
    from kmcuda_amd.engine import Engine
    dev = _dev()
    rs = numpy.random.RandomState(77)
    n, d, k = 200000, 256, 1024
    x = rs.rand(n, d).astype(numpy.float32)
    xs = torch.from_numpy(x).to(dev)
    eng = Engine(n, d, k, "L2", device=0)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    cbuf = torch.empty((k, d), dtype=torch.float32, device=dev)
    ref_asg = None
    for it in range(4):
        c = x[rs.choice(n, k, replace=False)].copy()
        cbuf.copy_(torch.from_numpy(c))          # torch op, then the engine reads cbuf
        eng.lloyd_assign(xs, cbuf, asg, prev)
        got = asg.cpu().numpy().view(numpy.uint32).copy()   # torch op right behind the engine's kernels
        ref, _, _ = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (got == ref).all(), "pass %d" % it
        ref_asg = ref
    eng.close()


def test_stage2_on_a_converged_state():
    """Stage 2 of the default filter (contenders scored in fp32) on centroids that have converged onto
    unstructured data (small best/second gaps: the state that sends most rows past stage 1)."""
    rs = numpy.random.RandomState(123)
    n, d, k = 20000, 256, 300
    x = rs.rand(n, d).astype(numpy.float32)
    c = x[rs.choice(n, k, replace=False)].copy()
    asg = None
    for _ in range(6):   # a few oracle Lloyd steps: centroids drift toward the data mean
        asg, _, _ = oracle.lloyd_assign(x, c, assignments=asg)
        for j in range(k):
            m = asg == j
            if m.any():
                c[j] = x[m].mean(axis=0)
    got, prev, counters = _assign(x, c, asg0=asg)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=asg)
    assert (got == ref).all() and (prev == ref_prev).all() and counters[0] == ref_changed
