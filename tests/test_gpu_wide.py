"""Feature counts beyond 256 (the register-resident filters' home ground; their one instantiation for 257..512 features
is slower than this one and only runs under KMCUDA_AMD_WIDE_MIN_D=513): stage 1 streams BOTH operands through LDS
(kmcuda_amd/csrc/lloyd_wide.hip: 256 rows x 256 centroids per block, f16 matrix cores, best / second-best per row in
registers -- no score is ever written; a library GEMM into a score matrix until round 4), the same sweep over the rows
it lists collects their contenders, the exact kernels settle the rest.  Bar, as everywhere: assignments, previous
assignments and the reassignment counter BIT-EXACT against the oracle (kmeans_assign_lloyd, kmeans.cu:293-364) for any
input."""
import numpy
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _passes(x, cs, metric="L2", cached=False, half=False):
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    n, d = x.shape
    k = cs[0].shape[0]
    xs = torch.from_numpy(x).to(dev)
    x16 = xs.to(torch.float16) if half else None
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, metric, device=0)
    _passes.kind = eng.filter_kind()
    if half:
        eng.set_half_rows(x16)
    if cached:
        eng.set_row_cache(True)
    out = []
    for c in cs:
        eng.reset_counters(0)
        eng.lloyd_assign(xs, torch.from_numpy(c).to(dev), asg, prev)
        counters = eng.counters()
        out.append((asg.cpu().numpy().view(numpy.uint32).copy(), prev.cpu().numpy().view(numpy.uint32).copy(), counters))
    eng.close()
    return out


@pytest.mark.parametrize("cached", [False, True])
@pytest.mark.parametrize("n,d,k", [(3000, 1024, 1024), (1500, 600, 64), (2000, 520, 100), (900, 2048, 33),
                                   (4097, 768, 257), (50, 1536, 7), (20000, 640, 300), (70000, 576, 1100),
                                   (2600, 512, 300), (1900, 300, 70), (30000, 384, 1024), (129, 257, 5), (5000, 448, 64)])
def test_wide_rows_bit_exact(n, d, k, cached):
    rs = numpy.random.RandomState(n + d + k)
    x = rs.rand(n, d).astype(numpy.float32)
    c0 = x[rs.choice(n, k, replace=False)].copy()
    cs = [c0, (c0 + rs.randn(k, d).astype(numpy.float32) * 0.01).astype(numpy.float32), (c0 * 0.9 + 0.05).astype(numpy.float32)]
    got = _passes(x, cs, cached=cached)
    assert _passes.kind == (2, (d + 63) // 64 * 64)     # the streamed filter, operands padded to 64 features
    ref_asg = None
    for (asg, prev, counters), c in zip(got, cs):
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg == ref).all()
        assert (prev == ref_prev).all()
        assert counters[0] == ref_changed
        ref_asg = ref
    # the filter, not the exact scan, did the work (counters[1]: rows handed to the full scan)
    assert got[0][2][1] < n // 4


@pytest.mark.parametrize("switch", ["KMCUDA_AMD_WIDE", "KMCUDA_AMD_GEMM"])
def test_wide_rows_filter_off_is_the_exact_kernel(monkeypatch, switch):
    monkeypatch.setenv(switch, "0")
    rs = numpy.random.RandomState(3)
    x = rs.rand(700, 640).astype(numpy.float32)
    c = x[rs.choice(700, 40, replace=False)].copy()
    (asg, prev, counters), = _passes(x, [c])
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (asg == ref).all() and (prev == ref_prev).all() and counters[0] == ref_changed


def test_wide_rows_ties_nans_nonfinite_centroids():
    rs = numpy.random.RandomState(11)
    n, d, k = 2500, 800, 96
    x = rs.rand(n, d).astype(numpy.float32)
    c = x[rs.choice(n, k, replace=False)].copy()
    c[40] = c[3]          # duplicate centroids: exact ties, the lower index must win
    c[77] = c[3]
    c[10, 5] = numpy.nan  # NaN centroid: never chosen (kmeans.cu:425-426)
    c[11, :] = numpy.inf
    x[5, 0] = numpy.nan   # "insane" sample -> assignment K (kmeans.cu:312, :349-356)
    x[6, 17] = numpy.nan  # NaN elsewhere: search fails, row left untouched
    x[7] = c[3]           # exact hit on a duplicated centroid
    x[8, 3] = numpy.inf
    x[9, :] = 1e30        # centred halves overflow: never decided from half scores
    for cached in (False, True):
        (asg, prev, counters), = _passes(x, [c], cached=cached)
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
        assert (asg == ref).all()
        assert (prev == ref_prev).all()
        assert counters[0] == ref_changed
        assert asg[5] == k and asg[6] == 0xFFFFFFFF and asg[7] == 3


def test_wide_rows_angular_and_half_rows():
    rs = numpy.random.RandomState(17)
    x = rs.randn(2000, 768).astype(numpy.float32)
    x /= numpy.linalg.norm(x, axis=1)[:, None]
    c = x[rs.choice(2000, 64, replace=False)].copy()
    (asg, _, _), = _passes(x, [c], metric="cos")
    ref, _, _ = oracle.lloyd_assign(x, c, metric=oracle.COS)
    assert (asg != ref).mean() < 1e-3     # acosf: libm vs ocml (tests/test_gpu_lloyd.py::test_assign_angular)
    # fp16x2 path: the rows as halves feed the row operand; results as on the widened values
    xh = x.astype(numpy.float16).astype(numpy.float32)
    ch = c.astype(numpy.float16).astype(numpy.float32)
    (asg, prev, counters), = _passes(xh, [ch], half=True)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(xh, ch)
    assert (asg == ref).all() and counters[0] == ref_changed


def test_wide_rows_whole_run_through_the_boundary():
    """kmeans_cuda() on 1024-feature rows: the stop rule on the device, the fp64 update, the row copy kept across the
    iterations -- against the oracle's run from the same seeds (same stop iteration; assignments equal up to the
    update's last-bit differences)."""
    from kmcuda_amd import kmeans_cuda
    rs = numpy.random.RandomState(2)
    centres = rs.rand(40, 1024).astype(numpy.float32) * 4
    x = (centres[rs.randint(0, 40, 6000)] + 0.5 * rs.randn(6000, 1024)).astype(numpy.float32)
    cen, asg = kmeans_cuda(x, 40, init="k-means++", seed=7, tolerance=0.001, yinyang_t=0, device=1)
    ocen, oasg, olog = oracle.kmeans(x, 40, init="k-means++", seed=7, tolerance=0.001, yinyang_t=0)
    assert (asg != oasg).mean() < 1e-3
    numpy.testing.assert_allclose(cen, ocen, rtol=2e-4, atol=2e-4)
    ref, _, _ = oracle.lloyd_assign(x, cen)
    assert (ref == asg).all()     # the returned assignments ARE the reference's for the returned centroids


def test_wide_rows_yinyang_schedules(monkeypatch):
    """yinyang_t > 0 on 768-feature rows: the default schedule keeps running Lloyd passes through the wide filter
    (carrying bounds once the run goes on, tests/test_gpu_carry.py); the reference schedule runs the exact Yinyang
    kernels (its bounds have no matrix-core filter at this width).
    Same hand-over point and lines up to it; equally good clusterings."""
    from kmcuda_amd import kmeans_cuda
    from test_gpu_kmeans import StdoutListener
    rs = numpy.random.RandomState(8)
    centres = rs.rand(24, 768).astype(numpy.float32) * 3
    x = (centres[rs.randint(0, 24, 5000)] + 0.4 * rs.randn(5000, 768)).astype(numpy.float32)
    res = {}
    for schedule in ("default", "reference"):
        monkeypatch.delenv("KMCUDA_AMD_YY", raising=False)
        if schedule == "reference":
            monkeypatch.setenv("KMCUDA_AMD_YY", "reference")
        out = StdoutListener()
        with out:
            c, a = kmeans_cuda(x, 24, init="k-means++", seed=3, tolerance=0.0005, yinyang_t=0.2, device=1, verbosity=1)
        lines = [ln for ln in out.text.splitlines() if ln.startswith("iteration")]
        res[schedule] = (lines, c, a, "refreshing Yinyang bounds" in out.text)
    assert res["reference"][3] or len(res["reference"][0]) == len(res["default"][0])
    assert not res["default"][3]
    assert abs(len(res["default"][0]) - len(res["reference"][0])) <= 3
    assert (res["default"][2] != res["reference"][2]).mean() < 0.02
    ref, _, _ = oracle.lloyd_assign(x, res["default"][1])
    assert (ref == res["default"][2]).all()


def test_which_filter_serves_which_width(monkeypatch):
    """<= 256 features: register-resident (padded to 16 .. 256); above: streamed, padded to 64; KMCUDA_AMD_WIDE_MIN_D
    moves the border (513: the register-resident filter's 512-wide instantiation; results the same either way)."""
    from kmcuda_amd.engine import Engine

    def kind(d):
        eng = Engine(1000, d, 10, "L2", device=0)
        out = eng.filter_kind()
        eng.close()
        return out
    assert kind(256) == (1, 256) and kind(100)[0] == 1 and kind(257) == (2, 320) and kind(512) == (2, 512) and kind(513) == (2, 576)
    monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "513")
    assert kind(300) == (1, 512) and kind(512) == (1, 512) and kind(513) == (2, 576)
    monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "128")
    assert kind(128) == (2, 128) and kind(100)[0] == 1
    monkeypatch.delenv("KMCUDA_AMD_WIDE_MIN_D")
    monkeypatch.setenv("KMCUDA_AMD_WIDE", "0")
    assert kind(300) == (1, 512) and kind(600) == (0, 0)
    rs = numpy.random.RandomState(4)
    x = rs.rand(3000, 200).astype(numpy.float32)
    c = x[rs.choice(3000, 70, replace=False)].copy()
    monkeypatch.delenv("KMCUDA_AMD_WIDE")
    monkeypatch.setenv("KMCUDA_AMD_WIDE_MIN_D", "1")      # the streamed filter on narrow rows: the same answers
    (asg, prev, counters), = _passes(x, [c])
    assert _passes.kind == (2, 256)
    ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c)
    assert (asg == ref).all() and (prev == ref_prev).all() and counters[0] == ref_changed


@pytest.mark.parametrize("d", [640, 320])
def test_set_carry_on_streamed_rows(d):
    """kmamd_set_carry on an engine whose rows take the streamed filter: it carries the bounds itself (MODE 2 / 3 of
    lloyd_wide_kernel; tests/test_gpu_carry.py has the side-by-side loops).  The oracle's assignments in every pass, rows
    spared while the bounds are on."""
    from kmcuda_amd.engine import Engine
    dev = torch.device("cuda", 0)
    rs = numpy.random.RandomState(9)
    n, k = 6000, 50
    cen = rs.rand(k, d).astype(numpy.float32) * 5
    x = (cen[rs.randint(0, k, n)] + 0.3 * rs.randn(n, d)).astype(numpy.float32)
    xs = torch.from_numpy(x).to(dev)
    asg = torch.full((n,), -1, dtype=torch.int32, device=dev)
    prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
    eng = Engine(n, d, k, "L2", device=0)
    assert eng.filter_kind() == (2, (d + 63) // 64 * 64)
    eng.set_row_cache(True)
    ref_asg = None
    c = x[rs.choice(n, k, replace=False)].copy()
    spared = []
    for it in range(9):
        if it == 2:
            eng.set_carry(True)
        if it == 7:
            eng.set_carry(False)
        eng.reset_counters(0)
        eng.lloyd_assign(xs, torch.from_numpy(c).to(dev), asg, prev)
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg.cpu().numpy().view(numpy.uint32) == ref).all() and eng.counters()[0] == ref_changed
        assert (prev.cpu().numpy().view(numpy.uint32) == ref_prev).all()
        ref_asg = ref
        spared.append(eng.carry_stats()[0])
        c = (c + rs.randn(k, d).astype(numpy.float32) * 0.002).astype(numpy.float32)
    # bounds on from pass 2, which leaves them; pass 3 moves them and counts its would-be list; 4, 5 and 6 are listed
    assert spared[3] == 0 and spared[6] > n and spared[8] == spared[6], spared
    eng.close()
