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


@pytest.mark.parametrize("d,carries", [(640, False), (320, True)])
def test_set_carry_on_streamed_rows(d, carries):
    """kmamd_set_carry on an engine whose rows take the streamed filter.  Beyond 512 features: accepted, nothing
    carried, every pass a plain one.  257..512 features: the carried passes change to the register-resident filter (the
    one that leaves and reads bounds; its own row copy over the same frozen mean) and back when the bounds are switched
    off (include/kmcuda_amd.h: kmamd_filter_kind).  The oracle's assignments either way."""
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
    eng.set_row_cache(True)
    ref_asg = None
    c = x[rs.choice(n, k, replace=False)].copy()
    kinds = []
    for it in range(9):
        if it == 2:
            eng.set_carry(True)
        if it == 7:
            eng.set_carry(False)
        eng.reset_counters(0)
        eng.lloyd_assign(xs, torch.from_numpy(c).to(dev), asg, prev)
        ref, ref_prev, ref_changed = oracle.lloyd_assign(x, c, assignments=ref_asg)
        assert (asg.cpu().numpy().view(numpy.uint32) == ref).all() and eng.counters()[0] == ref_changed
        ref_asg = ref
        kinds.append(eng.filter_kind())
        c = (c + rs.randn(k, d).astype(numpy.float32) * 0.002).astype(numpy.float32)
    streamed = (2, (d + 63) // 64 * 64)
    if carries:
        assert kinds == [streamed] * 2 + [(1, 512)] * 5 + [streamed] * 2, kinds
        assert eng.carry_stats()[0] > n, eng.carry_stats()     # (tight blobs, tiny drifts: the bounds decide most rows)
    else:
        assert kinds == [streamed] * 9 and eng.carry_stats()[0] == 0, kinds
    eng.close()
